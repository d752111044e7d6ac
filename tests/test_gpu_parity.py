"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  The host-buffer cases live in parity_cases.py (shared with the CPU thread-emulator
run of the same kernel sources); the device-resident ones (torch tensors in HBM) are here.
Tolerances (BASELINE.json north_star: 1e-5 relative on the float32 tensors):
  f32 compute : |got - ref| <= 1e-5 * max(1, |ref|)  on broadband inputs
  f64 compute : |got - ref| <= 3e-7 * max(1, |ref|)  (float32 output rounding only)
"""
import os

import numpy as np
import pytest

import parity_cases as PC
import workloads as W
from auditory_amd import capi, runtime, synth
from parity_cases import TOL_F32, TOL_F64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    assert capi.load()._name.endswith("auditory_amd/libauditory_hip.so")
    return torch


@pytest.mark.parametrize("case", PC.CASES, ids=[c[0] for c in PC.CASES])
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_melspec_vs_oracle(orc, torch_cuda, case, cdt):
    PC.case_melspec_vs_oracle(orc, case, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_n512_kernel_variants(orc, torch_cuda, cdt):
    PC.case_n512_variants(orc, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_n400_kernel_variants(orc, torch_cuda, cdt):
    PC.case_n400_variants(orc, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_n2048_kernel_variants(orc, torch_cuda, cdt):
    PC.case_n2048_variants(orc, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_n512_odd_step(orc, torch_cuda, cdt):
    PC.case_n512_odd_step_and_sample_types(orc, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_workgroup_order(orc, torch_cuda, cdt):
    PC.case_workgroup_order(orc, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_interleaved_stereo(orc, torch_cuda, cdt):
    PC.case_interleaved_stereo(orc, cdt, names=("cfg2_16k_n400_nf40", "cfg2_16k_n512_nf40", "cfg1_44k_n1103_nf32",
                                                 "cfg5_44k_n2048_nf128"))


def test_zero_signal_and_empty_batch(orc, torch_cuda):
    PC.case_zero_signal_and_empty_batch(orc)


@pytest.mark.parametrize("name", ["rate_8k_n200_nf32", "rate_24k_n600_nf32", "rate_32k_n800_nf32", "rate_48k_n1200_nf32", "rate_96k_n2400_nf32",
                                  "win20_44k_n882_nf32", "win50_44k_n2205_nf64", "odd_15k_n375_nf32", "mixed_16k_n480_nf32"])
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_smooth_routes(orc, torch_cuda, name, cdt):
    """smooth window lengths (processspeech's parameters at the common sample rates; 44.1 kHz windows with factors of 7) on both
    routes of the any-N kernel and with every accepted number of frames per workgroup"""
    PC.case_smooth_routes(orc, name, cdt)


def test_generic_lds_limits(orc, torch_cuda):
    PC.case_generic_lds_limits(orc)


def test_plan_rejects_unsupported(orc, torch_cuda):
    PC.case_plan_rejects_unsupported(orc)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_gabor_4d_and_2d_vs_oracle(orc, torch_cuda, cdt):
    PC.case_gabor_4d_and_2d_vs_oracle(orc, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_process_fused_item_kernel(orc, torch_cuda, cdt):
    """mel + gabor as ONE launch (workgroup-per-item kernel, the item's mel matrix in LDS) vs the oracle and vs the two launches"""
    PC.case_process_fused_vs_oracle(orc, cdt, PC.TorchMem(), n=7)
    PC.case_process_fused_vs_oracle(orc, cdt, PC.TorchMem(), name="sndenv_16k_n400_nf32", n=5, pools=(8, 4))


def test_sndenv_mirror_reads_like_the_reference(orc, torch_cuda):
    PC.case_sndenv_mirror_reads_like_the_reference(orc)


@pytest.mark.parametrize("name", ["sndenv_16k_n400_nf32", "cfg2_16k_n512_nf40", "cfg1_44k_n1103_nf32", "rate_48k_n1200_nf32",
                                  "rate_8k_n200_nf32"])   # wave kernels; the chirp kernel; smooth lengths in place (F = 2, 8)
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_prev_smooth(orc, torch_cuda, name, cdt):
    PC.case_prev_smooth(orc, name, cdt)


@pytest.mark.parametrize("name", ["sndenv_16k_n400_nf32", "cfg2_16k_n512_nf40", "rate_48k_n1200_nf32", "rate_8k_n200_nf32",
                                  "rate_22k_n551_nf32", "win20_44k_n882_nf32"])   # + in place (F = 2, 8), the chirp kernel at N = 551, radix 7
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_mfcc_tail(orc, torch_cuda, name, cdt):
    PC.case_mfcc_tail(orc, name, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_mfcc_tail_from_stored_tensors(orc, torch_cuda, cdt):
    PC.case_mfcc_tail(orc, "sndenv_16k_n400_nf32", cdt, options={"kernel": 1})     # generic kernel: the unfused tail


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_per_step_api(orc, torch_cuda, cdt):
    PC.case_per_step_api(orc, cdt)


def test_sndenv_resident_signal_staleness(orc, torch_cuda):
    PC.case_sndenv_resident_signal_staleness(orc)


def test_sndenv_mirror_2d_gabor_kwta_layer(orc, torch_cuda):
    PC.case_sndenv_mirror_2d_gabor_kwta_layer(orc)


def test_kwta_vs_oracle(orc, torch_cuda):
    PC.case_kwta_vs_oracle(orc)


def test_kwta_shapes(orc, torch_cuda):
    PC.case_kwta_shapes(orc)


@pytest.mark.parametrize("name,seg_ms", [("cfg2_16k_n400_nf40", None), ("cfg2_16k_n512_nf40", None),
                                         ("cfg5_44k_n2048_nf128", 200.0), ("cfg1_44k_n1103_nf32", None)])
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_input_levels(orc, torch_cuda, name, seg_ms, cdt):
    PC.case_input_levels(orc, name, cdt, seg_ms)


@pytest.mark.parametrize("name,seg_ms", [("cfg2_16k_n400_nf40", None), ("cfg2_16k_n512_nf40", None),
                                         ("cfg5_44k_n2048_nf128", 200.0), ("cfg1_44k_n1103_nf32", None)])
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_mel_logoff_renorm(orc, torch_cuda, name, seg_ms, cdt):
    PC.case_mel_logoff_renorm(orc, name, cdt, seg_ms)


@pytest.mark.parametrize("N", [400, 512, 2048])
def test_random_wave_kernel_configs(orc, torch_cuda, N):
    """48 seeded random geometries per wave kernel (the emulator tier runs the first 8 of each)"""
    for seed in range(48):
        PC.case_random_wave_config(orc, seed, N)


def test_recreated_tone_fixtures_f64(orc, torch_cuda):
    PC.case_recreated_tone_fixtures_f64(orc)


@pytest.mark.parametrize("name,seg_ms", [("cfg2_16k_n512_nf40", None), ("cfg2_16k_n400_nf40", None),
                                         ("cfg5_44k_n2048_nf128", 300.0), ("cfg1_44k_n1103_nf32", None)],
                         ids=["n512", "n400", "n2048", "n1103"])
@pytest.mark.parametrize("cdt", [capi.AUD_F64, capi.AUD_FAST_F32], ids=["f64", "f32"])
def test_input_dtypes_agree(orc, torch_cuda, name, seg_ms, cdt):
    """float32, float64 and int16 PCM samples (sound.go:116-141 normalisation on the device) through every
    kernel family; int16 takes the 4-byte-per-pair route when a frame's pairs are aligned and the guarded
    route when they are not -- the results must not depend on which.  float32 plans see PCM / 0x7FFF rounded to float32 whichever
    way it arrives (int16 == float32 samples, bit for bit); float64 plans see the reference's float64 quotient from int16 and
    from float64 samples alike (int16 == float64 samples, bit for bit), float32 samples being the rounded ones."""
    torch = torch_cuda
    from auditory_amd.batch import BatchProcessor
    oc = W.OracleCfg(orc, name, seg_ms)
    L = oc.full_len()
    sig, pcm = synth.batch(12, 4, L - oc.N // 2, oc.sr, row_len=L)
    plan = W.product_plan(oc, cdt)
    bp = BatchProcessor(plan, "cuda:0")
    items = bp.upload_items(runtime.make_items(np.arange(4) * L, [L] * 4, [0] * 4))
    outs = []
    for t in (torch.from_numpy(sig.astype(np.float32)), torch.from_numpy(sig), torch.from_numpy(pcm)):
        outs.append(bp.melspec(t.cuda().contiguous().view(-1), items, 4).cpu().numpy())
    torch.cuda.synchronize()
    f32 = cdt == capi.AUD_FAST_F32
    twin = 0 if f32 else 1                             # the sample type that must match int16 PCM bit for bit
    assert np.array_equal(outs[twin], outs[2], equal_nan=True)
    # (float32 samples are OTHER numbers than PCM / 0x7FFF -- rounded at 6e-8 relative; a one-bin mel filter 60 dB under the frame's
    # peak moves by a few 1e-6 with them: a sanity bound only, each input's own parity is checked against the oracle below)
    ok, msg = W.close_enough(outs[1], outs[0], 1e-5)
    assert ok, msg
    ref, _, _ = PC.oracle_items(orc, oc, sig, [(r, 0) for r in range(4)])
    ok, msg = W.feature_close(outs[2], ref, cdt, lin_axis=1)
    assert ok, msg
    if not f32:   # the float32 samples are other numbers than PCM / 0x7FFF: against the oracle on exactly those
        ref32, _, _ = PC.oracle_items(orc, oc, sig.astype(np.float32).astype(np.float64), [(r, 0) for r in range(4)])
        ok, msg = W.feature_close(outs[0], ref32, cdt, lin_axis=1)
        assert ok, msg
    assert not (ref[:, :, -1] == 0).all()              # zero tail keeps every frame in bounds
    # the same streams at an odd row pitch: rows 1 and 3 start at odd sample offsets (unaligned pairs)
    Lo = L + 1 if L % 2 == 0 else L + 2
    for arr, want in ((pcm, outs[2]), (sig.astype(np.float32), outs[0])):
        wide = np.zeros((4, Lo), arr.dtype)
        wide[:, :L] = arr
        items_o = bp.upload_items(runtime.make_items(np.arange(4) * Lo, [L] * 4, [0] * 4))
        got = bp.melspec(torch.from_numpy(wide).cuda().contiguous().view(-1), items_o, 4).cpu().numpy()
        assert np.array_equal(got, want, equal_nan=True)
    # and as the channels of two interleaved stereo clips (rows 0 | 1 and 2 | 3): strided work items over one buffer
    for arr, want in ((pcm, outs[2]), (sig.astype(np.float32), outs[0])):
        inter = np.ascontiguousarray(arr.reshape(2, 2, L).transpose(0, 2, 1))        # [clip, sample, channel]
        items_s = bp.upload_items(runtime.make_items((np.arange(4) // 2) * (2 * L) + np.arange(4) % 2, [L] * 4, [0] * 4,
                                                     sig_stride=2))
        got = bp.melspec(torch.from_numpy(inter).cuda().contiguous().view(-1), items_s, 4).cpu().numpy()
        assert np.array_equal(got, want, equal_nan=True)
    plan.close()


@pytest.mark.parametrize("cdt", [capi.AUD_F64, capi.AUD_FAST_F32], ids=["f64", "f32"])
def test_int16_normalisation_exhaustive(orc, torch_cuda, cdt):
    """every int16 value once: the 4-byte-per-pair int16 route (x * RN(1/32767) + one residual step) must give
    what the correctly rounded division gives in the plan's compute type -- numpy's float32 / float64 quotient fed in as
    float32 / float64 samples (sound.go:138: float64(v) / float64(0x7FFF))"""
    torch = torch_cuda
    from auditory_amd.batch import BatchProcessor
    oc = W.OracleCfg(orc, "cfg2_16k_n512_nf40")
    pcm = np.arange(-32768, 32768, dtype=np.int64)
    pcm = np.concatenate([pcm[::2], pcm[1::2][::-1], np.zeros(oc.N, np.int64)]).astype(np.int16)  # not a ramp
    f32 = pcm.astype(np.float32) / np.float32(32767.0) if cdt == capi.AUD_FAST_F32 else pcm.astype(np.float64) / 32767.0
    n_seg = (65536 + oc.sp.stride_samples - 1) // oc.sp.stride_samples
    plan = W.product_plan(oc, cdt)
    bp = BatchProcessor(plan, "cuda:0")
    items = bp.upload_items(runtime.make_items([0] * n_seg, [len(pcm)] * n_seg,
                                               [s * oc.sp.stride_samples for s in range(n_seg)]))
    power = [torch.empty((n_seg, oc.H, oc.T), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    a = bp.melspec(torch.from_numpy(pcm).cuda(), items, n_seg, power=power[0]).cpu().numpy()
    b = bp.melspec(torch.from_numpy(f32).cuda(), items, n_seg, power=power[1]).cpu().numpy()
    torch.cuda.synchronize()
    assert np.array_equal(a, b) and np.array_equal(power[0].cpu().numpy(), power[1].cpu().numpy())
    plan.close()


@pytest.mark.parametrize("cdt", [capi.AUD_F64, capi.AUD_FAST_F32], ids=["f64_strict", "f32_relaxed"])
def test_process_batch_mel_plus_gabor(orc, torch_cuda, cdt):
    """BASELINE configs[3] (N = 512 + agabor.Convolve): fused API, default FilterSet, 4-D [11, 32, 2, 8] pools.  The float64
    plan -- what every default selects -- under the STRICT criterion |d| <= 1e-5 max(1, |ref|) on every element of both tensors
    (and the tighter float64 tolerances); the float32 opt-in under its relaxed one."""
    torch = torch_cuda
    from auditory_amd.batch import BatchProcessor
    oc = W.OracleCfg(orc, "cfg2_16k_n512_nf40")
    L = oc.full_len()
    n = 6
    sig, _ = synth.batch(4, n, 16000, oc.sr, row_len=L)
    k = orc.gabor_to_tensor(W.DEFAULT_GABOR_SPECS, 9, 9)
    g = dict(k=k, stride_x=3, stride_y=3, gain=2.0, py=11, px=32)
    sig32 = sig.astype(np.float32)                      # what the device sees
    rc, ref_mel, ref_g = orc.process_batch(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig32.astype(np.float64).ravel(),
                                           np.arange(n) * L, np.full(n, L), np.zeros(n), gabor=g)
    assert rc == 0
    plan = W.product_plan(oc, cdt, PC.GABOR_DEFAULT)
    bp = BatchProcessor(plan, "cuda:0")
    items = bp.upload_items(runtime.make_items(np.arange(n) * L, [L] * n, [0] * n))
    mel, gab = bp.process(torch.from_numpy(sig.astype(np.float32)).cuda().view(-1), items, n, 11, 32)
    torch.cuda.synchronize()
    if cdt == capi.AUD_F64:
        for what, got, ref in (("mel", mel, ref_mel), ("gabor", gab, ref_g)):
            ok, msg = W.close_enough(got.cpu().numpy(), ref, 1e-5)
            assert ok, "strict criterion, %s: %s" % (what, msg)
    ok, msg = W.feature_close(mel.cpu().numpy(), ref_mel, cdt, lin_axis=1)
    assert ok, "mel " + msg
    ok, msg = W.feature_close(gab.cpu().numpy(), ref_g, cdt, tol_f64=W.TOL_F64_GABOR)
    assert ok, "gabor " + msg
    plan.close()


def test_zeroed_plan_desc_is_the_conforming_plan(orc, torch_cuda):
    """A memset-0 aud_plan_desc with only the geometry and the parameter blocks filled in -- compute_dtype left at its zero
    value, as a C caller's `aud_plan_desc d = {0}` or a Go zero value leaves it -- must give the float64 plan and pass the
    north star's criterion on EVERY element (the reference is float64 throughout: dft.go:42-85, mel.go:120-153)."""
    import ctypes as C
    torch = torch_cuda
    lib = capi.load()
    ctx = runtime.get_ctx(0)
    oc = W.OracleCfg(orc, "cfg2_16k_n400_nf40")
    d = capi.PlanDesc()
    C.memset(C.byref(d), 0, C.sizeof(d))
    d.win_samples, d.step_samples, d.segment_steps, d.border_steps = oc.N, oc.S, oc.T, oc.sp.border_steps
    lib.aud_dft_defaults(C.byref(d.dft))
    fb = capi.MelFBank()
    lib.aud_mel_defaults(C.byref(fb))
    fb.n_filters, fb.lo_hz, fb.hi_hz = oc.nf, oc.m.lo_hz, oc.m.hi_hz
    bins, filt = np.zeros(oc.nf + 2, np.int32), np.zeros((oc.nf, oc.nf + 2))
    assert lib.aud_mel_init_filters(C.byref(fb), oc.N, oc.sr, bins.ctypes.data_as(C.c_void_p), None, filt.ctypes.data_as(C.c_void_p)) == 0
    d.mel = fb
    assert d.compute_dtype == 0 == capi.AUD_F64
    h = C.c_void_p()
    ctx.check(lib.aud_plan_create(ctx.handle, C.byref(d), bins.ctypes.data_as(C.POINTER(C.c_int32)),
                                  filt.ctypes.data_as(C.POINTER(C.c_double)), None, C.byref(h)))
    try:
        assert lib.aud_plan_kernel_name(h).decode() == "w20x10"
        n, L = 8, oc.full_len()
        sig, _ = synth.batch(77, n, 16000, oc.sr, row_len=L)
        sig32 = sig.astype(np.float32)
        dsig = torch.from_numpy(sig32).cuda().view(-1)
        items = torch.from_numpy(np.frombuffer(runtime.make_items(np.arange(n) * L, [L] * n, [0] * n).tobytes(), np.uint8).copy()).cuda()
        mel = torch.from_numpy(np.full((n, oc.nf, oc.T), 3.0, np.float32)).cuda()
        ctx.check(lib.aud_melspec_batch_dev(h, dsig.data_ptr(), capi.AUD_F32, items.data_ptr(), n, mel.data_ptr(), None, None,
                                            torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        ref, _, _ = PC.oracle_items(orc, oc, sig32.astype(np.float64), [(r, 0) for r in range(n)])
        ok, msg = W.close_enough(mel.cpu().numpy(), ref, 1e-5)
        assert ok, "zero-initialised descriptor: " + msg
        ok, msg = W.close_enough(mel.cpu().numpy(), ref, W.TOL_F64)      # and it really is the float64 arithmetic
        assert ok, msg
    finally:
        lib.aud_plan_destroy(h)


@pytest.mark.parametrize("name", ["cfg2_16k_n400_nf40", "cfg2_16k_n512_nf40", "cfg1_44k_n1103_nf32"])
def test_segment_device_resident(orc, torch_cuda, name):
    """aud_segment_batch_dev on tensors in HBM (float32 samples, float64 plan): the same values as the host-staged entry
    gives for the same float32 samples, with and without the optional outputs; w20x10 / w16x16 run the fused tail, the
    generic kernel (N = 1103) the two launches of its parts."""
    torch = torch_cuda
    from auditory_amd.batch import BatchProcessor
    oc = W.OracleCfg(orc, name)
    L = oc.full_len() - 3 * oc.S                  # the last frames run off the end: masked steps (Q7)
    n = 5
    sig, _ = synth.batch(29, n, L, oc.sr)
    sig32 = sig.astype(np.float32)
    plan = W.product_plan(oc, capi.AUD_F64, mfcc_coefs=13)
    try:
        host_items = runtime.make_items(np.arange(n) * L, [L] * n, [0] * n)
        want = plan.melspec_mfcc_host(sig32.astype(np.float64).ravel(), host_items)
        bp = BatchProcessor(plan, "cuda:0")
        items = bp.upload_items(host_items)
        dsig = torch.from_numpy(sig32).cuda().view(-1)
        full = bp.segment(dsig, items, n)
        lean = bp.segment(dsig, items, n, want_spectrum=False, deltas=False)
        torch.cuda.synchronize()
        for key in ("mel", "power", "log_power", "mfcc", "deltas", "delta_deltas", "energy"):
            assert np.array_equal(full[key].cpu().numpy().astype(np.float64), want[key], equal_nan=True), key
        for key in ("mel", "mfcc", "energy"):
            assert np.array_equal(lean[key].cpu().numpy(), full[key].cpu().numpy(), equal_nan=True), key
        # both entries route through aud_segment_batch_dev, so the above is self-consistency: the device-resident result
        # against the ORACLE too, item by item, incl. the masked final steps (the streams end three steps early: Q7)
        fused = plan.kernel_name in ("w16x16", "w20x10")
        tols = dict(mfcc=4e-6, deltas=6e-6, delta_deltas=5e-5, energy=3e-7) if fused else \
            dict(mfcc=8e-6, deltas=4e-5, delta_deltas=2e-4, energy=3e-7)
        x64 = sig32.astype(np.float64)
        for r in range(n):
            o = orc.process_segment_mfcc(oc.sp, oc.d, oc.m, oc.bins, oc.filt, x64[r], segment=0)
            assert o["done"] < oc.T                                    # some final steps really are masked
            ok, msg = W.feature_close(full["mel"][r].cpu().numpy(), o["mel_seg"], capi.AUD_F64, lin_axis=0)
            assert ok, "mel item %d: %s" % (r, msg)
            for key, tol in tols.items():
                ok, msg = W.close_enough(full[key][r].cpu().numpy(), o[key], tol)
                assert ok, "%s item %d vs oracle: %s" % (key, r, msg)
            assert np.all(full["mfcc"][r].cpu().numpy()[1:, o["done"]:] == 0)
            assert np.all(full["mel"][r].cpu().numpy()[:, o["done"]:] == 0)
    finally:
        plan.close()


@pytest.mark.parametrize("cdt", [capi.AUD_F64, capi.AUD_FAST_F32], ids=["f64", "f32"])
def test_process_then_kwta_device_resident(orc, torch_cuda, cdt, n=6):
    """SndEnv.ApplyGabor with Kwta.On (sndenv.go:481-497) on device-resident tensors: mel + gabor, then the
    k-WTA stage on the gabor tensor where it lies in HBM; bit-exact against the oracle run on that tensor.  The float64 plan's
    gabor tensor is also held to the strict criterion against the oracle's own mel + Convolve."""
    torch = torch_cuda
    from auditory_amd import kwta
    from auditory_amd.batch import BatchProcessor
    oc = W.OracleCfg(orc, "cfg2_16k_n512_nf40")
    L = oc.full_len()
    sig, _ = synth.batch(9, n, 16000, oc.sr, row_len=L)
    plan = W.product_plan(oc, cdt, PC.GABOR_DEFAULT)
    bp = BatchProcessor(plan, "cuda:0")
    items = bp.upload_items(runtime.make_items(np.arange(n) * L, [L] * n, [0] * n))
    dsig = torch.from_numpy(sig.astype(np.float32)).cuda().view(-1)
    _, gab = bp.process(dsig, items, n, 11, 32)
    if cdt == capi.AUD_F64:
        gk = orc.gabor_to_tensor(W.DEFAULT_GABOR_SPECS, 9, 9)
        rc, _, ref_g = orc.process_batch(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig.astype(np.float32).astype(np.float64).ravel(),
                                         np.arange(n) * L, np.full(n, L), np.zeros(n),
                                         gabor=dict(k=gk, stride_x=3, stride_y=3, gain=2.0, py=11, px=32))
        assert rc == 0
        ok, msg = W.close_enough(gab.cpu().numpy(), ref_g, 1e-5)
        assert ok, "strict criterion, gabor: " + msg
    k, ko = PC._kwta_pair(orc)
    _, gab2, act2 = bp.process_sndenv(dsig, items, n, 11, 32, k)     # the one-call form
    act = torch.empty(gab.shape, dtype=torch.float32, device=gab.device)
    cyc = torch.zeros(n, dtype=torch.int32, device=gab.device)
    state = torch.zeros((n, 11 * 32, 2), dtype=torch.float32, device=gab.device)
    stream = torch.cuda.current_stream().cuda_stream
    kwta.kwta_batch_dev(k, gab, act, pool=True, state=state, cycles=cyc, stream=stream)
    torch.cuda.synchronize()
    raw = gab.cpu().numpy()
    st_o = np.zeros((n, 11 * 32, 2), np.float32)
    for i in range(n):
        ref, c = orc.kwta_pool(ko, raw[i], st_o[i])
        assert np.array_equal(act[i].cpu().numpy(), ref)
        assert int(cyc[i]) == c
    assert np.array_equal(state.cpu().numpy(), st_o)
    # the gabor tensor was only read
    assert np.array_equal(gab.cpu().numpy(), raw)
    assert np.array_equal(gab2.cpu().numpy(), raw) and np.array_equal(act2.cpu().numpy(), act.cpu().numpy())
    plan.close()


def _full_size_properties(orc, torch, name, B, dur, cdt, spot_step, parseval_steps, gain_tol, shift_tol=0.0):
    """A BASELINE batch at full size through size-independent properties; the oracle only spot-checks a few streams."""
    from auditory_amd.batch import BatchProcessor
    oc = W.OracleCfg(orc, name)
    L = oc.full_len()
    Lp = (L + 63) // 64 * 64
    sig, _ = synth.batch(2, B, dur, oc.sr, row_len=Lp)
    plan = W.product_plan(oc, cdt)
    bp = BatchProcessor(plan, "cuda:0")
    dsig = torch.from_numpy(sig.astype(np.float32)).cuda().view(-1)
    items = bp.upload_items(runtime.make_items(np.arange(B) * Lp, [Lp] * B, [0] * B))
    power = torch.empty((B, oc.H, oc.T), dtype=torch.float32, device=dsig.device)
    mel = bp.melspec(dsig, items, B, power=power).cpu().numpy()
    pw = power.cpu().numpy().astype(np.float64)
    x32 = sig.astype(np.float32).astype(np.float64)     # what the device saw
    # (a) spot parity on streams spread over the batch
    idx = sorted(set(list(range(0, B, spot_step)) + [B - 1]))
    ref = np.stack([orc.process_segment(oc.sp, oc.d, oc.m, oc.bins, oc.filt, x32[i])["mel_seg"] for i in idx])
    ok, msg = W.feature_close(mel[idx], ref, cdt, lin_axis=1)
    assert ok, msg
    # (b) Parseval on every frame: sum_k c_k P[k] = N * sum x^2
    c = np.full(oc.H, 2.0); c[0] = 1.0
    if oc.N % 2 == 0:
        c[-1] = 1.0                                     # (the Nyquist bin exists for even N only)
    for s in parseval_steps:
        st = oc.S * (s - 2)
        fr = np.zeros((B, oc.N))
        a = max(st, 0)
        fr[:, a - st:] = x32[:, a:st + oc.N]
        lhs = (pw[:, :, s] * c).sum(axis=1)
        rhs = oc.N * (fr ** 2).sum(axis=1)
        assert np.abs(lhs - rhs).max() <= 2e-6 * rhs.max()
    # (c) batch-position invariance: reversed item order gives the reversed result, bit for bit
    items_r = bp.upload_items(runtime.make_items(np.arange(B)[::-1] * Lp, [Lp] * B, [0] * B))
    mel_r = bp.melspec(dsig, items_r, B).cpu().numpy()
    assert np.array_equal(mel_r[::-1], mel, equal_nan=True)
    # (d) time shift: start0 = +S moves every column one step left
    items_s = bp.upload_items(runtime.make_items(np.arange(B) * Lp, [Lp] * B, [oc.S] * B))
    mel_s = bp.melspec(dsig, items_s, B).cpu().numpy()
    if shift_tol == 0.0:
        assert np.array_equal(mel_s[:, :, :-1], mel[:, :, 1:], equal_nan=True)
    else:   # (the chirp kernel transforms PAIRS of frames: a frame's new partner changes its last bits, 2^-53 of its own peak)
        assert np.array_equal(np.isnan(mel_s[:, :, :-1]), np.isnan(mel[:, :, 1:]))
        assert np.nanmax(np.abs(mel_s[:, :, :-1] - mel[:, :, 1:])) <= shift_tol
    fin = ~np.isnan(mel_s[:, :, -1])
    assert np.all(mel_s[:, :, -1][fin] == 0)            # last frame now runs off the end
    # (e) gain: 2x input -> power x4 exactly (power of two), mel + ln 4
    mel2 = bp.melspec(dsig * 2.0, items, B).cpu().numpy()
    ok_cells = ~np.isnan(mel)
    silent = ok_cells & (mel == -10.0)                  # frames inside the zero tail: power exactly 0 -> LogMin (Q2)
    assert np.array_equal(mel2[silent], mel[silent])
    live = ok_cells & ~silent
    assert np.abs(mel2 - (mel + np.log(4.0)))[live].max() <= gain_tol * np.abs(mel[live]).max() + 1e-6
    plan.close()
    return mel


@pytest.mark.parametrize("cdt", [capi.AUD_F64, capi.AUD_FAST_F32], ids=["f64_strict", "f32_relaxed"])
def test_full_size_properties_cfg2(orc, torch_cuda, cdt, B=256):
    """BASELINE configs[1] as worded at full size (B = 256 x 1 s @16 kHz, N = 512, 40 mel): the float64 plan every default
    selects (spot streams under the float64 tolerance, i.e. well inside the strict 1e-5), and the float32 opt-in"""
    mel = _full_size_properties(orc, torch_cuda, "cfg2_16k_n512_nf40", B, 16000, cdt, 17, (0, 1, 2, 50, 103),
                                1e-6 if cdt == capi.AUD_F64 else 2e-6)
    assert (mel == -10.0).any()


def test_full_size_properties_metric_config_f64(orc, torch_cuda, B=256):
    """the metric's own parameter set (N = 400) at full size, float64 plan: what bench.py's headline runs"""
    _full_size_properties(orc, torch_cuda, "cfg2_16k_n400_nf40", B, 16000, capi.AUD_F64, 17, (0, 1, 2, 50, 103), 1e-6)


def test_full_size_properties_cfg5(orc, torch_cuda, B=320):
    """BASELINE configs[4] (44.1 kHz, 5 s streams, N = 2048, 128 mel with its NaN row) at 320 streams = 282 MB of input -- past
    the 256 MB Infinity Cache; the full 1280-stream batch (1.13 GB) runs in bench.py's also.cfg5 with 40 streams checked
    against the oracle (the float64 host copies this test keeps for its Parseval and invariance checks bound it here)"""
    mel = _full_size_properties(orc, torch_cuda, "cfg5_44k_n2048_nf128", B, 5 * 44100, capi.AUD_F64, 31, (0, 2, 250, 503), 1e-6)
    assert np.isnan(mel[:, 0, :]).all()                 # filter 0 is a degenerate triangle (Q3)


def test_full_size_properties_cfg1_chirp(orc, torch_cuda, B=256):
    """BASELINE configs[0]'s parameters at bench size (256 segments of 100 ms at 44.1 kHz, N = 1103, 32 mel): the chirp kernel
    through Parseval on every frame, batch-order and time-shift invariance, power-of-two gain -- the oracle spot-checks"""
    oc = W.OracleCfg(orc, "cfg1_44k_n1103_nf32")
    _full_size_properties(orc, torch_cuda, "cfg1_44k_n1103_nf32", B, oc.full_len(), capi.AUD_F64, 17, (0, 1, 2, 7, 13), 1e-6,
                          shift_tol=1e-6)


@pytest.mark.parametrize("name", ["rate_48k_n1200_nf32", "rate_8k_n200_nf32", "win50_44k_n2205_nf64"])
def test_full_size_properties_smooth_in_place(orc, torch_cuda, name, B=256):
    """processspeech's parameters on 48 kHz / 8 kHz audio and a 50 ms window at 44.1 kHz (radix 7) at bench size: the any-N
    kernel's in-place route (two / eight / one frame per workgroup) through the same properties, time shift bit for bit"""
    oc = W.OracleCfg(orc, name)
    _full_size_properties(orc, torch_cuda, name, B, oc.full_len(), capi.AUD_F64, 17, (0, 1, 2, 7, 13), 1e-6)


def test_rccl_allgather_single_rank(torch_cuda):
    """aud_comm_* / aud_allgather_dev (the non-Python route to the path's one collective): with a
    1-rank communicator the gather must hand the slab back unchanged."""
    import ctypes as C
    torch = torch_cuda
    lib = capi.load()
    ctx = runtime.get_ctx(0)
    uid = C.create_string_buffer(128)
    assert lib.aud_comm_unique_id(uid) == 0
    ctx.check(lib.aud_comm_init(ctx.handle, 1, 0, uid))
    try:
        send = torch.arange(40 * 104 * 3, dtype=torch.float32, device="cuda")
        recv = torch.zeros_like(send)
        ctx.check(lib.aud_allgather_dev(ctx.handle, send.data_ptr(), recv.data_ptr(), send.numel(),
                                        torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
    finally:
        lib.aud_comm_destroy(ctx.handle)


def test_direct_allgather_two_processes_one_gpu(torch_cuda, tmp_path):
    """aud_gather_* with REAL inter-process handles: two processes on this one GPU exchange their receive buffers' handles
    through files, push their slabs into each other (hipIpcOpenMemHandle + device-to-device copies on the per-peer
    streams, inside a captured hipGraph on one of them) and check both buffers"""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gather_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = _join_workers(procs, 300)
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and "GATHER-OK" in o, (o[-1000:], e[-3000:])
    assert "GATHER-TIMEOUT-OK" in outs[0][0]        # a peer that never arrives: bounded, counted, NaN-filled, sticky (rank 0's view)
    print(outs[0][0].strip().splitlines()[-2:])


def _join_workers(procs, timeout):
    """communicate() with every worker under ONE deadline; whatever is still alive when it passes (a peer stuck waiting
    for the other's handle file, say) is killed -- exactly these pids -- so that nothing stays on the GPU after a failure"""
    import subprocess
    import time
    deadline = time.monotonic() + timeout
    outs = []
    try:
        for p in procs:
            try:
                outs.append(p.communicate(timeout=max(1.0, deadline - time.monotonic())))
            except subprocess.TimeoutExpired:
                p.kill()
                o, e = p.communicate()
                outs.append((o, (e or "") + "\n[killed after %d s]" % timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    return outs


def test_two_ranks_one_gpu_bench_flow(torch_cuda, tmp_path):
    """bench.py's N > 1 flow as TWO processes on this box's one GPU (tools/two_ranks_one_gpu.sh made a test): gloo as the
    control plane, the library's direct device-to-device reassembly as the collective (RCCL refuses two ranks on one
    device).  Timing means nothing here; what it checks on hardware is the two-rank control flow: shards, inter-process
    buffer export, graph-captured pushes, rank 0's strict parity check of BOTH ranks' blocks, and the line's N > 1 keys."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                                       "--gather-mode", "direct", "--no-cpu-baseline", "--no-stream-read", "--steps", "20",
                                       "--warmup", "5", "--min-seconds", "0.05", "--cfg3-total", "1024"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root))
    outs = _join_workers(procs, 420)
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (o[-1000:], e[-3000:])
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["rccl_ranks"] == 2
    assert line["gathered_shape"] == [1024, 40, 104] and "direct pattern" in line["collective"]
    assert line["parity"]["pass"] and line["parity"]["n_past_1e-5"] == 0 and "every rank" in line["parity"]["checked"]
    assert line["no_collective"]["parity"]["pass"]


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_resident_signal(orc, torch_cuda, cdt):
    PC.case_resident_signal(orc, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_gabor_geometry_fuzz(orc, torch_cuda, cdt):
    """64 seeded random Convolve geometries (matrix shape, taps, strides, filter counts off the quad grid, NaN cells, rank-4 pools
    and rank-2 outputs in both orders, shapes the Go code rejects) through both gabor kernels against the oracle"""
    for seed in range(64):
        PC.case_gabor_fuzz(orc, seed, cdt)


# window lengths of every kind for the any-N claim on hardware: powers of two, 3/5-smooth, primes and prime x small (Bluestein:
# 211, 514 = 2 x 257, 1009, 1103, 2027, 2206 = 2 x 1103), prime squares (169, 289: the O(p) radix pass), long smooth ones
ANY_N = [6, 25, 49, 96, 121, 169, 200, 211, 243, 289, 343, 375, 441, 480, 514, 625, 729, 882, 1000, 1009, 1024, 1103, 1200, 1323,
         1500, 1764, 2000, 2027, 2048, 2206, 2400]


def test_random_any_n_configs(orc, torch_cuda):
    """96 seeded random parameter sets (window length, step, segment, border, mel table, compute type) through whatever kernel
    the plan selects -- the emulator tier runs 48 with lengths up to 1024"""
    seen = set()
    for seed in range(96):
        try:
            seen.add(PC.case_random_any_n(orc, seed, ANY_N).split()[5])
        except pytest.skip.Exception:
            pass
    assert "generic" in seen


@pytest.mark.gpu
@pytest.mark.parametrize("N,cdt,kind,quirks", [(1103, capi.AUD_F64, "float", True), (1103, capi.AUD_F64, "int16", False),
                                               (1027, capi.AUD_F64, "float", False), (1151, capi.AUD_F64, "float", True),
                                               (1025, capi.AUD_F64, "int16", False), (1131, capi.AUD_F64, "float", True),
                                               (551, capi.AUD_F64, "float", True), (1014, capi.AUD_F64, "int16", False),
                                               (1001, capi.AUD_F64, "float", False), (1126, capi.AUD_F64, "float", True),
                                               (513, capi.AUD_F64, "int16", False), (1150, capi.AUD_F64, "float", True), (507, capi.AUD_F64, "float", True)],
                         ids=["n1103_f64_quirks", "n1103_f64_i16", "n1027_f64", "n1151_f64_quirks", "n1025_f64_i16",
                              "n1131_f64_quirks", "n551_f64_quirks", "n1014_f64_i16", "n1001_f64", "n1126_f64_quirks", "n513_f64_i16",
                              "n1150_f64_quirks", "n507_f64_quirks"])
def test_chirp_kernel(orc, torch_cuda, N, cdt, kind, quirks):
    """the fixed-geometry chirp kernel of L = 2304 (melspec_chirp.hip: the reference's N = 1103 and the other odd window lengths
    it serves) against the oracle and against the any-N route it replaces, three seeds each"""
    for seed in range(3):
        PC.case_chirp_kernel(orc, N, cdt, seed=seed, sig_kind=kind, quirks=quirks)


@pytest.mark.gpu
@pytest.mark.parametrize("N,cdt,kind", [(5123, capi.AUD_F64, "float"), (10243, capi.AUD_F32, "int16"), (12000, capi.AUD_F64, "float"),
                                        (10241, capi.AUD_F64, "int16")],
                         ids=["n5123_f64", "n10243_f32_i16", "n12000_f64", "n10241_f64_i16"])
def test_direct_kernel(orc, torch_cuda, N, cdt, kind):
    """window lengths no LDS-resident transform serves run the O(N H) kernel (melspec_direct.hip) instead of being refused"""
    PC.case_direct_kernel(orc, N, cdt, sig_kind=kind)


@pytest.mark.gpu
@pytest.mark.parametrize("sr", [16000, 44100])
def test_speech_like_sndenv(orc, torch_cuda, tmp_path, sr):
    """BASELINE configs[0] as worded, on hardware: SURVEY 8d's cfg-1 WAV (3 s of speech-like audio, written by the test)
    through Sound.Load -> ToTensor -> Init -> ProcessSegment x SegCnt -> ApplyGabor with processspeech's parameters,
    float64 plan, every one of the 30 segments against the oracle under the strict criterion -- at 16 kHz (w20x10 with the
    fused tail) and at the shipped WAVs' 44.1 kHz (N = 1103: Bluestein in place)."""
    rep = {}
    PC.case_speech_like_sndenv(orc, sr, tmp_path, None, rep)
    assert rep["seg_cnt"] == 30 and rep["zero_frames"] >= 60
    print("speech-like %d Hz: worst scaled errors %s; %d all-zero frames exact" % (
        sr, {k: "%.2g" % v for k, v in rep["worst"].items()}, rep["zero_frames"]))
