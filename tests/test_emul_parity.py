"""CPU-only coverage of the HIP kernel sources themselves.

The product's .hip files are compiled unchanged against a host stand-in for the HIP runtime
(tests/emul/, one OS thread per GPU thread, real barriers) and driven through the same C ABI and
the same parity cases as on the GPU.  This is how indexing, barrier placement and arithmetic are
checked when no GPU is at hand, and how the kernels get AddressSanitizer / UBSan / ThreadSanitizer
coverage (GPU sanitizers are not available on this pool).  It is test infrastructure: the shipped
library has no CPU path (tests/test_capi_host.py::test_init_without_gpu_fails_loudly)."""
import os
import subprocess
import sys

import pytest

import parity_cases as PC
from auditory_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emul"))
import backend  # noqa: E402
import build_emul  # noqa: E402


@pytest.fixture(scope="module")
def emu():
    with backend.emulated("plain") as lib:
        yield lib


EMU_CASES = [c for c in PC.CASES if c[0] != "cfg5_44k_n2048_nf128"]


@pytest.mark.parametrize("case", EMU_CASES, ids=[c[0] for c in EMU_CASES])
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_melspec_vs_oracle(orc, emu, case, cdt):
    PC.case_melspec_vs_oracle(orc, case, cdt)


@pytest.mark.parametrize("N,cdt,kind,quirks", [(1103, capi.AUD_F64, "float", True), (1103, capi.AUD_F64, "int16", False),
                                               (1027, capi.AUD_F64, "float", False), (1151, capi.AUD_F64, "float", True),
                                               (1025, capi.AUD_F64, "int16", False), (551, capi.AUD_F64, "float", True),
                                               (1014, capi.AUD_F64, "int16", False), (1001, capi.AUD_F64, "float", False), (507, capi.AUD_F64, "float", True)],
                         ids=["n1103_f64_quirks", "n1103_f64_i16", "n1027_f64", "n1151_f64_quirks", "n1025_f64_i16", "n551_f64_quirks",
                              "n1014_f64_i16", "n1001_f64", "n507_f64_quirks"])
def test_emul_chirp_kernel(orc, emu, N, cdt, kind, quirks):
    """the fixed-geometry chirp kernel of L = 2304 (melspec_chirp.hip) vs the oracle and vs the any-N route, at both ends of the
    window lengths it serves"""
    PC.case_chirp_kernel(orc, N, cdt, seed=N % 7, sig_kind=kind, quirks=quirks)


@pytest.mark.parametrize("N,cdt,kind", [(5123, capi.AUD_F64, "float"), (10243, capi.AUD_F32, "int16"), (12000, capi.AUD_F64, "float")],
                         ids=["n5123_f64", "n10243_f32_i16", "n12000_f64"])
def test_emul_direct_kernel(orc, emu, N, cdt, kind):
    """window lengths no LDS-resident transform serves run the O(N H) kernel (melspec_direct.hip) instead of being refused"""
    PC.case_direct_kernel(orc, N, cdt, sig_kind=kind)


@pytest.mark.parametrize("name,cdt", [("rate_8k_n200_nf32", capi.AUD_F64), ("win20_44k_n882_nf32", capi.AUD_F64), ("odd_15k_n375_nf32", capi.AUD_F32)],
                         ids=["n200_f64", "n882_f64", "n375_f32"])
def test_emul_smooth_routes(orc, emu, name, cdt):
    PC.case_smooth_routes(orc, name, cdt)


def test_emul_melspec_nan_row_config(orc, emu):
    # cfg 5 parameters (N=2048, 128 mel, NaN row) on a short segment to keep the thread count sane
    PC.case_melspec_vs_oracle(orc, ("cfg5_44k_n2048_nf128", 0.25, 1, [0]), capi.AUD_F32, seg_ms=200.0)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_n512_kernel_variants(orc, emu, cdt):
    PC.case_n512_variants(orc, cdt, seg_ms=200.0, dur=0.4, rows=2, segs=(0, 1))


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_n400_kernel_variants(orc, emu, cdt):
    PC.case_n400_variants(orc, cdt, seg_ms=200.0, dur=0.45, rows=2, segs=(0, 1, 2))


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_n2048_kernel_variants(orc, emu, cdt):
    PC.case_n2048_variants(orc, cdt, seg_ms=100.0, dur=0.25, rows=1)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_n512_odd_step(orc, emu, cdt):
    PC.case_n512_odd_step_and_sample_types(orc, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_interleaved_stereo(orc, emu, cdt):
    PC.case_interleaved_stereo(orc, cdt, seg_ms=100.0)


def test_emul_zero_signal_and_empty_batch(orc, emu):
    PC.case_zero_signal_and_empty_batch(orc)


def test_emul_generic_lds_limits(orc, emu):
    # (the two 64 KB-exact plans are created here; their transforms run on the GPU tier -- 4096-point stages are slow to emulate)
    PC.case_generic_lds_limits(orc, run=())


def test_emul_plan_rejects_unsupported(orc, emu):
    PC.case_plan_rejects_unsupported(orc)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_gabor_4d_and_2d_vs_oracle(orc, emu, cdt):
    PC.case_gabor_4d_and_2d_vs_oracle(orc, cdt)


def test_emul_sndenv_mirror(orc, emu):
    PC.case_sndenv_mirror_reads_like_the_reference(orc)


@pytest.mark.parametrize("name,seg_ms", [("cfg2_16k_n400_nf40", None), ("cfg2_16k_n512_nf40", None),
                                         ("cfg5_44k_n2048_nf128", 200.0), ("odd_15k_n375_nf32", None)])   # (generic kernel: a small N here, N = 1103 on the GPU)
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_input_levels(orc, emu, name, seg_ms, cdt):
    PC.case_input_levels(orc, name, cdt, seg_ms)


@pytest.mark.parametrize("name,seg_ms", [("cfg2_16k_n400_nf40", None), ("cfg2_16k_n512_nf40", None),
                                         ("cfg5_44k_n2048_nf128", 200.0), ("odd_15k_n375_nf32", None)])   # (generic kernel: a small N here, N = 1103 on the GPU)
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_mel_logoff_renorm(orc, emu, name, seg_ms, cdt):
    PC.case_mel_logoff_renorm(orc, name, cdt, seg_ms)


def test_emul_recreated_tone_fixtures_f64(orc, emu):
    PC.case_recreated_tone_fixtures_f64(orc)


@pytest.mark.parametrize("name", ["sndenv_16k_n400_nf32", "cfg2_16k_n512_nf40", "cfg1_44k_n1103_nf32", "rate_48k_n1200_nf32",
                                  "rate_8k_n200_nf32"])   # wave kernels; the chirp kernel; smooth lengths in place (F = 2, 8)
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_prev_smooth(orc, emu, name, cdt):
    PC.case_prev_smooth(orc, name, cdt)


@pytest.mark.parametrize("name", ["sndenv_16k_n400_nf32", "cfg2_16k_n512_nf40", "rate_48k_n1200_nf32", "rate_8k_n200_nf32",
                                  "rate_22k_n551_nf32", "win20_44k_n882_nf32"])   # + in place (F = 2, 8), the chirp kernel at N = 551, radix 7
@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_mfcc_tail(orc, emu, name, cdt):
    PC.case_mfcc_tail(orc, name, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_mfcc_tail_from_stored_tensors(orc, emu, cdt):
    PC.case_mfcc_tail(orc, "sndenv_16k_n400_nf32", cdt, options={"kernel": 1, "fused_tail": 0})     # the tail on the stored tensors
    PC.case_mfcc_tail(orc, "sndenv_16k_n400_nf32", cdt, options={"kernel": 1})                      # the any-N kernel carries it (round 6)
    PC.case_mfcc_tail(orc, "cfg1_44k_n1103_nf32", cdt)                                             # ... on the Bluestein route too


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_per_step_api(orc, emu, cdt):
    PC.case_per_step_api(orc, cdt)


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_reference_wav(orc, emu, cdt):
    PC.case_reference_wav(orc, cdt)


def _sanitizer_run(variant, which, timeout=900):
    build_emul.build(variant)
    rt = {"asan": "libasan.so", "tsan": "libtsan.so"}[variant]
    pre = subprocess.check_output(["gcc", "-print-file-name=" + rt]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=pre, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               TSAN_OPTIONS="halt_on_error=1:report_signal_unsafe=0", OMP_NUM_THREADS="1",
               OPENBLAS_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "emul", "drive.py"), variant, which],
                       capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0 and "DRIVE-OK" in r.stdout, (r.stdout[-2000:] + "\n" + r.stderr[-6000:])


def test_emul_kernels_under_asan_ubsan():
    _sanitizer_run("asan", "quick")


def test_emul_kernels_under_tsan():
    _sanitizer_run("tsan", "quick")


def test_emul_kwta_vs_oracle(orc, emu):
    PC.case_kwta_vs_oracle(orc, quick=True)


def test_emul_kwta_shapes(orc, emu):
    PC.case_kwta_shapes(orc)


def test_emul_sndenv_resident_signal_staleness(orc, emu):
    PC.case_sndenv_resident_signal_staleness(orc)


def test_emul_sndenv_mirror_2d_gabor_kwta_layer(orc, emu):
    PC.case_sndenv_mirror_2d_gabor_kwta_layer(orc)


def test_emul_workgroup_order(orc, emu):
    PC.case_workgroup_order(orc, capi.AUD_F32, with_n2048=False)   # (one wave per frame: slow to emulate; GPU tier)


def test_emul_plan_info(orc, emu):
    """aud_plan_get_info: the launch facts of the kernel a plan selected, per family; unknown names are AUD_EINVAL"""
    import workloads as W
    expect = {"cfg2_16k_n400_nf40": ("w20x10", 6, 4), "cfg2_16k_n512_nf40": ("w16x16", 4, 4),
              "cfg5_44k_n2048_nf128": ("w64x16", 1, 12), "cfg1_44k_n1103_nf32": ("chirp2304", 0, 4)}
    for name, (fam, fpw, waves) in expect.items():
        oc = W.OracleCfg(orc, name, 100.0)
        plan = W.product_plan(oc, capi.AUD_F64)
        try:
            assert plan.kernel_name == fam
            assert plan.info("frames_per_wave") == fpw and plan.info("waves_per_wg") == waves
            if fam == "chirp2304":  # the prime window length takes the Bluestein route: L = the cheapest 2-3-5-smooth length
                assert plan.info("bluestein_L") == 2304 and plan.info("chirp_kernel") == 1   # >= 2 N - 1 (16 x 16 x 3 x 3)
                assert plan.info("lds_bytes") == 16 * 153 * 16 + 32 and plan.info("wgs_per_cu") == 4
                plan.set_option("chirp_kernel", 0)   # the any-N route of the same plan
                assert plan.kernel_name == "generic" and plan.info("bluestein_L") == 2304 and plan.info("lds_bytes") == 0
                plan.set_option("chirp_kernel", 1)
                assert plan.kernel_name == "chirp2304"
            else:
                assert 0 < plan.info("lds_bytes") <= 160 * 1024 and plan.info("wgs_per_cu") >= 1
                assert plan.info("bluestein_L") == 0
            with pytest.raises(capi.AuditoryError):
                plan.info("no_such_fact")
            plan.set_option("kernel", 1)          # the generic kernel of a factorable length: no Bluestein
            if fam != "chirp2304":
                assert plan.kernel_name == "generic" and plan.info("bluestein_L") == 0
        finally:
            plan.close()


def test_emul_segment_prev_smooth_power_is_optional(orc, emu):
    """aud_segment_batch_dev with dft.PrevSmooth != 0 and power == NULL: the scan's PowerSegment comes out of the
    workspace, the results equal the ones of the call that keeps a power buffer.  (The emulator's "device" pointers are
    host pointers, so the _dev entry is driven with numpy arrays here.)"""
    import numpy as np
    import workloads as W
    from auditory_amd import synth
    oc = W.OracleCfg(orc, "sndenv_16k_n400_nf32")
    L = int(0.3 * oc.sr)
    sig, _ = synth.batch(31, 2, L, oc.sr)
    items = PC.make_items(oc, L, [(0, 0), (1, 1)])
    plan = W.product_plan(oc, capi.AUD_F64, mfcc_coefs=13, dft_override=(0.35, 0.65))
    try:
        want = plan.melspec_mfcc_host(sig.ravel(), items)
        n, T = len(items), oc.T
        ws_bytes = plan.segment_workspace_bytes(n)
        assert ws_bytes >= 2 * n * oc.H * T * 4           # LogPowerSegment + PowerSegment
        ws = np.zeros(ws_bytes + 16, np.uint8)
        wsp = (ws.ctypes.data + 15) & ~15
        flat = np.ascontiguousarray(sig.ravel())
        out = dict(mel=np.zeros((n, oc.nf, T), np.float32), mfcc=np.zeros((n, 13, T), np.float32), energy=np.zeros((n, T), np.float32))
        plan.segment_dev(flat.ctypes.data, capi.AUD_F64, items.ctypes.data, n, out["mel"].ctypes.data, 0, 0, out["mfcc"].ctypes.data,
                         0, 0, out["energy"].ctypes.data, wsp, ws_bytes)
        for key in out:
            assert np.array_equal(out[key].astype(np.float64), want[key], equal_nan=True), key
    finally:
        plan.close()


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_process_fused_item_kernel(orc, emu, cdt):
    """mel + gabor as one launch (workgroup-per-item kernel) on the metric's parameters, and on a short segment whose gabor
    windows wrap into the next mel row (flat offsets)"""
    PC.case_process_fused_vs_oracle(orc, cdt, PC.HostMem(), n=2)
    PC.case_process_fused_vs_oracle(orc, cdt, PC.HostMem(), name="sndenv_16k_n400_nf32", n=2, pools=(8, 4))


@pytest.mark.parametrize("cdt", [capi.AUD_F32, capi.AUD_F64], ids=["f32", "f64"])
def test_emul_resident_signal(orc, emu, cdt):
    PC.case_resident_signal(orc, cdt)


def test_emul_input_levels_bluestein_pair(orc, emu):
    """N = 1103 (prime): two real frames ride one Bluestein transform, each behind its own power-of-two scale -- a silent
    frame beside a loud one must keep its exactly-zero spectrum (LogMin rule), rows from 1e-150 to 2^60 their own floor"""
    PC.case_input_levels(orc, "cfg1_44k_n1103_nf32", capi.AUD_F64, quick=True)


def test_emul_host_results_chunked_copy(orc, emu):
    """The host entry points fetch their float32 results in up to eight pinned chunks, widening each while the next is in
    flight (capi_internal.h fetch_widened): a call large enough to be chunked (>= 2^18 values, three output tensors whose
    boundaries fall inside chunks) must give exactly what per-item calls (one chunk each) give."""
    import numpy as np
    import workloads as W
    from auditory_amd import synth
    oc = W.OracleCfg(orc, "sndenv_16k_n400_nf32")
    n, L = 48, int(0.3 * oc.sr)
    sig, _ = synth.batch(47, n, L, oc.sr)
    segs = [(r, r % 2) for r in range(n)]
    items = PC.make_items(oc, L, segs)
    plan = W.product_plan(oc, capi.AUD_F64)
    try:
        assert n * (oc.nf + 2 * oc.H) * oc.T >= 1 << 18
        mel, pw, lp = plan.melspec_host(sig.ravel(), items, True, True)
        for i in (0, 1, 17, n - 1):
            m1, p1, l1 = plan.melspec_host(sig.ravel(), items[i:i + 1], True, True)
            assert np.array_equal(mel[i], m1[0]) and np.array_equal(pw[i], p1[0]) and np.array_equal(lp[i], l1[0]), i
        mel2, pw2, _ = plan.melspec_host(sig.ravel(), items, True, False)      # a skipped part in the middle of the layout
        assert np.array_equal(mel2, mel) and np.array_equal(pw2, pw)
    finally:
        plan.close()


@pytest.mark.parametrize("sr", [16000, 44100])
def test_emul_speech_like_sndenv(orc, emu, tmp_path, sr):
    """configs[0] as worded on SURVEY 8d's cfg-1 input: all 30 segments at both rates"""
    rep = {}
    PC.case_speech_like_sndenv(orc, sr, tmp_path, None, rep)
    assert rep["seg_cnt"] == 30 and rep["zero_frames"] >= 60
