"""Dry run of bench.py's control flow on the CPU (no timing claims): torch.cuda is replaced by a
stub and the C ABI by the emulator build, so that argument plumbing, the JSON contract and the
cpu_baseline leg are exercised before any GPU minute is spent on them."""
import json
import os
import sys
import types

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "emul"))


class _Stream:
    cuda_stream = 0


class _Event:
    def __init__(self, enable_timing=False):
        import time
        self._t = None
        self._time = time

    def record(self):
        self._t = self._time.perf_counter()

    def elapsed_time(self, other):
        return max(1e-6, (other._t - self._t) * 1e3)

    def query(self):
        return True   # (the emulator's launches are synchronous)


class _Graph:
    def __init__(self):
        raise RuntimeError("no hipGraph in the dry run")


def _patch(monkeypatch):
    real_device = torch.device
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda d=None: None)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda d=None: _Stream())
    monkeypatch.setattr(torch.cuda, "Event", _Event)
    monkeypatch.setattr(torch.cuda, "CUDAGraph", _Graph)
    monkeypatch.setattr(torch.cuda, "Stream", _SideStream)
    monkeypatch.setattr(_Stream, "wait_stream", lambda self, other: None, raising=False)
    monkeypatch.setattr(torch, "device", lambda *a, **k: real_device("cpu"))


def test_bench_dry_run_two_ranks(tmp_path):
    """the --gpus 2 control flow (sharded batches, max-over-ranks timing, the all-gather region) on gloo"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(HERE, "bench_dry_worker.py")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, worker, "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--batch", "2", "--ring-mb", "0.2", "--min-seconds", "0", "--cfg3-total", "6",
                                       "--dist-backend", "gloo"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1][-2000:] + outs[1][1][-2000:]
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]   # only rank 0 prints the JSON line
    # several ranks: `value` is BASELINE configs[2] as stated (a fixed total batch, sharded, kernel + all-gather): strong scaling
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and "cpu_baseline" not in line
    assert line["rccl_ranks"] == 2 and line["gathered_shape"] == [6, 40, 104]           # configs[2] region, shrunk to 6 in total
    assert line["config"]["batch_per_gpu"] == 3 and "all-gather" in line["config"]["sharding"]
    assert line["parity"]["pass"] and line["parity"]["n_past_1e-5"] == 0 and "every rank" in line["parity"]["checked"]
    assert line["no_collective"]["batch"] == 2 and line["no_collective"]["parity"]["pass"]   # the collective-free sharded step
    assert "also" not in line and "modes" not in line
    # the direct-pattern reassembly is timed as a side key; between two emulator PROCESSES a peer's buffer cannot be mapped,
    # which every rank must agree on and survive (on the GPU box the key carries a value and a parity object)
    assert "error" in line["direct_gather"] and "peer" in line["direct_gather"]["error"]


@pytest.mark.parametrize("total", [16, 19], ids=["even", "uneven"])
def test_bench_dry_run_eight_ranks(tmp_path, total):
    """The target world size: EIGHT gloo ranks through bench.py's N > 1 flow (configs[2] shrunk to `total` utterances) -- shard
    sizes, the gathered tensor's shape, rank 0's parity check over all eight blocks, and the keys that make the first real
    8-GPU run tell the whole story in one line: weak_scaling (with the same step on rank 0 alone) and scaling_bound."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(HERE, "bench_dry_worker.py")
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, worker, "--gpus", "8", "--steps", "2", "--warmup", "1",
                                       "--batch", "2", "--ring-mb", "0.2", "--min-seconds", "0", "--cfg3-total", str(total),
                                       "--dist-backend", "gloo", "--no-direct-alt"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), "".join(o[1][-800:] for o in outs)
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    for o in outs[1:]:
        assert not [l for l in o[0].splitlines() if l.startswith("{")]            # only rank 0 prints
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["scaling"] == "strong"
    assert line["gathered_shape"] == [total, 40, 104]
    sizes = line["shard_sizes"]
    assert len(sizes) == 8 and sum(sizes) == total and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert line["config"]["batch_per_gpu"] == sizes[0]
    assert line["parity"]["pass"] and "(8 ranks)" in line["parity"]["checked"] and line["parity"]["elements"] >= 8 * 2 * 40 * 104
    ws, sb = line["weak_scaling"], line["scaling_bound"]
    assert ws["value"] == line["no_collective"]["value"] and ws["batch_per_gpu"] == 2 and ws["rank0_alone_value"] > 0
    assert ws["x_of_1gpu"] == pytest.approx(ws["value"] / ws["rank0_alone_value"], rel=1e-2)
    assert sb["kernel_us"] > 0 and sb["xgmi_us"] == pytest.approx(sizes[0] * 40 * 104 * 4 / 76.8e9 * 1e6, abs=0.06)
    assert sb["max_x_of_1gpu"] == pytest.approx(8 * sb["kernel_us"] / max(sb["kernel_us"], sb["xgmi_us"]), rel=1e-2)
    assert sb["bound_by"] in ("kernel", "xgmi") and line["xgmi_bound_us"]["per_step"] == sb["xgmi_us"]
    assert line["launch" if "launch" in line else "config"] and "eager (hipGraph capture failed" in line["config"]["launch"]


def test_bench_dry_run_line_survives_a_dead_side_measurement(tmp_path):
    """Two gloo ranks; both processes end abruptly (os._exit: what a GPU fault's abort does) inside the optional direct-pattern
    side measurement, the last thing the run does.  Rank 0's guard process -- forked before the GPU was touched, holding a copy
    of the finished line -- must print that line: every key measured before the fault, and the reason in `direct_gather`."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(HERE, "bench_dry_worker.py")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", AUD_BENCH_TEST_DIE_IN_SIDE_MEASUREMENT="1")
        procs.append(subprocess.Popen([sys.executable, worker, "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--batch", "2", "--ring-mb", "0.2", "--min-seconds", "0", "--cfg3-total", "6",
                                       "--dist-backend", "gloo"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode for p in procs] == [7, 7], outs[0][1][-2000:] + outs[1][1][-2000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["parity"]["pass"] and line["value"] > 0
    assert "ended during" in line["direct_gather"]["error"]


class _SideStream(_Stream):
    def __init__(self, dev=None):
        pass

    def wait_stream(self, other):
        pass


@pytest.mark.parametrize("mode", ["rccl", "direct"])
def test_bench_dry_run_single_rank_collective(monkeypatch, capsys, mode):
    """--dist-single: the multi-GPU region (shard + collective per step, gathered tensor checked) on ONE rank, both gather
    modes -- gloo stands in for RCCL, the emulator's handles for the inter-process ones"""
    import socket
    import backend
    import bench
    _patch(monkeypatch)
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    monkeypatch.setenv("MASTER_PORT", str(sk.getsockname()[1]))
    sk.close()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2", "--warmup", "1", "--batch", "2", "--ring-mb", "0.2",
                                      "--min-seconds", "0", "--cfg3-total", "4", "--dist-single", "--dist-backend", "gloo",
                                      "--gather-mode", mode, "--no-cpu-baseline"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with backend.emulated("plain"):
        bench.main()
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["rccl_ranks"] == 1 and line["gathered_shape"] == [4, 40, 104]
    assert line["parity"]["pass"] and ("direct pattern" in line["collective"]) == (mode == "direct")
    assert line["no_collective"]["parity"]["pass"]


@pytest.mark.parametrize("extra", [[], ["--only-headline"], ["--workload", "cfg4"], ["--sig-dtype", "i16", "--only-headline"],
                                   ["--compute", "f32", "--launch", "eager", "--only-headline"],
                                   ["--workload", "sndenv"], ["--stereo", "--only-headline"], ["--workload", "sndenv_cfg1"],
                                   ["--workload", "rate48k"]])
def test_bench_dry_run(monkeypatch, capsys, extra):
    import backend
    import bench
    _patch(monkeypatch)
    monkeypatch.setattr(bench, "cpu_baseline",
                        lambda wl, pcm: bench.__dict__["_real_cpu_baseline"](wl, pcm, target_s=0.2, max_threads=2, all_seconds=0.3))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2", "--warmup", "1", "--batch", "2", "--ring-mb", "0.2",
                                      "--min-seconds", "0", "--cfg3-total", "4", "--cfg5-batch", "2"] + extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with backend.emulated("plain"):
        bench.main()
    out = capsys.readouterr().out.strip().splitlines()
    line = json.loads(out[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["unit"] == "audio-seconds/sec" and line["n_gpus"] == 1 and line["steps"] == 2 * line["config"]["repeats"]
    assert "25 ms" in line["metric"] and line["config"]["win_samples"] == (1103 if "sndenv_cfg1" in extra else 1200 if "rate48k" in extra else 400)   # the metric's own parameter set
    assert set(line["parity"]) >= {"criterion", "elements", "max_scaled_err", "n_past_1e-5", "pass"} and line["parity"]["pass"]
    assert line["vs_baseline"] is None and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert line["cpu_baseline"]["cores"] <= 16 and line["cpu_baseline"]["kind"] == "port"
    ac = line["cpu_baseline"]["all_cores"]      # SURVEY 8d's all-host-cores flavour beside the bounded one
    if line["cpu_baseline"]["affinity_cores"] > line["cpu_baseline"]["cores"]:
        assert ac["threads"] == line["cpu_baseline"]["affinity_cores"] and ac["value"] > 0 and ac["seconds"] < 30
    else:
        assert ac is None
    assert ("1e-5" in line["dtype_note"]) == (line["dtype"] == "f64")       # the contract, in the line: float32 epilogue, margin
    any_n = line["config"]["kernel"] in ("generic", "chirp2304")    # the any-N rows say what THEY wait for, and that nothing of them is float32
    assert line["roofline"]["note"].startswith("workgroup-level transform through LDS" if any_n else
                                               "float64 VALU floor" if line["dtype"] == "f64" else "float32 plan")
    assert line["epilogue"] == (line["dtype"] if any_n else "f32") and line["roofline"]["limited_by"] == (
        "lds_round_trips_and_barriers" if any_n else "valu_" + line["dtype"])
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["value"] > 0 and line["roofline"]["achieved"] >= 0
    if not extra:   # the default line nests the other BASELINE configurations, each with its own strict parity object
        assert set(line["also"]) == {"n512_f64", "cfg3", "cfg4", "cfg5"} and set(line["modes"]) >= {"n400_f32", "n512_f32"}
        for k, v in line["also"].items():
            assert v["parity"]["pass"], (k, v["parity"])
        assert line["also"]["cfg4"]["parity"]["gabor"]["pass"] and line["also"]["cfg3"]["total_batch"] == 4
        assert line["also"]["cfg5"]["kernel"] == "w64x16" and line["also"]["cfg5"]["batch"] == 2


@pytest.mark.parametrize("extra", [["--win-ms", "32", "--compute", "f32"], ["--win-ms", "25", "--streams", "2"]])
def test_ab_bench_dry_run(monkeypatch, capsys, extra):
    """tools/ab_bench.py (the interleaved A/B of kernel variants run at first GPU contact) end to end on the CPU"""
    import importlib.util
    import backend
    _patch(monkeypatch)
    monkeypatch.setattr(torch.Tensor, "to", lambda self, *a, **k: self)
    real_empty = torch.empty
    monkeypatch.setattr(torch, "empty", lambda *a, **k: real_empty(*a, **{**k, "device": "cpu"}))
    spec = importlib.util.spec_from_file_location("ab_bench", os.path.join(ROOT, "tools", "ab_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["ab_bench.py", "--batch", "2", "--rounds", "1", "--launches", "2", "--warm", "1",
                                      "--generic"] + extra)
    with backend.emulated("plain"):
        mod.main()
    out = capsys.readouterr().out
    assert "median" in out and "shipped generic" in out
    assert ("w16x16" in out) == ("32" in extra) and ("w20x10" in out) == ("25" in extra)


def test_smoke_dry_run(monkeypatch, capsys):
    """__graft_entry__.smoke() end to end on the CPU (emulator build, torch.cuda stubbed)"""
    import backend
    import __graft_entry__ as G
    _patch(monkeypatch)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    with backend.emulated("plain"):
        G.smoke()
    out = capsys.readouterr().out
    assert "smoke ok: strict float64 w20x10" in out and "smoke strict float64 w20x10 mel" in out and "RELAXED float32 w16x16" in out


def test_bench_dry_run_peer_dies(tmp_path):
    """a rank that dies at the collective must not leave the other one hanging: the survivor exits NON-ZERO within the
    --dist-timeout it was given (process-group timeout + the bounded first step of bench.py's collective region)"""
    import socket
    import subprocess
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(HERE, "bench_dry_worker.py")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", AUD_DRY_DIE_IN_COLLECTIVE="1")
        procs.append(subprocess.Popen([sys.executable, worker, "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--batch", "2", "--ring-mb", "0.2", "--min-seconds", "0", "--cfg3-total", "6",
                                       "--dist-backend", "gloo", "--dist-timeout", "20"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    t0 = time.time()
    outs = [p.communicate(timeout=240) for p in procs]
    assert procs[1].returncode == 9, outs[1][1][-1500:]                       # the injected death
    assert procs[0].returncode not in (0, None), "the survivor reported success: " + outs[0][0][-500:]
    assert time.time() - t0 < 200
    assert not any(ln.startswith("{") for ln in outs[0][0].splitlines()), "no JSON line from a broken job"


def test_device_api_tests_dry_run(monkeypatch, orc):
    """the torch-tensor based -m gpu tests (BatchProcessor plumbing, sample types, fused mel+gabor, the
    size-independent properties at a small batch) on the CPU: emulator build, torch.cuda stubbed"""
    import backend
    import test_gpu_parity as G
    _patch(monkeypatch)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    real_empty, real_zeros = torch.empty, torch.zeros
    monkeypatch.setattr(torch, "empty", lambda *a, **k: real_empty(*a, **{**k, "device": "cpu"}))
    monkeypatch.setattr(torch, "zeros", lambda *a, **k: real_zeros(*a, **{**k, "device": "cpu"}))
    with backend.emulated("plain"):
        F32, F64 = G.capi.AUD_FAST_F32, G.capi.AUD_F64
        G.test_input_dtypes_agree(orc, torch, "cfg2_16k_n512_nf40", 300.0, F32)
        G.test_input_dtypes_agree(orc, torch, "cfg2_16k_n512_nf40", 300.0, F64)
        G.test_int16_normalisation_exhaustive(orc, torch, F32)
        G.test_int16_normalisation_exhaustive(orc, torch, F64)
        G.test_input_dtypes_agree(orc, torch, "cfg2_16k_n400_nf40", 200.0, F64)
        G.test_input_dtypes_agree(orc, torch, "cfg5_44k_n2048_nf128", 80.0, F32)
        G.test_input_dtypes_agree(orc, torch, "cfg1_44k_n1103_nf32", None, F64)
        G.test_process_batch_mel_plus_gabor(orc, torch, F64)
        G.test_process_batch_mel_plus_gabor(orc, torch, F32)
        G.test_process_then_kwta_device_resident(orc, torch, F64, n=1)
        G.test_full_size_properties_cfg2(orc, torch, F64, B=4)
        G.test_zeroed_plan_desc_is_the_conforming_plan(orc, torch)


def test_rocprof_summary_on_synthetic_csvs(tmp_path):
    """tools/rocprof_summary.py against CSVs laid out like rocprofv3's (<dir>/<pass>/<host>/<pid>_*.csv): the
    per-launch HBM bytes it hands to bench.py (FETCH_SIZE doubled and in KB, WRITE_SIZE in KB, per the guide)"""
    import subprocess
    k = "void aud::(anonymous namespace)::k_melspec_w20<double, 0, 4, 4>(aud::MelspecArgs, aud::WaveArgs)"
    d = tmp_path / "prof"
    (d / "stats" / "box" ).mkdir(parents=True)
    (d / "stats" / "box" / "77_kernel_stats.csv").write_text(
        '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
        '"%s",220,1100000,5000.0,97.1,4800,6100,80.0\n"other",3,30000,10000.0,2.9,1,2,3\n' % k)
    for name, ctr, val in (("fetch", "FETCH_SIZE", 8000.0), ("write", "WRITE_SIZE", 4160.0)):
        (d / name / "box").mkdir(parents=True)
        rows = ['"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size",'
                '"Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count",'
                '"Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"']
        for i in range(4):
            rows.append('%d,%d,1,1,77,77,458752,9,"%s",256,40960,0,72,0,96,"%s",%s,1,2' % (i, i, k, ctr, val))
        (d / name / "box" / "77_counter_collection.csv").write_text("\n".join(rows) + "\n")
    out = tmp_path / "profiles"
    out.mkdir()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), str(d), "rXX", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "k_melspec_w20" in r.stdout and "5000.0" in r.stdout
    t = json.load(open(out / "pmc_traffic.json"))
    assert t["read_bytes"] == 2 * 8000.0 * 1024 and t["write_bytes"] == 4160.0 * 1024
    assert t["read_bytes_raw_fetch_size"] == 8000.0 * 1024
    assert t["hbm_bytes_per_launch"] == t["read_bytes"] + t["write_bytes"] and "w20" in t["kernel"]
    kj = json.load(open(out / "rXX_kernels.json"))["kernels"][k]
    assert kj["family"] == "w20x10" and kj["compute"] == "f64" and kj["avg_duration_ns"] == 5000.0 and kj["write_bytes"] == 4160.0 * 1024
