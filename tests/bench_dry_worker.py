"""One rank of tests/test_bench_dryrun.py::test_bench_dry_run_two_ranks: bench.main() with torch.cuda
stubbed out and the C ABI bound to the emulator build (CPU dry run of the multi-rank control flow)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, "emul")]

import pytest  # noqa: E402
import torch  # noqa: E402

import backend  # noqa: E402
import bench  # noqa: E402
import test_bench_dryrun as T  # noqa: E402

mp = pytest.MonkeyPatch()
T._patch(mp)
with backend.emulated("plain"):
    bench.main()
mp.undo()
