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
if os.environ.get("AUD_DRY_DIE_IN_COLLECTIVE") == os.environ.get("RANK"):
    # fault injection for test_bench_dry_run_peer_dies: this rank vanishes at the moment it would enter the path's collective
    import torch.distributed as dist  # noqa: E402
    mp.setattr(dist, "all_gather_into_tensor", lambda *a, **k: os._exit(9))
with backend.emulated("plain"):
    bench.main()
mp.undo()
