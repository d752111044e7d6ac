import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import memguard  # noqa: E402  resident-memory ceiling: a runaway allocation ends this process, not the GPU box

memguard.install()


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """CPU tier only (`-m "not gpu"`): spread the tests over (cores - 1, at most 7) xdist workers unless the caller chose a worker
    count (or AUD_TEST_WORKERS=0).  The emulator tests spend their time in thread barriers, so this cuts the
    tier from ~8 to ~3 minutes on 8 cores.  The GPU tier always runs in one process (one process on the card)."""
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER"):
        return None  # an xdist worker: never spawn workers of its own
    want = os.environ.get("AUD_TEST_WORKERS", str(max(2, min(8, os.cpu_count() or 4) - 1)))
    if (getattr(config.option, "markexpr", "") or "").strip() != "not gpu" or want in ("", "0"):
        return None
    if not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None):
        return None
    if getattr(config.option, "collectonly", False) or getattr(config.option, "usepdb", False):
        return None
    config.option.numprocesses = int(want)
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    cpu_tier = (getattr(config.option, "markexpr", "") or "").strip() == "not gpu"
    if cpu_tier and not hasattr(config, "workerinput"):
        # controller (or a plain single-process run): build the checkers once, before any worker needs them
        # (the builders also take a file lock, so a worker that gets there first is safe too)
        try:
            from oracle import oracle
            oracle.build()
            sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
            import build_emul
            build_emul.build("plain")
        except Exception as e:  # the tests that need them will report the real error
            print("conftest: pre-build skipped: %r" % (e,))


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


def pytest_collection_modifyitems(config, items):
    """CPU tier: the few long tests (sanitizer runs of the emulator, hipcc's resource report, the multi-rank dry runs) go FIRST,
    so that the xdist workers pack the many short ones around them instead of starting a two-minute test at the end"""
    if (getattr(config.option, "markexpr", "") or "").strip() != "not gpu":
        return
    long_ones = ("under_tsan", "under_asan", "no_scratch_and_fit", "device_api_tests_dry_run", "eight_ranks", "test_bench_dry_run[",
                 "kwta_vs_oracle", "peer_dies", "two_ranks", "single_rank_collective", "workgroup_order")
    rank = {id(it): (0 if any(k in it.nodeid for k in long_ones) else 1) for it in items}
    items.sort(key=lambda it: rank[id(it)])
