"""Host-memory ceiling for every process this repo starts on a GPU box (bench.py, the test tiers, smoke()).

A runaway host allocation on a GPU box does not end in MemoryError: the box dies and takes the
round's GPU access with it (that happened in round 1).  RLIMIT_AS cannot be used for this -- the
HIP runtime reserves terabytes of address space at start-up -- so the ceiling is on RESIDENT
memory: a daemon thread reads /proc/self/statm five times a second and ends the process with exit
code 98 and a one-line message when the resident set passes the limit (AUD_RSS_LIMIT_GB, default
12 GiB; a bench or test process of this repo needs well under 2)."""
import os
import sys
import threading
import time

_started = False


def rss_bytes():
    with open("/proc/self/statm") as fh:
        return int(fh.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def install(limit_gb=None, period_s=0.2):
    global _started
    if _started or not os.path.exists("/proc/self/statm"):
        return
    limit = float(os.environ.get("AUD_RSS_LIMIT_GB", limit_gb if limit_gb is not None else 12.0)) * (1 << 30)

    def watch():
        while True:
            try:
                r = rss_bytes()
            except OSError:
                return
            if r > limit:
                sys.stderr.write("memguard: resident set %.1f GiB passed the %.1f GiB ceiling -- exiting 98\n"
                                 % (r / (1 << 30), limit / (1 << 30)))
                sys.stderr.flush()
                os._exit(98)
            time.sleep(period_s)

    t = threading.Thread(target=watch, name="memguard", daemon=True)
    t.start()
    _started = True
