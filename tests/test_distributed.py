"""N>1 path on the CPU: world_size-2 (and 3, uneven shards) gloo jobs; see dist_worker.py."""
import os
import socket
import subprocess
import sys

import pytest

from auditory_amd.batch import shard_range

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 4096, 4099):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_total", [(2, 6), (3, 7)])
def test_sharded_melspec_allgather_gloo(world, n_total):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(n_total)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0 and "RANK-OK" in o, e[-3000:]


def test_direct_allgather_indexing_emulated():
    """aud_gather_* (the direct-pattern all-gather: one device-to-device push per peer) on the CPU emulator: three ranks as
    three contexts of one process (the emulator's inter-process handle carries the pointer), uneven counts, two rounds --
    every rank's receive buffer ends up [slot r] = rank r's slab, nothing beyond a slab's count is touched, and the
    misuse paths are AUD_EINVAL"""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(HERE, "emul"))
    import backend
    from auditory_amd import capi, runtime
    from auditory_amd.batch import DirectGather
    G, slab = 3, 1000
    with backend.emulated("plain"):
        ctxs = [runtime.Context(0) for _ in range(G)]
        gs = [DirectGather(ctxs[r], G, r, slab) for r in range(G)]
        with pytest.raises(capi.AuditoryError):                       # peers not opened yet
            gs[0].allgather(0, 10)
        for g in gs:
            g.open_peers([x.handle for x in gs])
        with pytest.raises(capi.AuditoryError):
            gs[0].open_peers([x.handle for x in gs])                  # twice
        bufs = [np.ctypeslib.as_array(C.cast(g.recv_ptr, C.POINTER(C.c_float)), shape=(2, G, slab)) for g in gs]
        for rnd, counts in enumerate(([1000, 700, 1], [5, 1000, 999], [1, 2, 3])):
            for b in bufs:
                b[rnd & 1] = -1.0
            sends = [np.arange(counts[r], dtype=np.float32) + 1000.0 * (r + 1) + rnd for r in range(G)]
            for r in range(G):
                assert gs[r].allgather(sends[r].ctypes.data, counts[r]) == rnd & 1     # the slabs alternate per step
            for r in range(G):
                gs[r].wait()                                                            # every peer's flag has reached this step
            assert [g.timeouts() for g in gs] == [0] * G
            for b in bufs:
                for r in range(G):
                    assert np.array_equal(b[rnd & 1, r, :counts[r]], sends[r]) and (b[rnd & 1, r, counts[r]:] == -1.0).all()
            if rnd == 1:      # the previous step's slab is untouched by this one
                assert bufs[0][0, 0, 0] == 1000.0 and bufs[0][0, 1, 0] == 2000.0
        # a peer that never arrives: the wait ENDS at its poll bound and is counted (never a hung queue)
        gs[0].allgather(sends[0].ctypes.data, 1)
        gs[0].wait()
        assert gs[0].timeouts() == G - 1
        for r in range(1, G):
            gs[r].allgather(sends[r].ctypes.data, 1)
        with pytest.raises(capi.AuditoryError):
            gs[1].allgather(sends[1].ctypes.data, slab + 1)           # more than the slab
        for g in gs:
            g.close()
        for c in ctxs:
            c.close()
