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
    """aud_gather_* (the direct-pattern all-gather: one device-to-device push per peer, arrival flags, two receive slabs) on the
    CPU emulator: parity_cases.case_direct_gather_three_ranks"""
    sys.path.insert(0, os.path.join(HERE, "emul"))
    import backend
    import parity_cases as PC
    with backend.emulated("plain"):
        PC.case_direct_gather_three_ranks()


def test_host_gather_single_process(tmp_path):
    """batch.HostGather with both ranks in ONE process (two mappings of the same shared buffer): slots, slabs, short slabs,
    argument checks, and the file is gone after rank 0's close"""
    import torch
    from auditory_amd import capi
    from auditory_amd.batch import HostGather
    name = "auditory_hip_test_%d" % os.getpid()
    a = HostGather(name, 2, 0, 12)
    with pytest.raises(FileExistsError):
        HostGather(name, 2, 0, 12)                       # the name is taken: a second job must not scribble into it
    with pytest.raises(capi.AuditoryError):
        HostGather(name, 2, 1, 13, timeout=0.2)          # another geometry: the sizes do not match, it never "appears"
    b = HostGather(name, 2, 1, 12)
    a.put(torch.arange(12, dtype=torch.float32).view(3, 4), 0)
    b.put(torch.arange(8, dtype=torch.float32) + 100.0, 0)    # a short slab: the tail of the slot is not touched
    b.put(torch.full((12,), 7.0), 1)
    for g in (a, b):
        v = g.view(0)
        assert v.shape == (2, 12) and torch.equal(v[0], torch.arange(12.0)) and torch.equal(v[1, :8], torch.arange(8.0) + 100.0)
        assert torch.equal(v[1, 8:], torch.zeros(4)) and torch.equal(g.view(1)[1], torch.full((12,), 7.0))
        assert torch.equal(g.view(1)[0], torch.zeros(12))
    with pytest.raises(capi.AuditoryError):
        a.put(torch.zeros(13), 0)
    with pytest.raises(capi.AuditoryError):
        a.put(torch.zeros(4, dtype=torch.float64), 0)
    with pytest.raises(capi.AuditoryError):
        a.put(torch.zeros(4), 2)
    b.close()
    assert os.path.exists(a.path)
    a.close()
    assert not os.path.exists(os.path.join("/dev/shm", name))
