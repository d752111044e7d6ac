"""One of the two processes of tests/test_gpu_parity.py::test_direct_allgather_two_processes_one_gpu (both on cuda:0):
usage gather_worker.py RANK DIR -- handles and ready-flags travel through files in DIR."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

os.environ.setdefault("AUD_GATHER_WAIT_MS", "30000")   # (process start-up skew between the two workers, not link time)

import torch  # noqa: E402

from auditory_amd import runtime  # noqa: E402
from auditory_amd.batch import DirectGather  # noqa: E402


def wait_for(path, timeout=120.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise SystemExit("timeout waiting for " + path)
        time.sleep(0.05)
    return path


def main():
    rank, d = int(sys.argv[1]), sys.argv[2]
    other, slab = 1 - rank, 4096
    dev = torch.device("cuda", 0)
    ctx = runtime.get_ctx(0)
    g = DirectGather(ctx, 2, rank, slab)
    with open(os.path.join(d, "h%d.tmp" % rank), "wb") as fh:
        fh.write(g.handle)
    os.rename(os.path.join(d, "h%d.tmp" % rank), os.path.join(d, "h%d" % rank))
    peer = open(wait_for(os.path.join(d, "h%d" % other)), "rb").read()
    handles = [None, None]
    handles[rank], handles[other] = g.handle, peer
    g.open_peers(handles)
    recv = g.recv(dev)
    recv.fill_(-1.0)
    torch.cuda.synchronize()
    open(os.path.join(d, "ready%d" % rank), "w").close()               # buffers initialised on both sides before any push
    wait_for(os.path.join(d, "ready%d" % other))
    sends = [torch.arange(slab, dtype=torch.float32, device=dev) + 1000.0 * (rank + 1) + 7.0 * step for step in range(2)]
    count = slab if rank == 0 else slab - 7
    # four steps, NO host-side barrier between the ranks: every step = pushes + arrival signal, then the (bounded) wait for the
    # peer's flag, behind which a kernel of THIS rank reads the step's slab (`seen`).  Rank 0 runs them as a captured pair of
    # steps replayed twice (the fork / join of the per-peer streams, the signal and the wait are capturable; the step numbers
    # live on the device and advance per replay), rank 1 eagerly.
    seen = torch.zeros((4, 2, slab), dtype=torch.float32, device=dev)

    def step(i, out):
        st = torch.cuda.current_stream(dev).cuda_stream
        which = g.allgather(sends[i % 2].data_ptr(), count, st)
        assert which == i % 2
        g.wait(st)
        out.copy_(recv[which])          # on the same stream, behind the wait: the slab must be complete here

    if rank == 0:
        graph = torch.cuda.CUDAGraph()
        pair = torch.zeros((2, 2, slab), dtype=torch.float32, device=dev)
        with torch.cuda.graph(graph):
            for i in range(2):
                step(i, pair[i])
        for rep in range(2):
            graph.replay()
            seen[2 * rep:2 * rep + 2].copy_(pair)
    else:
        for i in range(4):
            step(i, seen[i])
    torch.cuda.synchronize()
    assert g.timeouts() == 0, "an arrival wait ran into its poll bound"
    got = seen.cpu()
    for i in range(4):
        for r, n in ((0, slab), (1, slab - 7)):
            want = torch.arange(n, dtype=torch.float32) + 1000.0 * (r + 1) + 7.0 * (i % 2)
            assert torch.equal(got[i, r, :n], want), (rank, i, r)
            assert bool((got[i, r, n:] == -1.0).all()), (rank, i, r)
    open(os.path.join(d, "checked%d" % rank), "w").close()             # keep the buffers alive until both have read them
    wait_for(os.path.join(d, "checked%d" % other))
    fine = g.flags_fine()
    g.close()

    # ---- a peer that never arrives, on hardware: a second gather with a SHORT poll bound; rank 1 maps everything and then
    # does nothing, rank 0 pushes and waits.  The wait must END (0.3 s), count the time-out, fill rank 1's slot of the step's
    # slab with NaN, and the next call must be refused (AUD_EBROKEN) without any synchronisation of rank 0's own.
    import numpy as np
    from auditory_amd import capi
    os.environ["AUD_GATHER_WAIT_MS"] = "300"
    g = DirectGather(ctx, 2, rank, slab)
    with open(os.path.join(d, "t%d.tmp" % rank), "wb") as fh:
        fh.write(g.handle)
    os.rename(os.path.join(d, "t%d.tmp" % rank), os.path.join(d, "t%d" % rank))
    handles[rank], handles[other] = g.handle, open(wait_for(os.path.join(d, "t%d" % other)), "rb").read()
    g.open_peers(handles)
    recv = g.recv(dev)
    recv.fill_(-1.0)
    torch.cuda.synchronize()
    open(os.path.join(d, "tready%d" % rank), "w").close()
    wait_for(os.path.join(d, "tready%d" % other))
    if rank == 0:
        st = torch.cuda.current_stream(dev).cuda_stream
        t0 = time.time()
        which = g.allgather(sends[0].data_ptr(), slab, st)
        g.wait(st)
        torch.cuda.synchronize()
        dt = time.time() - t0
        assert g.timeouts() == 1 and dt < 20.0, (g.timeouts(), dt)
        got = recv.cpu().numpy()
        assert np.array_equal(got[which, 0], sends[0].cpu().numpy()) and np.isnan(got[which, 1]).all()
        assert (got[1 - which] == -1.0).all()
        for call in (lambda: g.allgather(sends[0].data_ptr(), slab, st), lambda: g.wait(st)):
            try:
                call()
                raise SystemExit("a broken gather accepted another call")
            except capi.AuditoryError as ex:
                assert ex.status == capi.AUD_EBROKEN and "poll bound" in str(ex), str(ex)
        open(os.path.join(d, "tdone"), "w").close()
        print("GATHER-TIMEOUT-OK waited %.2f s" % dt)
    else:
        wait_for(os.path.join(d, "tdone"))
    g.close()
    print("GATHER-OK", rank, "flags_fine", fine)


if __name__ == "__main__":
    main()
