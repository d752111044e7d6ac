"""One of the two processes of tests/test_gpu_parity.py::test_direct_allgather_two_processes_one_gpu (both on cuda:0):
usage gather_worker.py RANK DIR -- handles and ready-flags travel through files in DIR."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

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
    send = torch.arange(slab, dtype=torch.float32, device=dev) + 1000.0 * (rank + 1)
    count = slab if rank == 0 else slab - 7
    if rank == 0:   # the fork / join of the per-peer streams is capturable
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g.allgather(send.data_ptr(), count, torch.cuda.current_stream(dev).cuda_stream)
        graph.replay()
    else:
        g.allgather(send.data_ptr(), count, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    open(os.path.join(d, "pushed%d" % rank), "w").close()              # the host-side barrier the contract asks for
    wait_for(os.path.join(d, "pushed%d" % other))
    torch.cuda.synchronize()
    got = recv.cpu()
    for r, n in ((0, slab), (1, slab - 7)):
        want = torch.arange(n, dtype=torch.float32) + 1000.0 * (r + 1)
        assert torch.equal(got[r, :n], want), (rank, r)
        assert bool((got[r, n:] == -1.0).all()), (rank, r)
    open(os.path.join(d, "checked%d" % rank), "w").close()             # keep the buffers alive until both have read them
    wait_for(os.path.join(d, "checked%d" % other))
    g.close()
    print("GATHER-OK", rank)


if __name__ == "__main__":
    main()
