"""Worker of tests/test_distributed.py: one rank of a world_size-N gloo job on the CPU.
Each rank computes the mel features of ITS shard of the batch (through the C ABI -- the CPU
thread-emulator build of the kernels, since there is no GPU here), then all ranks reassemble the
full feature tensor with the path's single all-gather and check it against the oracle."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE, os.path.join(HERE, "emul")]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import backend  # noqa: E402
import workloads as W  # noqa: E402
from auditory_amd import capi, runtime, synth  # noqa: E402
from auditory_amd.batch import HostGather, allgather_features, shard_range  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    n_total = int(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oc = W.OracleCfg(orc, "cfg2_16k_n512_nf40", segment_ms=100.0)   # T = 14: keeps the emulator quick
    L = 4000
    sig, _ = synth.batch(7, n_total, L, oc.sr)                       # every rank can build any row
    lo, hi = shard_range(n_total, rank, world)
    with backend.emulated("plain"):
        plan = W.product_plan(oc, capi.AUD_F32)
        assert plan.kernel_name == "w16x16"
        n = hi - lo
        items = runtime.make_items(np.arange(n) * L, [L] * n, [0] * n)
        mel, _, _ = plan.melspec_host(sig[lo:hi].ravel(), items)      # this rank only touches its shard
        plan.close()
    full = allgather_features(torch.from_numpy(mel.astype(np.float32)), world, n_total=n_total).numpy()
    assert full.shape == (n_total, oc.nf, oc.T)
    rc, ref, _ = orc.process_batch(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig.ravel(),
                                   np.arange(n_total) * L, np.full(n_total, L), np.zeros(n_total))
    ok, msg = W.feature_close(full, ref, capi.AUD_F32, lin_axis=1)
    assert ok, msg
    # every rank ends with the same bytes
    t = torch.from_numpy(full.copy())
    dist.broadcast(t, 0)
    assert np.array_equal(t.numpy(), full)
    # SURVEY 8(e)'s alternative: no collective, every rank copies its slab into its slot of ONE host buffer all ranks map;
    # behind a barrier rank 0 (and everyone else) reads the batch in rank order -- the same values as the all-gather's
    longest = max(b - a for a, b in (shard_range(n_total, r, world) for r in range(world)))
    hg = HostGather.create(world, rank, longest * oc.nf * oc.T)
    for slab in (0, 1):
        hg.put(torch.from_numpy(mel.astype(np.float32)) + float(slab), slab)
    dist.barrier()
    for slab in (0, 1):
        rows = hg.view(slab).view(world, longest, oc.nf, oc.T)
        got = torch.cat([rows[r, :b - a] for r, (a, b) in enumerate(shard_range(n_total, r, world) for r in range(world))]).numpy()
        assert np.array_equal(got, full + np.float32(slab), equal_nan=True), slab
    dist.barrier()          # nobody unmaps (rank 0: unlinks) before everyone has read
    path = hg.path
    hg.close()
    dist.barrier()
    assert not os.path.exists(path)
    dist.destroy_process_group()
    print("RANK-OK", rank, lo, hi)


if __name__ == "__main__":
    main()
