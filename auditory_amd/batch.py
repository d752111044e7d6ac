"""Device-resident batch driver: many utterances as one grid, sharded over the GPUs of a node.

torch is used for what it is good at -- device memory, streams, torch.distributed (RCCL) --
and never for arithmetic: every feature value comes out of libauditory_hip.so.
"""
import numpy as np
import torch

from . import capi, runtime

_SIG_DTYPES = {torch.float32: capi.AUD_F32, torch.float64: capi.AUD_F64, torch.int16: capi.AUD_I16}


def shard_range(n_items, rank, world):
    """contiguous block of ceil/floor(n/world) items for `rank`"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class BatchProcessor:
    """Runs the fused frame->mel (and gabor) kernels on tensors that already live in HBM."""

    def __init__(self, plan, device):
        self.plan = plan
        self.device = torch.device(device)

    def upload_items(self, items):
        """numpy ITEM_DTYPE array -> device uint8 tensor holding the same bytes"""
        raw = np.frombuffer(np.ascontiguousarray(items).tobytes(), np.uint8).copy()
        return torch.from_numpy(raw).to(self.device)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def melspec(self, sig, items_dev, n_items, mel=None, power=None, log_power=None):
        p = self.plan
        if sig.device != self.device or not sig.is_contiguous():
            raise ValueError("signal must be a contiguous tensor on %s" % self.device)
        if mel is None:
            mel = torch.empty((n_items, p.nf, p.T), dtype=torch.float32, device=self.device)
        p.melspec_dev(sig.data_ptr(), _SIG_DTYPES[sig.dtype], items_dev.data_ptr(), n_items,
                      mel.data_ptr(), power.data_ptr() if power is not None else 0,
                      log_power.data_ptr() if log_power is not None else 0, self._stream())
        return mel

    def gabor(self, mel, out, by_time=False):
        n, rows, cols = mel.shape
        self.plan.gabor_dev(mel.data_ptr(), n, rows, cols, list(out.shape[1:]), out.data_ptr(),
                            by_time, self._stream())
        return out

    def process(self, sig, items_dev, n_items, pools_y, pools_x, mel=None, gabor=None):
        p = self.plan
        if mel is None:
            mel = torch.empty((n_items, p.nf, p.T), dtype=torch.float32, device=self.device)
        if gabor is None:
            gabor = torch.zeros((n_items, pools_y, pools_x, 2, p.n_gabor), dtype=torch.float32,
                                device=self.device)
        p.process_dev(sig.data_ptr(), _SIG_DTYPES[sig.dtype], items_dev.data_ptr(), n_items,
                      mel.data_ptr(), pools_y, pools_x, gabor.data_ptr(), self._stream())
        return mel, gabor


def allgather_features(local, world_size, group=None):
    """The one collective of the path: reassemble [B/G, ...] slabs into [B, ...] on every rank
    (torch.distributed all_gather_into_tensor == ncclAllGather; RCCL on ROCm, gloo on CPU)."""
    import torch.distributed as dist
    out = torch.empty((local.shape[0] * world_size,) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    if dist.get_backend(group) == "gloo":
        parts = list(out.chunk(world_size, 0))
        dist.all_gather(parts, local.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out
