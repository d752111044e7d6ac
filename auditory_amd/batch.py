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

    def segment(self, sig, items_dev, n_items, want_spectrum=True, deltas=True):
        """SndEnv.ProcessSegment with Mel.MFCC on (sound/sndenv.go:342-431) for n_items segments, device-resident:
        returns dict(mel, power, log_power, mfcc, deltas, delta_deltas, energy) of float32 tensors (None where not asked)"""
        p = self.plan
        if sig.device != self.device or not sig.is_contiguous():
            raise ValueError("signal must be a contiguous tensor on %s" % self.device)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=self.device)  # noqa: E731
        nc = p.mfcc_coefs
        out = dict(mel=new(n_items, p.nf, p.T), power=new(n_items, p.H, p.T) if want_spectrum else None,
                   log_power=new(n_items, p.H, p.T) if want_spectrum else None, mfcc=new(n_items, nc, p.T),
                   deltas=new(n_items, nc, p.T) if deltas else None, delta_deltas=new(n_items, nc, p.T) if deltas else None,
                   energy=new(n_items, p.T))
        nbytes = p.segment_workspace_bytes(n_items)
        ws = torch.empty(nbytes + 16, dtype=torch.uint8, device=self.device)
        ptr = lambda t: t.data_ptr() if t is not None else 0  # noqa: E731
        p.segment_dev(sig.data_ptr(), _SIG_DTYPES[sig.dtype], items_dev.data_ptr(), n_items, ptr(out["mel"]), ptr(out["power"]),
                      ptr(out["log_power"]), ptr(out["mfcc"]), ptr(out["deltas"]), ptr(out["delta_deltas"]), ptr(out["energy"]),
                      ws.data_ptr(), nbytes, self._stream())
        self._ws = ws  # stays alive until the next call (the launches are asynchronous)
        return out

    def kwta(self, gabor, params, act=None, pool=True, state=None, cycles=None, sum_order=0):
        """SndEnv.ApplyKwta on a device-resident gabor tensor [n, PY, PX, UY, UX] (settling starts from the raw
        values); `params` is an auditory_amd.kwta.KWTA.  Returns the settled tensor."""
        from . import kwta as kwta_mod
        if act is None:
            act = torch.empty_like(gabor)
        kwta_mod.kwta_batch_dev(params, gabor, act, pool=pool, state=state, cycles=cycles,
                                device=self.plan.ctx.device, sum_order=sum_order, stream=self._stream())
        return act

    def process_sndenv(self, sig, items_dev, n_items, pools_y, pools_x, params, state=None):
        """ProcessSegment + ApplyGabor with Kwta.On for every item, device-resident end to end:
        returns (mel, raw gabor, settled gabor)."""
        mel, gab = self.process(sig, items_dev, n_items, pools_y, pools_x)
        return mel, gab, self.kwta(gab, params, state=state)


def allgather_features(local, world_size, group=None, n_total=None):
    """The one collective of the path: reassemble the per-rank [B_r, ...] slabs into [B, ...] on
    every rank (torch.distributed all_gather_into_tensor == ncclAllGather; RCCL on ROCm, gloo on
    CPU).  Shards follow shard_range(); when B does not divide evenly the short shards are padded
    to the longest one for the collective and the padding is dropped afterwards."""
    import torch.distributed as dist
    if n_total is None:
        n_total = local.shape[0] * world_size
    counts = [shard_range(n_total, r, world_size) for r in range(world_size)]
    longest = max(hi - lo for lo, hi in counts)
    rank = dist.get_rank(group)
    assert local.shape[0] == counts[rank][1] - counts[rank][0], "local slab does not match shard_range"
    send = local.contiguous()
    if send.shape[0] < longest:
        pad = torch.zeros((longest - send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype,
                          device=send.device)
        send = torch.cat([send, pad], 0)
    out = torch.empty((longest * world_size,) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    if dist.get_backend(group) == "gloo":
        dist.all_gather(list(out.chunk(world_size, 0)), send, group=group)
    else:
        dist.all_gather_into_tensor(out, send, group=group)
    if longest * world_size == n_total:
        return out
    parts = [out[r * longest:r * longest + (hi - lo)] for r, (lo, hi) in enumerate(counts)]
    return torch.cat(parts, 0)


class DirectGather:
    """The path's reassembly as direct device-to-device copies (aud_gather_*, include/auditory_hip.h): every rank pushes
    its slab into slot `rank` of every rank's receive area, one copy per peer on its own stream -- on a fully connected
    xGMI node one link each, where a ring all-gather is per-link bound over G - 1 serial hops -- and stores
    its step number into the peer's arrival flag behind the copy; `wait` queues the poll for every peer's flag, behind which
    the step's slab is complete (what ncclAllGather's return means).  The receive area [2, n_ranks, slab_floats] float32 --
    consecutive steps alternate between its two slabs -- belongs to the library (it must be exportable between processes);
    `recv` is a torch view of it."""

    def __init__(self, ctx, n_ranks, rank, slab_floats):
        import ctypes as C
        self.ctx, self.lib, self.n_ranks, self.rank, self.slab = ctx, ctx.lib, int(n_ranks), int(rank), int(slab_floats)
        ptr, handle = C.c_void_p(), C.create_string_buffer(128)
        ctx.check(self.lib.aud_gather_create(ctx.handle, self.n_ranks, self.rank, self.slab, C.byref(ptr), handle))
        self.recv_ptr, self.handle = ptr.value, handle.raw

    def open_peers(self, handles):
        """handles[p] = rank p's 128-byte handle pair (this rank's own entry is ignored)"""
        for p, h in enumerate(handles):
            if p != self.rank:
                self.ctx.check(self.lib.aud_gather_open_peer(self.ctx.handle, p, bytes(h)))

    def exchange(self, group=None):
        """distribute the handles over torch.distributed (any backend) and map every peer's buffer"""
        import torch.distributed as dist
        handles = [None] * self.n_ranks
        dist.all_gather_object(handles, self.handle, group=group)
        self.open_peers(handles)

    def allgather(self, send_ptr, count, stream=0):
        """push `count` floats at device address send_ptr into the step's slab of every rank and signal each peer; returns the
        slab index (0 / 1) this step uses.  When `stream` has passed the call this rank's pushes are done; `wait` is the
        arrival side"""
        import ctypes as C
        which = C.c_int(-1)
        self.ctx.check(self.lib.aud_allgather_direct_dev(self.ctx.handle, send_ptr, int(count), C.byref(which), stream))
        return which.value

    def wait(self, stream=0):
        """queue the (bounded) poll for every peer's arrival flag of the last step on `stream`: behind it the step's slab is
        complete on this rank"""
        self.ctx.check(self.lib.aud_gather_wait_dev(self.ctx.handle, stream))

    def timeouts(self):
        """waits that ran into their poll bound so far (synchronises the device)"""
        import ctypes as C
        n = C.c_int(0)
        self.ctx.check(self.lib.aud_gather_timeouts(self.ctx.handle, C.byref(n)))
        return n.value

    def flags_fine(self):
        """1: the arrival flags live in fine-grained (coherent) device memory; 0: ordinary device memory (only with
        AUD_GATHER_COARSE_FLAGS=1 in the environment)"""
        return int(self.lib.aud_gather_flags_fine(self.ctx.handle))

    def recv(self, device):
        """the receive area as a [2, n_ranks, slab_floats] float32 tensor (no copy): slab `allgather()` returned"""
        if torch.device(device).type == "cpu":  # (tests' CPU thread-emulator build: "device" memory is host memory)
            import ctypes as C
            return torch.from_numpy(np.ctypeslib.as_array(C.cast(self.recv_ptr, C.POINTER(C.c_float)),
                                                          shape=(2, self.n_ranks, self.slab)))

        class _View:
            pass

        v = _View()
        v.__cuda_array_interface__ = {"shape": (2, self.n_ranks, self.slab), "typestr": "<f4", "data": (self.recv_ptr, False),
                                      "version": 2}
        self._keep = v
        return torch.as_tensor(v, device=device)

    def close(self):
        self.ctx.check(self.lib.aud_gather_destroy(self.ctx.handle))


class HostGather:
    """SURVEY 8(e)'s alternative for consumers that want the features on the HOST: no collective -- every rank copies its
    slab over its own PCIe link into ITS slot of one host buffer that all ranks of the node map.  The buffer
    [slabs, n_ranks, slab_floats] float32 is POSIX shared memory (`/dev/shm/<name>`: rank 0 creates it, the others map it),
    registered with the HIP runtime on GPU ranks so that `put` is an asynchronous device-to-host copy on the caller's
    stream (capturable).  When every rank's stream has passed its `put` (the caller's barrier, or an event per rank), slab
    `s` of `view()` is the batch's full [B, ...] tensor in rank order -- rank 0 usually being the one that reads it.
    Floor per step: one slab at the PCIe rate (64 GB/s: 133 us for configs[2]'s 8.5 MB), against one slab per xGMI link for the
    device all-gather -- this mode is about WHERE the result is wanted, not about speed."""

    def __init__(self, name, n_ranks, rank, slab_floats, device="cpu", slabs=2, timeout=60.0):
        import os
        import time
        self.n_ranks, self.rank, self.slab, self.slabs = int(n_ranks), int(rank), int(slab_floats), int(slabs)
        if not (0 <= self.rank < self.n_ranks) or self.slab <= 0 or self.slabs <= 0:
            raise capi.AuditoryError(capi.AUD_EINVAL, "HostGather: bad rank / slab")
        self.path = os.path.join("/dev/shm", name)
        floats = self.slabs * self.n_ranks * self.slab
        self.nbytes = 4 * floats
        if self.rank == 0:
            fd = os.open(self.path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            try:
                os.ftruncate(fd, self.nbytes)
            finally:
                os.close(fd)
        else:
            t0 = time.monotonic()
            while not (os.path.exists(self.path) and os.path.getsize(self.path) == self.nbytes):
                if time.monotonic() - t0 > timeout:
                    raise capi.AuditoryError(capi.AUD_EINVAL, "HostGather: rank 0's buffer %s did not appear" % self.path)
                time.sleep(0.01)
        self.buf = torch.from_file(self.path, shared=True, size=floats, dtype=torch.float32).view(self.slabs, self.n_ranks, self.slab)
        self.device = torch.device(device)
        self.registered = False
        if self.device.type == "cuda":
            rc = torch.cuda.cudart().cudaHostRegister(self.buf.data_ptr(), self.nbytes, 0)
            if int(rc) != 0:
                self.close()
                raise capi.AuditoryError(capi.AUD_EHIP, "HostGather: the runtime refused to register the shared buffer (%s)" % rc)
            self.registered = True

    @classmethod
    def create(cls, n_ranks, rank, slab_floats, device="cpu", group=None, slabs=2):
        """agree on the buffer's name over torch.distributed (any backend), map it on every rank, and return when all have"""
        import os
        import torch.distributed as dist
        name = [None]
        if rank == 0:
            name[0] = "auditory_hip_gather_%d_%s" % (os.getpid(), os.urandom(4).hex())
        if n_ranks > 1:
            dist.broadcast_object_list(name, src=0, group=group)
        g = cls(name[0], n_ranks, rank, slab_floats, device, slabs)
        if n_ranks > 1:
            dist.barrier(group=group)
        return g

    def put(self, local, slab=0):
        """queue the copy of this rank's [B_r, ...] float32 tensor into its slot of slab `slab` on the current stream"""
        flat = local.reshape(-1)
        if flat.dtype != torch.float32 or flat.numel() > self.slab or not (0 <= slab < self.slabs):
            raise capi.AuditoryError(capi.AUD_EINVAL, "HostGather.put: float32 tensor of at most %d elements" % self.slab)
        self.buf[slab, self.rank, :flat.numel()].copy_(flat, non_blocking=True)

    def view(self, slab=0):
        """[n_ranks, slab_floats] float32 host tensor (no copy): rank r's slab in row r"""
        return self.buf[slab]

    def close(self):
        import os
        if self.registered:
            torch.cuda.cudart().cudaHostUnregister(self.buf.data_ptr())
            self.registered = False
        self.buf = None
        if self.rank == 0 and os.path.exists(self.path):
            os.unlink(self.path)
