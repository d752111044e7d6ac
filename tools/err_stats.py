"""prints error statistics of the GPU mel output vs the oracle for cfg2 (diagnostic)"""
import sys, os
import numpy as np, torch
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import workloads as W
from auditory_amd import capi, runtime, synth
from auditory_amd.batch import BatchProcessor
from oracle import oracle as orc
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_16k_n512_nf40"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
oc = W.OracleCfg(orc, name)
L = oc.full_len()
sig, _ = synth.batch(2, B, 16000, oc.sr, row_len=L)
rc, ref, _ = orc.process_batch(oc.sp, oc.d, oc.m, oc.bins, oc.filt, sig.ravel(), np.arange(B) * L, np.full(B, L), np.zeros(B))
for cdt in (capi.AUD_F32, capi.AUD_F64):
    plan = W.product_plan(oc, cdt)
    bp = BatchProcessor(plan, "cuda:0")
    items = bp.upload_items(runtime.make_items(np.arange(B) * L, [L] * B, [0] * B))
    mel = bp.melspec(torch.from_numpy(sig.astype(np.float32)).cuda().view(-1), items, B).cpu().numpy()
    err = np.abs(mel - ref) / np.maximum(1, np.abs(ref))
    q = np.quantile(err, [0.5, 0.99, 0.9999, 1.0])
    w = np.unravel_index(np.argmax(err), err.shape)
    print(name, "cdt", cdt, plan.kernel_name, "err q50/99/99.99/max", q, "worst at", w, "ref", ref[w], "per-filter max", err.max(axis=(0, 2))[:8])
    plan.close()
