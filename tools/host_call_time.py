#!/usr/bin/env python3
"""Run ON THE GPU BOX: what the host-buffer entry point costs end to end (pageable float64 Go-tensor-like input copied in,
kernel, float32 result copied out and widened) -- the PCIe-inclusive rate DESIGN.md 6 quotes.  It is never bench.py's `value`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as W
from auditory_amd import capi, runtime, synth
from oracle import oracle as orc
oc=W.OracleCfg(orc,"cfg2_16k_n400_nf40")
L=oc.full_len(); n=256
sig,_=synth.batch(3,n,16000,oc.sr,row_len=L)
plan=W.product_plan(oc,capi.AUD_F64)
items=runtime.make_items(np.arange(n)*L,[L]*n,[0]*n)
for _ in range(3): plan.melspec_host(sig.ravel(),items)
t=time.perf_counter(); reps=20
for _ in range(reps): plan.melspec_host(sig.ravel(),items)
dt=(time.perf_counter()-t)/reps
print("aud_melspec_batch_host, 256 utterances of 1 s (float64 host signal %.1f MB in, float64 mel %.1f MB out): %.2f ms per call = %.0f audio-s/s" % (sig.nbytes/1e6, n*40*104*8/1e6, dt*1e3, n/dt))
