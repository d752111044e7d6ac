#!/usr/bin/env python3
"""Run ON THE GPU BOX: the seeded random-geometry case of the wave kernels (tests/parity_cases.py
case_random_wave_config: steps, segment lengths, borders, filter counts, masked tails, input levels, the fused segment tail
with a random coefficient count -- each against the oracle) for many more seeds than the test tier runs.

  python tools/soak_wave_fuzz.py [first_seed [n_seeds]]      default: seeds 48 .. 347 for N = 400, 512, 2048
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    import torch
    import parity_cases as PC
    from oracle import oracle as orc   # the checker
    assert torch.cuda.is_available(), "needs the GPU"
    bad = 0
    for N in (400, 512, 2048):
        for seed in range(first, first + count):
            try:
                PC.case_random_wave_config(orc, seed, N)
            except AssertionError as ex:
                bad += 1
                print("FAIL N=%d seed=%d: %s" % (N, seed, str(ex)[:400]), flush=True)
                if bad > 10:
                    raise SystemExit(1)
        print("N=%d: seeds %d..%d done" % (N, first, first + count - 1), flush=True)
    print("soak done, failures: %d" % bad)
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
