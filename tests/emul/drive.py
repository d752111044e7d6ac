"""TEST INFRASTRUCTURE ONLY: runs parity cases against a sanitizer build of the emulated kernels.
Invoked as a subprocess (the sanitizer runtime has to be LD_PRELOADed into python):
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) python tests/emul/drive.py asan quick
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import backend  # noqa: E402
import parity_cases as PC  # noqa: E402
from auditory_amd import capi  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    variant, which = sys.argv[1], sys.argv[2]
    by_name = {c[0]: c for c in PC.CASES}
    with backend.emulated(variant):
        if which == "quick":
            # small shapes: every kernel family, both compute types, every control path
            PC.case_melspec_vs_oracle(orc, ("sndenv_16k_n400_nf32", 0.25, 1, [0, 2]), capi.AUD_F32)
            PC.case_n512_variants(orc, capi.AUD_F32, seg_ms=100.0, dur=0.2, rows=1)
            PC.case_melspec_vs_oracle(orc, ("cfg2_16k_n512_nf40", 0.2, 1, [0]), capi.AUD_F64, seg_ms=100.0)
            PC.case_n512_odd_step_and_sample_types(orc, capi.AUD_F32)
            PC.case_n400_variants(orc, capi.AUD_F32, seg_ms=100.0, dur=0.25, rows=1, segs=(0, 1))
            PC.case_n2048_variants(orc, capi.AUD_F32, seg_ms=60.0, dur=0.12, rows=1)
            PC.case_melspec_vs_oracle(orc, ("odd_15k_n375_nf32", 0.2, 1, [0]), capi.AUD_F32)
            PC.case_zero_signal_and_empty_batch(orc)
            PC.case_prev_smooth(orc, "sndenv_16k_n400_nf32", capi.AUD_F32)
            PC.case_mfcc_tail(orc, "sndenv_16k_n400_nf32", capi.AUD_F32)
            PC.case_mfcc_tail(orc, "cfg2_16k_n512_nf40", capi.AUD_F64)        # fused tail, float64 scratch in the spectrum rows
            PC.case_input_levels(orc, "cfg2_16k_n400_nf40", capi.AUD_F64, quick=True)   # scaled and unscaled splits
            PC.case_gabor_4d_and_2d_vs_oracle(orc, capi.AUD_F32)
            # the workgroup-per-item kernel (tile loop, the item's mel matrix in LDS, Convolve behind its barrier), the LDS-staged
            # and per-position gabor kernels, a resident signal
            PC.case_process_fused_vs_oracle(orc, capi.AUD_F64, PC.HostMem(), name="sndenv_16k_n400_nf32", n=2, pools=(8, 4))
            PC.case_resident_signal(orc, capi.AUD_F32)
            PC.case_kwta_quick(orc)
            # round 5: the in-place Bluestein route (pair transform in float64, one frame per transform in float32; one padded
            # LDS buffer, stages through registers), the bin-per-lane spectrum outputs, and the direct all-gather's arrival flags
            # (system-scope atomics polled by one lane per peer: what ThreadSanitizer is for)
            PC.case_melspec_vs_oracle(orc, ("cfg1_44k_n1103_nf32", 0.12, 1, [0]), capi.AUD_F64)
            PC.case_melspec_vs_oracle(orc, ("cfg1_44k_n1103_nf32", 0.12, 1, [0]), capi.AUD_F32)
            # round 6, second half: the chirp kernel is what the two cfg1 float64 cases above now run; the any-N kernel's pair route
            # behind its option, smooth lengths in place (eight frames per workgroup as one batched transform; radix 7), the direct kernel
            PC.case_melspec_vs_oracle(orc, ("cfg1_44k_n1103_nf32", 0.12, 1, [0]), capi.AUD_F64, options={"chirp_kernel": 0})
            PC.case_melspec_vs_oracle(orc, ("rate_8k_n200_nf32", 0.25, 1, [0, 1]), capi.AUD_F64)
            PC.case_melspec_vs_oracle(orc, ("win20_44k_n882_nf32", 0.12, 1, [0]), capi.AUD_F32)
            PC.case_direct_kernel(orc, 5123, capi.AUD_F64, sig_kind="int16")
            PC.case_direct_gather_three_ranks()
            # round 6: the exact resident Signal (host shadow, block compares, partial uploads: memcmp / memcpy bounds are what
            # AddressSanitizer is for; the direct gather case above now ends in the sticky time-out with its NaN fill)
            if variant != "tsan":   # (host-side, single-threaded logic: the address sanitizer's business)
                PC.case_sndenv_resident_signal_staleness(orc)
        else:
            PC.case_melspec_vs_oracle(orc, by_name[which], capi.AUD_F32)
    print("DRIVE-OK", variant, which)


if __name__ == "__main__":
    main()
