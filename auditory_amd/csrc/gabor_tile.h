// agabor.Convolve (agabor/gabor.go:225-315) on ONE item's mel matrix held in LDS -- the second phase of the workgroup-per-
// item kernel (melspec_w20.hip k_melspec_w20_item): the waves that computed the item's frames reduce its [nf, T] mel matrix
// with the filter set where it lies, no second launch and no re-read from memory.
//
// A unit of work = (block of 64 output positions, quad of 4 filters): a lane owns one position (fIdx, tIdx) and keeps four
// filter sums; the taps are indexed by loop counters only, i.e. wave-uniform, and arrive through the scalar path.  The
// matrix is addressed by the reference's FLAT offset (f + ff) * cols + (t + ft) (etensor has no per-dimension bounds check:
// SURVEY Q10) -- the LDS copy is flat [rows * cols], so windows that wrap into the next mel row read what the Go code reads.
// NaN -> 0.5 (gabor.go:278-280) was applied when the value was stored (wave_mel_epilogue TOLDS).
//
// Arithmetic: a row of SX taps is summed in float32 (fused multiply-adds on float32 copies of the taps), the SY row sums are
// added in the plan's compute type.  For float64 plans that leaves ~1e-7 relative on an output against the all-float64 sum
// (the float32 rounding of the taps and of a 9-term row sum; the mel values themselves are the float32-stored ones either
// way), at 0.6 of the float64 multiply-adds' issue cost (tools/ubench/valu_rates.hip: v_fma_f64 4.4-4.6 cycles, v_fma_f32
// 2.6-2.8); float32 plans sum everything in float32.
#pragma once
#include "device_common.h"

namespace aud {

template <typename TT, int KSX, int KSY>
__device__ __forceinline__ void gabor_from_lds(const ItemArgs& g, const float* melL, int cols, int item, int wave, int n_waves,
                                               int lane) {
    const int SX = KSX > 0 ? KSX : g.SX, SY = KSY > 0 ? KSY : g.SY;
    const int per_item = g.nF * g.nT;
    const int blocks = (per_item + 63) >> 6, quads = (g.nG + 3) >> 2;
    const int units = blocks * quads;
    const int area = SX * SY;
    const TT gain = TT(g.gain);
    float* out_item = g.out + size_t(item) * g.d0 * g.d1 * 2 * g.nG;
    const bool vec_out = (g.nG & 3) == 0 && (reinterpret_cast<uintptr_t>(g.out) & 15) == 0;
    for (int u = wave; u < units; u += n_waves) {  // wave-uniform
        const int blk = u / quads, q = u - blk * quads;
        const int r = blk * 64 + lane;
        const bool has = r < per_item;
        const int rr = has ? r : per_item - 1;
        const int f_idx = rr / g.nT, t_idx = rr - f_idx * g.nT;
        const float* win = melL + (f_idx * g.sty) * cols + t_idx * g.stx;
        const int gc = g.nG - 4 * q < 4 ? g.nG - 4 * q : 4;  // filters of this quad (wave-uniform)
        const float* __restrict__ kq = g.k32 + size_t(4 * q) * area;
        TT acc[4] = {TT(0), TT(0), TT(0), TT(0)};
        for (int ff = 0; ff < SY; ++ff) {
            const float* row = win + ff * cols;
            float rs[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (KSX > 0) {
                float mv[KSX > 0 ? KSX : 1];
#pragma unroll
                for (int ft = 0; ft < KSX; ++ft) mv[ft] = row[ft];
#pragma unroll
                for (int ft = 0; ft < KSX; ++ft)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        rs[c] = fmaf(kq[(c < gc ? c : 0) * area + ff * SX + ft], mv[ft], rs[c]);
            } else {
                for (int ft = 0; ft < SX; ++ft) {
                    const float v = row[ft];
#pragma unroll
                    for (int c = 0; c < 4; ++c) rs[c] = fmaf(kq[(c < gc ? c : 0) * area + ff * SX + ft], v, rs[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += TT(rs[c]);
        }
        // gabor.go:283-309: on / off rectification into [fIdx, tIdx, 0 / 1, flt]
        float on[4], off[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool pos = acc[c] >= TT(0);
            const float act = float(gain * (acc[c] < TT(0) ? -acc[c] : acc[c]));
            on[c] = pos ? act : 0.f;
            off[c] = pos ? 0.f : act;
        }
        float* cell = out_item + (size_t(f_idx) * g.d1 + t_idx) * 2 * g.nG;
        if (has) {
            if (vec_out) {
                *reinterpret_cast<float4*>(cell + 4 * q) = float4{on[0], on[1], on[2], on[3]};
                *reinterpret_cast<float4*>(cell + g.nG + 4 * q) = float4{off[0], off[1], off[2], off[3]};
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < gc) {
                        cell[4 * q + c] = on[c];
                        cell[g.nG + 4 * q + c] = off[c];
                    }
            }
        }
    }
}

}  // namespace aud
