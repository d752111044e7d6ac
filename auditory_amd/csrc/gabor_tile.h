// agabor.Convolve (agabor/gabor.go:225-315) on ONE item's mel matrix held in LDS -- the second phase of the workgroup-per-
// item kernel (melspec_w20.hip k_melspec_w20_item): the waves that computed the item's frames reduce its [nf, T] mel matrix
// with the filter set where it lies, no second launch and no re-read from memory.
//
// A unit of work = (block of 64 output positions, quad of 4 filters): a lane owns one position (fIdx, tIdx) and keeps four
// filter sums; the taps are indexed by loop counters only, i.e. wave-uniform, and must arrive through the SCALAR path: `kS`
// is the float32 tap table [nG][SY][SX] in memory behind a `const float* __restrict__` KERNEL PARAMETER, which is what lets
// the compiler issue s_load for them (multiply-adds then take the tap as their scalar operand: no vector register, no
// LDS traffic).  Two other routes were built and measured on 256 items per launch beside the mel kernel: taps behind a
// pointer inside an argument struct come through per-lane VECTOR loads with a memory wait in every row (+7 us per step);
// taps as broadcast reads of an LDS copy put 210 LDS instructions per unit on a pipe the mel kernel already keeps 60 %
// busy (+1 us against the scalar route).  The
// matrix is addressed by the reference's FLAT offset (f + ff) * cols + (t + ft) (etensor has no per-dimension bounds check:
// SURVEY Q10) -- the LDS copy is flat [rows * cols], so windows that wrap into the next mel row read what the Go code reads.
// NaN -> 0.5 (gabor.go:278-280) was applied when the value was stored (wave_mel_epilogue TOLDS).
//
// Arithmetic: a row of SX taps is summed in float32 (fused multiply-adds on float32 copies of the taps, two filters per
// instruction: pk_fma_tap), the SY row sums are
// added in the plan's compute type.  For float64 plans that leaves ~1e-7 relative on an output against the all-float64 sum
// (the float32 rounding of the taps and of a 9-term row sum; the mel values themselves are the float32-stored ones either
// way), at 0.6 of the float64 multiply-adds' issue cost (tools/ubench/valu_rates.hip: v_fma_f64 4.4-4.6 cycles, v_fma_f32
// 2.6-2.8); float32 plans sum everything in float32.
#pragma once
#include "device_common.h"

namespace aud {

// Two float32 multiply-adds per instruction: acc.{x,y} += taps.{x,y} * (LO ? m.x : m.y) as ONE v_pk_fma_f32 -- the tap pair is the
// instruction's scalar operand (an aligned SGPR pair out of an s_load), the mel value one half of a register pair as
// ds_read2_b32 delivered it, broadcast to both halves by the operand-select bits.  In these kernels every vector instruction
// costs about four SIMD cycles of issue whatever it computes, so halving the instruction count of the tap
// loop is what counts.  Each half is a true fused multiply-add: the results equal fmaf's bit for bit (the CPU thread emulator
// runs the fmaf form).
#if defined(__clang__)
typedef float gabor_f2 __attribute__((ext_vector_type(2)));
#else  // (the CPU thread emulator is built with g++)
struct alignas(8) gabor_f2 {
    float x, y;
};
#endif
template <bool LO>
__device__ __forceinline__ void pk_fma_tap(gabor_f2& acc, gabor_f2 taps, gabor_f2 m) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (LO) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(taps), "v"(m));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(taps), "v"(m));
#else
    const float v = LO ? m.x : m.y;
    acc.x = fmaf(taps.x, v, acc.x);
    acc.y = fmaf(taps.y, v, acc.y);
#endif
}

// The tap table of these kernels is QUAD-INTERLEAVED: kS [quads][SY][SX][4] float32 (filters past nG are zero rows), so that
// the pairs (filter 4q, 4q+1) and (4q+2, 4q+3) of one tap position are adjacent SGPR pairs of one scalar load.
template <typename TT, int KSX, int KSY>
__device__ __forceinline__ void gabor_from_lds(const ItemArgs& g, const float* melL, const float* __restrict__ kS, int cols, int item,
                                               int wave, int n_waves, int lane) {
    const int SX = KSX > 0 ? KSX : g.SX, SY = KSY > 0 ? KSY : g.SY;
    const int per_item = g.nF * g.nT;
    const int blocks = (per_item + 63) >> 6, quads = (g.nG + 3) >> 2;
    const int units = blocks * quads;
    const int area = SX * SY;
    const TT gain = TT(g.gain);
    const bool rank4 = g.rank == 4;
    float* out_item = g.out + size_t(item) * (rank4 ? size_t(g.d0) * g.d1 * g.d2 * g.d3 : size_t(g.d0) * g.d1);
    // [.., 2, nG] with nG a multiple of 4 and a 16-byte aligned tensor: a quad's on / off values are two 16-byte stores
    const bool vec_out = rank4 && g.d2 == 2 && g.d3 == g.nG && (g.nG & 3) == 0 && (reinterpret_cast<uintptr_t>(g.out) & 15) == 0;
    for (int u = wave; u < units; u += n_waves) {  // wave-uniform
        const int blk = u / quads, q = u - blk * quads;
        const int r = blk * 64 + lane;
        const bool has = r < per_item;
        const int rr = has ? r : per_item - 1;
        const int f_idx = rr / g.nT, t_idx = rr - f_idx * g.nT;
        const float* win = melL + (f_idx * g.sty) * cols + t_idx * g.stx;
        const int gc = g.nG - 4 * q < 4 ? g.nG - 4 * q : 4;  // filters of this quad (wave-uniform)
        const gabor_f2* __restrict__ kq = reinterpret_cast<const gabor_f2*>(kS + size_t(q) * area * 4);  // [SY][SX][2 pairs]
        TT acc[4] = {TT(0), TT(0), TT(0), TT(0)};
        auto do_row = [&](int ff) {
            const float* row = win + ff * cols;
            gabor_f2 r01 = {0.f, 0.f}, r23 = {0.f, 0.f};
            if constexpr (KSX > 0) {
                gabor_f2 mv[(KSX + 1) / 2 > 0 ? (KSX + 1) / 2 : 1];
#pragma unroll
                for (int h = 0; h < (KSX + 1) / 2; ++h) mv[h] = gabor_f2{row[2 * h], 2 * h + 1 < KSX ? row[2 * h + 1] : 0.f};
#pragma unroll
                for (int ft = 0; ft < KSX; ++ft) {
                    const gabor_f2 t01 = kq[(ff * KSX + ft) * 2], t23 = kq[(ff * KSX + ft) * 2 + 1];
                    if (ft & 1) {
                        pk_fma_tap<false>(r01, t01, mv[ft / 2]);
                        pk_fma_tap<false>(r23, t23, mv[ft / 2]);
                    } else {
                        pk_fma_tap<true>(r01, t01, mv[ft / 2]);
                        pk_fma_tap<true>(r23, t23, mv[ft / 2]);
                    }
                }
            } else {
                for (int ft = 0; ft < SX; ++ft) {
                    const gabor_f2 v = {row[ft], 0.f};
                    pk_fma_tap<true>(r01, kq[(ff * SX + ft) * 2], v);
                    pk_fma_tap<true>(r23, kq[(ff * SX + ft) * 2 + 1], v);
                }
            }
            acc[0] += TT(r01.x);
            acc[1] += TT(r01.y);
            acc[2] += TT(r23.x);
            acc[3] += TT(r23.y);
        };
        if constexpr (KSY > 0) {  // unrolled: the next rows' taps and values are requested while this row's multiply-adds issue
#pragma unroll
            for (int ff = 0; ff < KSY; ++ff) do_row(ff);
        } else {
            for (int ff = 0; ff < SY; ++ff) do_row(ff);
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
        if (has) {
            if (vec_out) {
                float* cell = out_item + (size_t(f_idx) * g.d1 + t_idx) * 2 * g.nG;
                *reinterpret_cast<float4*>(cell + 4 * q) = float4{on[0], on[1], on[2], on[3]};
                *reinterpret_cast<float4*>(cell + g.nG + 4 * q) = float4{off[0], off[1], off[2], off[3]};
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < gc) {
                        const int flt = 4 * q + c;
                        size_t o_on, o_off;
                        if (rank4) {  // [fIdx, tIdx, 0 / 1, flt] (gabor.go:300-307)
                            const size_t cell = (size_t(f_idx) * g.d1 + t_idx) * g.d2;
                            o_on = cell * g.d3 + flt;
                            o_off = (cell + 1) * g.d3 + flt;
                        } else {      // two rows per frequency stride, filters along x or by time (gabor.go:286-298)
                            const int x = g.by_time ? t_idx + g.t_max_strides * flt : flt + t_idx * g.nG;
                            o_on = size_t(2 * f_idx) * g.d1 + x;
                            o_off = size_t(2 * f_idx + 1) * g.d1 + x;
                        }
                        out_item[o_on] = on[c];
                        out_item[o_off] = off[c];
                    }
            }
        }
    }
}

}  // namespace aud
