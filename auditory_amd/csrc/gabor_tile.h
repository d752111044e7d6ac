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

// ---- one output position in the plan's compute type throughout (what k_gabor runs per thread, gabor.hip; and what the last
// tile of an item runs per lane behind its ticket, melspec_w20.hip k_melspec_w20_gabor) --------------------------------------
// The position keeps up to 8 filter sums in registers, so every mel value it loads feeds 8 multiply-adds.  The taps are
// indexed only by loop counters, i.e. wave-uniform: `kf` must be a KERNEL PARAMETER (or derived from one by constants) for
// them to come through the scalar path.  The mel matrix is addressed by the reference's flat offset (f + ff) * cols + (t + ft)
// (etensor has no per-dimension bounds check), NaN inputs read as 0.5 (gabor.go:278-280), and the result is rectified into
// the on / off pair with the 2-D / 4-D index maps of :286-309.  Every multiply-add is spelled as a fused one: the same bits
// under any contraction setting of the including file.
constexpr int kGaborChunk = 8;

// four floats at 4-byte alignment
struct __attribute__((packed, aligned(4))) GaborF4u {
    float x, y, z, w;
};

// rank-4 output [PY, PX, 2, 8] with exactly 2 x 8 units per pool and 8 filters in the chunk: the on / off values of one
// position are 16 consecutive floats, 64-byte aligned when the tensor is
template <typename TT>
__device__ __forceinline__ void gabor_store_pair_block(float* cell, const TT (&acc)[8], TT gain) {
    float on[8], off[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const bool pos = acc[c] >= TT(0);
        const float act = float(gain * (acc[c] < TT(0) ? -acc[c] : acc[c]));
        on[c] = pos ? act : 0.f;
        off[c] = pos ? 0.f : act;
    }
    float4* c4 = reinterpret_cast<float4*>(cell);
    c4[0] = float4{on[0], on[1], on[2], on[3]};
    c4[1] = float4{on[4], on[5], on[6], on[7]};
    c4[2] = float4{off[0], off[1], off[2], off[3]};
    c4[3] = float4{off[4], off[5], off[6], off[7]};
}

// gabor.go:283-309: the sums of filters g0 .. g0 + gc - 1 at position (f_idx, t_idx), rectified into the on / off pair and
// stored with the 2-D / 4-D index maps
template <typename TT>
__device__ __forceinline__ void gabor_emit(const GaborArgs& a, float* out, int f_idx, int t_idx, int g0, int gc, int NG,
                                           const TT (&acc)[kGaborChunk]) {
    const TT gain = TT(a.gain);
    if (a.rank == 4 && a.d2 == 2 && a.d3 == 8 && NG == 8 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) {
        gabor_store_pair_block<TT>(out + (size_t(f_idx) * a.d1 + t_idx) * 16, acc, gain);
        return;
    }
#pragma unroll
    for (int c = 0; c < kGaborChunk; ++c) {
        if (c >= gc) break;
        const int flt = g0 + c;
        const bool pos = acc[c] >= TT(0);
        const float act = float(gain * (acc[c] < TT(0) ? -acc[c] : acc[c]));
        size_t o_on, o_off;
        if (a.rank == 2) {
            const int y = f_idx * 2;
            const int x = a.by_time ? t_idx + a.t_max_strides * flt : flt + t_idx * NG;
            o_on = size_t(y) * a.d1 + x;
            o_off = size_t(y + 1) * a.d1 + x;
        } else {
            const size_t cell = (size_t(f_idx) * a.d1 + t_idx) * a.d2;
            o_on = cell * a.d3 + flt;
            o_off = (cell + 1) * a.d3 + flt;
        }
        out[o_on] = pos ? act : 0.f;
        out[o_off] = pos ? 0.f : act;
    }
}

// KSX, KSY, KNG > 0: compile-time filter geometry (the reference's default 9 x 9 x 8 set gets fully unrolled taps and no
// group loop); 0: taken from the arguments at run time.  `mel` / `out`: the ITEM's matrix and output tensor; r: the
// position's index f_idx * nT + t_idx.
template <typename TT, int KSX, int KSY, int KNG>
__device__ __forceinline__ void gabor_position(const GaborArgs& a, const TT* __restrict__ kf, const float* __restrict__ mel,
                                               float* out, int r) {
    const int f_idx = r / a.nT, t_idx = r - f_idx * a.nT;
    const int f = f_idx * a.sty, t = t_idx * a.stx;
    const int SX = KSX > 0 ? KSX : a.SX, SY = KSY > 0 ? KSY : a.SY, NG = KNG > 0 ? KNG : a.nG;
    const int area = SX * SY;
    for (int g0 = 0; g0 < NG; g0 += kGaborChunk) {
        TT acc[kGaborChunk];
#pragma unroll
        for (int c = 0; c < kGaborChunk; ++c) acc[c] = TT(0);
        const int gc = min(kGaborChunk, NG - g0);
        auto tap_val = [&](float mv, int ff, int ft) {
            if (mv != mv) mv = 0.5f;  // math.IsNaN -> .5
            const TT v = TT(mv);
            const TT* tap = kf + size_t(g0) * area + ff * SX + ft;
#pragma unroll
            for (int c = 0; c < kGaborChunk; ++c)
                if (c < gc) acc[c] = mad(tap[size_t(c) * area], v, acc[c]);
        };
        auto tap_row = [&](const float* row, int ff, int ft) { tap_val(row[ft], ff, ft); };
        for (int ff = 0; ff < SY; ++ff) {
            const float* row = mel + size_t(f + ff) * a.cols + t;
            if constexpr (KSX == 9) {
                // nine consecutive floats as two 16-byte loads (4-byte aligned: the hardware takes unaligned vector loads)
                // and one 4-byte load: a third of the load instructions, the same cache lines
                const GaborF4u lo4 = *reinterpret_cast<const GaborF4u*>(row), hi4 = *reinterpret_cast<const GaborF4u*>(row + 4);
                const float mv[9] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w, row[8]};
#pragma unroll
                for (int ft = 0; ft < 9; ++ft) tap_val(mv[ft], ff, ft);
            } else if constexpr (KSX > 0) {
#pragma unroll
                for (int ft = 0; ft < KSX; ++ft) tap_row(row, ff, ft);
            } else {
                for (int ft = 0; ft < SX; ++ft) tap_row(row, ff, ft);
            }
        }
        gabor_emit<TT>(a, out, f_idx, t_idx, g0, gc, NG, acc);
    }
}

// ONE WAVE runs Convolve for a whole item (the ticket tail of melspec_w20.hip): 9 x 9 taps, 8 filters, float64 or float32 sums as
// gabor_position's (the same order of fused multiply-adds: the same bits).  A lone wave cannot hide memory latency behind other
// waves, so it does not read the matrix position by position: per step it takes the BAND of mel rows that `fpi` = 64 / nT
// consecutive fIdx need -- (fpi - 1) stY + 9 rows, flat, plus the 9 values a window may wrap into -- with all its
// loads in flight at once (16-byte pieces, kGaborBandPieces per lane), parks it in `lds` (the wave's own exchange region: its
// tile is done), and every lane computes one position (fIdx, tIdx) of the band from there while the next band's loads are in
// flight.  The caller has checked gabor_tail_fits().
constexpr int kGaborBandPieces = 5;  // 16-byte pieces per lane and band: 5 x 64 x 4 = 1280 floats (the default set: 12 x 104 + 9)
__host__ __device__ inline int gabor_tail_fpi(const GaborArgs& a) { return a.nT >= 64 ? 1 : 64 / a.nT; }
__host__ __device__ inline int gabor_tail_band_floats(const GaborArgs& a) {
    return ((gabor_tail_fpi(a) - 1) * a.sty + a.SY) * a.cols + a.SX;
}
__host__ __device__ inline bool gabor_tail_fits(const GaborArgs& a, size_t lds_bytes) {
    return a.SX == 9 && a.SY == 9 && a.nG == 8 && a.nT >= 1 && a.nT <= 64 && a.nF >= 1 &&
           gabor_tail_band_floats(a) <= kGaborBandPieces * 64 * 4 && size_t(gabor_tail_band_floats(a) + 4) * 4 <= lds_bytes &&
           (a.cols & 3) == 0 && ((size_t(a.rows) * a.cols) & 3) == 0;  // (bands and items start on 16-byte pieces when the tensor does)
}

template <typename TT>
__device__ __forceinline__ void gabor_item_wave(const GaborArgs& a, const TT* __restrict__ kf, const float* __restrict__ mel,
                                                float* out, float* lds, int lane) {
    const int fpi = gabor_tail_fpi(a);
    const int total = a.rows * a.cols;                       // floats of the item's matrix
    const int band = gabor_tail_band_floats(a);
    const bool vec = (reinterpret_cast<uintptr_t>(mel) & 15) == 0;
    float4 pre[kGaborBandPieces];
    auto fetch = [&](int f0) {                               // the band of fIdx f0 .. f0 + fpi - 1 into registers
        const int start = f0 * a.sty * a.cols;
#pragma unroll
        for (int k = 0; k < kGaborBandPieces; ++k) {
            const int o = 4 * (lane + 64 * k);                // float offset inside the band
            float4 v = float4{0.f, 0.f, 0.f, 0.f};
#if defined(AUD_EXP_TAIL) && AUD_EXP_TAIL == 3
            if (o < band && start + o < total && a.nT < 0) {
#else
            if (o < band && start + o < total) {
#endif
                const float* src = mel + start + o;
                if (vec && start + o + 3 < total) v = *reinterpret_cast<const float4*>(src);
                else {
                    v.x = src[0];
                    if (start + o + 1 < total) v.y = src[1];
                    if (start + o + 2 < total) v.z = src[2];
                    if (start + o + 3 < total) v.w = src[3];
                }
            }
            pre[k] = v;
        }
    };
    fetch(0);
    for (int f0 = 0; f0 < a.nF; f0 += fpi) {                 // wave-uniform
        wave_lds_fence();                                     // every lane is done with the previous band
#pragma unroll
        for (int k = 0; k < kGaborBandPieces; ++k) {
            const int o = 4 * (lane + 64 * k);
            if (o < band) *reinterpret_cast<float4*>(lds + o) = pre[k];  // (the last piece may end up to 3 floats past the band)
        }
        wave_lds_fence();
        if (f0 + fpi < a.nF) fetch(f0 + fpi);                // in flight behind this band's arithmetic
        const int fi = lane / a.nT, ti = lane - fi * a.nT;
        const bool has = fi < fpi && f0 + fi < a.nF;
        const float* win = lds + (has ? (fi * a.sty) * a.cols + ti * a.stx : 0);
        TT acc[kGaborChunk];
#pragma unroll
        for (int c = 0; c < kGaborChunk; ++c) acc[c] = TT(0);
#if defined(AUD_EXP_TAIL) && AUD_EXP_TAIL == 1
        if (a.nT < 0)
#endif
#pragma unroll 1  // (a row at a time: unrolled, the rows' values and the prefetched band want more registers than four waves per SIMD leave)
        for (int ff = 0; ff < 9; ++ff) {
            float mv[9];
#pragma unroll
            for (int ft = 0; ft < 9; ++ft) mv[ft] = win[ff * a.cols + ft];
#pragma unroll
            for (int ft = 0; ft < 9; ++ft) {
                float m = mv[ft];
                if (m != m) m = 0.5f;  // math.IsNaN -> .5
                const TT v = TT(m);
                const TT* tap = kf + ff * 9 + ft;
#pragma unroll
                for (int c = 0; c < kGaborChunk; ++c) acc[c] = mad(tap[c * 81], v, acc[c]);
            }
        }
#if defined(AUD_EXP_TAIL) && AUD_EXP_TAIL == 2
        if (has && a.nT < 0)
#else
        if (has)
#endif
            gabor_emit<TT>(a, out, f0 + fi, ti, 0, 8, 8, acc);
    }
}

__host__ __device__ inline size_t gabor_out_item_elems(const GaborArgs& a) {
    return a.rank == 2 ? size_t(a.d0) * a.d1 : size_t(a.d0) * a.d1 * a.d2 * a.d3;
}

}  // namespace aud
