// agabor.Convolve (agabor/gabor.go:225-315) for a batch of mel matrices.
//
// One thread owns one output position (item, fIdx, tIdx) and keeps up to 8 filter sums in
// registers, so every mel sample it loads feeds 8 FMAs.  The filter taps are indexed only
// by loop counters, i.e. wave-uniform, and come in through the scalar path.  The mel
// matrix is addressed by the reference's flat offset (f+ff)*cols + (t+ft) (etensor has no
// per-dimension bounds check), NaN inputs read as 0.5 (:278-280), and the result is
// rectified into the on/off pair with the 2-D / 4-D index maps of :286-309.
#include <cstring>

#include "gabor_tile.h"

// (gabor_tile.h pulls in device_common.h, whose kernels spell their fused multiply-adds out under `fp contract(off)`;
// k_gabor below keeps the compiler's contraction -- acc += tap * v is one v_fma, as in the builds before the tile kernel)
#pragma clang fp contract(fast)

namespace aud {
namespace {

constexpr int kChunk = 8;

// four floats at 4-byte alignment
struct __attribute__((packed, aligned(4))) F4u {
    float x, y, z, w;
};

// rank-4 output [PY, PX, 2, 8] with exactly 2 x 8 units per pool and 8 filters in the chunk: the on / off values of one
// position are 16 consecutive floats, 64-byte aligned when the tensor is
template <typename TT>
__device__ __forceinline__ void store_pair_block(float* cell, const TT (&acc)[8], TT gain) {
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

// KSX, KSY, KNG > 0: compile-time filter geometry (the reference's default 9 x 9 x 8 set gets fully
// unrolled taps and no group loop); 0: taken from the arguments at run time.
constexpr int kGaborBlock = 256;
template <typename TT, int KSX, int KSY, int KNG>
__global__ __launch_bounds__(kGaborBlock) void k_gabor(const GaborArgs a) {
    const int per_item = a.nF * a.nT;
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= int64_t(a.n_items) * per_item) return;
    const int item = int(gid / per_item);
    const int r = int(gid - int64_t(item) * per_item);
    const int f_idx = r / a.nT, t_idx = r - f_idx * a.nT;
    const int f = f_idx * a.sty, t = t_idx * a.stx;
    const int SX = KSX > 0 ? KSX : a.SX, SY = KSY > 0 ? KSY : a.SY, NG = KNG > 0 ? KNG : a.nG;

    const float* __restrict__ mel = a.mel + size_t(item) * a.rows * a.cols;
    const TT* __restrict__ kf = static_cast<const TT*>(a.k);
    const int area = SX * SY;
    const TT gain = TT(a.gain);

    size_t out_item;
    if (a.rank == 2)
        out_item = size_t(a.d0) * a.d1;
    else
        out_item = size_t(a.d0) * a.d1 * a.d2 * a.d3;
    float* out = a.out + size_t(item) * out_item;

    for (int g0 = 0; g0 < NG; g0 += kChunk) {
        TT acc[kChunk];
#pragma unroll
        for (int c = 0; c < kChunk; ++c) acc[c] = TT(0);
        const int gc = min(kChunk, NG - g0);
        auto tap_val = [&](float mv, int ff, int ft) {
            if (mv != mv) mv = 0.5f;  // math.IsNaN -> .5
            const TT v = TT(mv);
            const TT* tap = kf + size_t(g0) * area + ff * SX + ft;
#pragma unroll
            for (int c = 0; c < kChunk; ++c)
                if (c < gc) acc[c] += tap[size_t(c) * area] * v;
        };
        auto tap_row = [&](const float* row, int ff, int ft) { tap_val(row[ft], ff, ft); };
        // three rows at a time: the next rows' loads are in flight behind a row's multiply-adds (rolled: every row's loads are
        // waited for in full, 15.47-15.55 us per configs[3] step; by three 15.35-15.39; all nine 15.96-16.14: round 5, one box)
#pragma unroll 3
        for (int ff = 0; ff < SY; ++ff) {
            const float* row = mel + size_t(f + ff) * a.cols + t;
            if constexpr (KSX == 9) {
                // nine consecutive floats as two 16-byte loads (4-byte aligned: the hardware takes unaligned vector loads)
                // and one 4-byte load: a third of the load instructions, the same cache lines
                const F4u lo4 = *reinterpret_cast<const F4u*>(row), hi4 = *reinterpret_cast<const F4u*>(row + 4);
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
        if (a.rank == 4 && a.d2 == 2 && a.d3 == 8 && NG == 8 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) {
            store_pair_block<TT>(out + (size_t(f_idx) * a.d1 + t_idx) * 16, acc, gain);
            continue;
        }
#pragma unroll
        for (int c = 0; c < kChunk; ++c) {
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
}

// The LDS-staged form (the default wherever an item's mel matrix and the taps fit 64 KB of LDS): one workgroup per item
// copies the matrix [rows * cols] (NaN read as 0.5, gabor.go:278-280) into LDS with 16-byte loads, then its four waves
// run gabor_from_lds (gabor_tile.h) -- 64 positions x 4 filters per unit, float32 taps through the scalar path, float32 row
// sums added in the compute type.  Against k_gabor above (one thread per position, 81 strided reads of
// the matrix from memory, all-float64 multiply-adds in float64 plans): 0.65 of the vector-ALU cycles and no dependent
// memory loads inside the tap loops.
constexpr int kLdsThreads = 256;

template <typename TT>
__global__ __launch_bounds__(kLdsThreads) void k_gabor_lds(const GaborArgs a, const ItemArgs g, const float* __restrict__ k32) {
    float* melL = reinterpret_cast<float*>(dyn_lds());
    const int n_mel = a.rows * a.cols;
    const int tid = int(threadIdx.x), item = int(blockIdx.x);
    const float* __restrict__ src = a.mel + size_t(item) * n_mel;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        for (int i = 4 * tid; i < n_mel; i += 4 * kLdsThreads) {
            if (i + 3 < n_mel) {
                float4 v = *reinterpret_cast<const float4*>(src + i);
                v.x = v.x != v.x ? 0.5f : v.x;
                v.y = v.y != v.y ? 0.5f : v.y;
                v.z = v.z != v.z ? 0.5f : v.z;
                v.w = v.w != v.w ? 0.5f : v.w;
                *reinterpret_cast<float4*>(melL + i) = v;
            } else {
                for (int u = i; u < n_mel; ++u) melL[u] = src[u] != src[u] ? 0.5f : src[u];
            }
        }
    } else {
        for (int i = tid; i < n_mel; i += kLdsThreads) melL[i] = src[i] != src[i] ? 0.5f : src[i];
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (g.SX == 9 && g.SY == 9) gabor_from_lds<TT, 9, 9>(g, melL, k32, a.cols, item, wave, kLdsThreads / 64, lane);
    else gabor_from_lds<TT, 0, 0>(g, melL, k32, a.cols, item, wave, kLdsThreads / 64, lane);
}

}  // namespace

size_t gabor_lds_bytes(int rows, int cols, int nG, int SX, int SY) {
    (void)nG; (void)SX; (void)SY;  // (the taps come through the scalar path, not LDS)
    return (size_t((rows * cols + 3) & ~3) + 4) * sizeof(float);
}

hipError_t launch_gabor(const GaborArgs& a, int compute_dtype, hipStream_t st) {
    const int64_t total = int64_t(a.n_items) * a.nF * a.nT;
    if (total == 0) return hipSuccess;
    const size_t lds = gabor_lds_bytes(a.rows, a.cols, a.nG, a.SX, a.SY);
    // float64 plans default to k_gabor: float64 taps and float64 multiply-adds throughout, as FilterSet.Filters (etensor.Float64)
    // and fSum (float64) are in the reference (gabor.go:66, :272-283).  The LDS-staged kernel rounds the taps to float32 and sums a
    // row of them in float32 -- gabor taps cancel, so near fSum = 0 that can land a value in the other on / off channel than the
    // reference does; it is the float32 plans' default and a float64 plan's explicit opt-in (mode 0).  Cost of the conforming
    // choice beside the mel kernel: 15.3 against 15.1 us per 256 items (profiles/round4_cfg4_gabor_variants.txt).
    const int mode = a.mode < 0 ? (compute_dtype == AUD_F64 ? 1 : 0) : a.mode;
    if (a.k32 && mode != 1 && lds <= 64 * 1024 && int64_t(a.rows) * a.cols < (int64_t(1) << 24)) {
        ItemArgs g;
        std::memset(&g, 0, sizeof(g));
        g.k32 = a.k32;
        g.nG = a.nG;
        g.SX = a.SX;
        g.SY = a.SY;
        g.stx = a.stx;
        g.sty = a.sty;
        g.gain = a.gain;
        g.rank = a.rank;
        g.d0 = a.d0;
        g.d1 = a.d1;
        g.d2 = a.d2;
        g.d3 = a.d3;
        g.by_time = a.by_time;
        g.t_max_strides = a.t_max_strides;
        g.nT = a.nT;
        g.nF = a.nF;
        g.out = a.out;
        if (compute_dtype == AUD_F64) hipLaunchKernelGGL(k_gabor_lds<double>, dim3(unsigned(a.n_items)), dim3(kLdsThreads), lds, st, a, g, a.k32);
        else hipLaunchKernelGGL(k_gabor_lds<float>, dim3(unsigned(a.n_items)), dim3(kLdsThreads), lds, st, a, g, a.k32);
        return hipGetLastError();
    }
    const dim3 grid(unsigned((total + kGaborBlock - 1) / kGaborBlock));
    const bool dflt = a.SX == 9 && a.SY == 9 && a.nG == 8;  // processspeech.go:226-253
    if (compute_dtype == AUD_F64) {
        if (dflt) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gabor<double, 9, 9, 8>), grid, dim3(kGaborBlock), 0, st, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gabor<double, 0, 0, 0>), grid, dim3(kGaborBlock), 0, st, a);
    } else {
        if (dflt) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gabor<float, 9, 9, 8>), grid, dim3(kGaborBlock), 0, st, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gabor<float, 0, 0, 0>), grid, dim3(kGaborBlock), 0, st, a);
    }
    return hipGetLastError();
}

}  // namespace aud
