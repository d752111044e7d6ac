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

// (gabor_tile.h pulls in device_common.h: `fp contract(off)`, every fused multiply-add spelled out -- gabor_position's too, so
// that k_gabor and the ticket tail of melspec_w20.hip give the same bits)

namespace aud {
namespace {

// One thread owns one output position (item, fIdx, tIdx): gabor_position (gabor_tile.h), taps behind the kernel's argument struct.
template <typename TT, int KSX, int KSY, int KNG>
__global__ __launch_bounds__(256) void k_gabor(const GaborArgs a) {
    const int per_item = a.nF * a.nT;
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= int64_t(a.n_items) * per_item) return;
    const int item = int(gid / per_item);
    const int r = int(gid - int64_t(item) * per_item);
    gabor_position<TT, KSX, KSY, KNG>(a, static_cast<const TT*>(a.k), a.mel + size_t(item) * a.rows * a.cols,
                                      a.out + size_t(item) * gabor_out_item_elems(a), r);
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
    const dim3 grid(unsigned((total + 255) / 256));
    const bool dflt = a.SX == 9 && a.SY == 9 && a.nG == 8;  // processspeech.go:226-253
    if (compute_dtype == AUD_F64) {
        if (dflt) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gabor<double, 9, 9, 8>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gabor<double, 0, 0, 0>), grid, dim3(256), 0, st, a);
    } else {
        if (dflt) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gabor<float, 9, 9, 8>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gabor<float, 0, 0, 0>), grid, dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

}  // namespace aud
