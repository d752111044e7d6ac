// Shared by the wave-autonomous kernel files (melspec_w16.hip, melspec_w20.hip, melspec_w64.hip) and their host-side
// dispatch (melspec_wave.hip).
#pragma once
#include "device_common.h"

namespace aud {

// what melspec_wave.hip needs from a kernel file: the instantiation for (compute type, sample type, epilogue slot
// capacity) and the bytes of LDS one wave's private region takes
// The first nine parameters repeat what a wave needs FIRST -- where the work items are and how tiles map to them -- as
// plain scalars ahead of the two argument structs: the build preloads leading scalar kernel arguments into SGPRs
// (-amdgpu-kernarg-preload-count), so a wave's first dependent load (its item record) leaves without waiting for the
// argument segment.  items / total_tiles / tiles / tile_mul / tile_shift: as MelspecArgs::items, n_items x tiles, tiles,
// tile_mul, tile_shift.
// blob / blob_bytes: as WaveArgs::blob, blob_bytes (the table staging's loads are the very first a wave issues);
// n_wgs / xcd_remap: the grid's size and MelspecArgs::xcd_remap (gridDim itself would be one more load in front of the
// item record's).
typedef void (*wave_kernel_t)(const aud_item*, unsigned, unsigned, unsigned, int, const void*, int, unsigned, int, const MelspecArgs,
                              const WaveArgs);
// the workgroup-per-item variant (melspec_w20.hip): items, n_items, tiles per item, blob, blob bytes, the float32 gabor taps (a
// direct restrict parameter: gabor_tile.h), the three argument structs
typedef void (*item_kernel_t)(const aud_item*, unsigned, unsigned, const void*, int, const float*, const MelspecArgs, const WaveArgs,
                              const ItemArgs);
item_kernel_t w20_item_kernel(bool f64, int sig_dtype, int n_slots, int waves);  // waves: 5, else null
__device__ __forceinline__ unsigned tile_div(unsigned mul, int shift, unsigned n) { return shift < 0 ? n : __umulhi(n, mul) >> shift; }
wave_kernel_t w16_kernel(bool f64, int sig_dtype, int n_slots);
wave_kernel_t w20_kernel(bool f64, int sig_dtype, int n_slots);
wave_kernel_t w64_kernel(bool f64, int sig_dtype, int n_slots);
size_t w16_region_bytes(bool f64);
size_t w20_region_bytes(bool f64);
size_t w64_region_bytes(bool f64);
namespace w16 { constexpr int kFW = 4, kN = 512, kM = 256; }
namespace w20 { constexpr int kFW = 6, kLPF = 10, kN = 400, kM = 200; }
namespace w64 { constexpr int kN = 2048, kM = 1024, kFPW = 4; }

namespace {

// The workgroup's table blob: loads first (kept in registers), stores after the caller has issued its operand loads.
// NT threads, 4 x 16 bytes per thread through a buffer descriptor over the blob: pieces past its end come back as zeros
// from the hardware's range check (no index clamps, no memory traffic) and are stored like the others -- the first
// 64 NT bytes of dynamic LDS belong to the blob and the waves' scratch regions, which nobody has written yet
// (melspec_wave_finish keeps the launch's LDS at least that large).  Larger blobs finish with a plain copy loop.
template <int NT>
struct BlobRegs {
    uint4 v[4];
};
template <int NT>
__device__ __forceinline__ void blob_fetch(const void* blob_ptr, int blob_bytes, int tid, BlobRegs<NT>& b) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(blob_ptr), 0, blob_bytes, 0x00020000);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (tid + NT * q) * 16, 0, 0);
        b.v[q] = uint4{r[0], r[1], r[2], r[3]};
    }
}
template <int NT>
__device__ __forceinline__ void blob_store(const WaveArgs& e, unsigned char* smem, int tid, const BlobRegs<NT>& b) {
    uint4* l = reinterpret_cast<uint4*>(smem);
    const uint4* __restrict__ g = static_cast<const uint4*>(e.blob);
    const int n16 = e.blob_bytes >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) l[tid + NT * q] = b.v[q];
#pragma unroll 1
    for (int i = tid + 4 * NT; i < n16; i += NT) l[i] = g[i];
}

// One (Z[k], Z[M - k]) pair of the real-FFT split -> bins k and M - k of the float32 spectrum (FOUR times the power over
// 2^sc; the 1/4 lives in the blob's mel weights): X[k] = (E + T)/2, X[M-k] = conj(E - T)/2, E = A + conj B,
// T = -i W_N^k (A - conj B), evaluated from the pair's k <= M/2 side.  Squaring AFTER the subtraction keeps a weak bin
// next to a strong partner at the FFT's own accuracy (squares first would not).
template <typename TT, bool SCALED = true>
__device__ __forceinline__ void split_pair(float* P, C2<TT> w, int M, int k, C2<TT> A, C2<TT> B, int sc) {
    const C2<TT> E = {A.x + B.x, A.y - B.y};
    const C2<TT> D = {A.x - B.x, A.y + B.y};
    const C2<TT> mD = {D.y, -D.x};
    const C2<TT> Tm = cmul(mD, w);
    const TT xr = E.x + Tm.x, xi = E.y + Tm.y;
    const TT yr = E.x - Tm.x, yi = E.y - Tm.y;
    P[k] = scaled_power<SCALED>(mad(xr, xr, xi * xi), sc);
    P[M - k] = scaled_power<SCALED>(mad(yr, yr, yi * yi), sc);  // k = 0 -> the Nyquist bin M; k = M/2 -> the same bin, same value
}

}  // namespace
}  // namespace aud
