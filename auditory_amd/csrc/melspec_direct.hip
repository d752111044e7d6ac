// The window lengths NOTHING ELSE RUNS: the DFT as the O(N H) sum it is defined by.
//
// dft.Filter (dft/dft.go:42-50) takes any window length -- its FFT is gonum's, which has no length limit -- so a drop-in must
// not refuse one.  The any-N kernel (melspec_generic.hip) keeps a frame's transform in LDS: smooth lengths up to N = 10 240
// (even) / 5 120 (odd), lengths with a prime factor above 25 up to where Bluestein's L >= 2 M - 1 fits (N = 5 120 even / 2 560
// odd).  Behind that -- 200 ms windows at 44.1 kHz (N = 8 820 = 2 x 4410: L >= 8 819), half-second windows anywhere -- this
// kernel: one workgroup per frame, a thread per bin k, X[k] = sum_n x[n] W_N^(k n) with the exponent k n mod N advanced by
// addition, float64 sums over a float64 table whatever the plan computes in (a float32 sum of 10^4 terms would sit at the
// float32 criterion).  Only the power spectrum (H values) lives in LDS, so N <= ~38 000 in float64; beyond that plans are
// refused (AUD_EINVAL) as before.  It is a completeness path, not a fast one: ~20 cycles per (bin, sample) and thread.
//
// Everything behind the power spectrum is frames_epilogue.h, shared with the any-N and chirp kernels.  Reference semantics:
// sound/sndenv.go:438-478 (window extraction, left zero pad, short-signal masking), dft/dft.go:42-85, mel/mel.go:120-153.
#include "device_common.h"
#include "frames_epilogue.h"

namespace aud {
namespace {

// the bins of one frame: P[k] = |sum_n x[n] exp(-2 pi i k n / N)|^2, n over the samples that are not left pad
template <typename TT, typename S>
__device__ __forceinline__ void direct_bins(const S* __restrict__ stream, int64_t stride, int64_t start, int N, int H,
                                            const C2<double>* __restrict__ tw, TT* P, int tid) {
    const int n0 = start < 0 ? int(-start < int64_t(N) ? -start : int64_t(N)) : 0;  // (uniform) window samples inside the pad
    for (int k = tid; k < H; k += 256) {
        double re = 0.0, im = 0.0;
        int e = int((int64_t(k) * n0) % N);
        for (int n = n0; n < N; ++n) {
            const S r = stream[(start + n) * stride];  // (one address for the whole wave)
            double x;
            if constexpr (sizeof(S) == 2) x = pcm16_to_double(int(r));  // int16 PCM / 0x7FFF, sound.go:138
            else x = double(r);
            const C2<double> w = tw[e];
            re = fma(x, w.x, re);
            im = fma(x, w.y, im);
            e += k;
            e -= e >= N ? N : 0;
        }
        P[k] = TT(re * re + im * im);  // dft.go:64-66
    }
}

template <typename TT>
__global__ __launch_bounds__(256) void k_melspec_direct(const MelspecArgs a) {
    TT* P = reinterpret_cast<TT*>(dyn_lds());  // [H | 1] power spectrum, then the fused tail's nf log-mel values
    const int tid = threadIdx.x;
    const int N = a.N, H = a.H, T = a.T;
    const int wg = int(tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap));
    const int item = wg / T, t0 = wg - item * T;  // one frame per workgroup (a.F = 1)
    const aud_item it = a.items[item];
    const int64_t stride = it.sig_stride > 1 ? it.sig_stride : 1;
    const int64_t start = int64_t(it.start0) + int64_t(a.S) * (t0 - a.border);
    const bool live = start + N <= int64_t(it.sig_len);  // (uniform; a masked frame's outputs are written as zeros by the epilogue)
    const C2<double>* __restrict__ tw = static_cast<const C2<double>*>(a.tw64);
    if (live) {
        if (a.sig_dtype == AUD_F32) direct_bins<TT>(static_cast<const float*>(a.sig) + it.sig_off, stride, start, N, H, tw, P, tid);
        else if (a.sig_dtype == AUD_F64) direct_bins<TT>(static_cast<const double*>(a.sig) + it.sig_off, stride, start, N, H, tw, P, tid);
        else direct_bins<TT>(static_cast<const int16_t*>(a.sig) + it.sig_off, stride, start, N, H, tw, P, tid);
    } else {
        for (int k = tid; k < H; k += 256) P[k] = TT(0);
    }
    __syncthreads();
    frames_epilogue<TT>(a, it, item, T, t0, P, tid);
}

}  // namespace

size_t melspec_direct_lds_bytes(int H, int nf, int compute_dtype) {
    const size_t tsz = compute_dtype == AUD_F64 ? 8 : 4;
    return (size_t(H | 1) + size_t(nf)) * tsz + 16;
}

// the attribute belongs to the kernel, not to a plan: only ever raised, to the device's limit
hipError_t melspec_direct_prepare() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_melspec_direct<double>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_melspec_direct<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_melspec_direct(const MelspecArgs& a, int compute_dtype, hipStream_t st) {
    if (a.F != 1 || !a.tw64) return hipErrorInvalidValue;
    const dim3 grid(unsigned(a.n_items) * unsigned(a.T));
    const size_t lds = melspec_direct_lds_bytes(a.H, a.nf, compute_dtype);
    if (compute_dtype == AUD_F64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_direct<double>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_direct<float>), grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace aud
