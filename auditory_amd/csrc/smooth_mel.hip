// dft.Params.PrevSmooth != 0: the reference blends each step's power with the previous
// step's (dft/dft.go:67-69), which makes the frames of a segment sequentially dependent:
//     p_0 = raw_0;   p_s = PrevSmooth * p_{s-1} + CurSmooth * raw_s   (s > 0)
// and everything downstream (log-power, mel) is computed from the smoothed p_s.  The frame->FFT
// kernels are frame-parallel, so this mode runs as three launches: raw power [item, H, T] from
// the FFT kernel, the scan below along T (one thread per (item, bin)), then mel from the smoothed
// power.  The default PrevSmooth = 0 never comes here.
#include "kernels.h"

namespace aud {
namespace {

__device__ __forceinline__ float dev_log(float v) { return logf(v); }
__device__ __forceinline__ double dev_log(double v) { return log(v); }

// number of leading live frames of an item: frame s is live iff start0 + S (s - border) + N <= sig_len
__device__ __forceinline__ int live_frames(const aud_item& it, int N, int S, int T, int border) {
    int n = 0;
    for (int s = 0; s < T; ++s) {
        const int64_t start = int64_t(it.start0) + int64_t(S) * (s - border);
        if (start + N <= int64_t(it.sig_len)) ++n;
        else break;
    }
    return n;
}

template <typename TT>
__global__ __launch_bounds__(256) void k_power_smooth(const SmoothArgs a) {
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= int64_t(a.n_items) * a.H) return;
    const int item = int(gid / a.H);
    const int nlive = live_frames(a.items[item], a.N, a.S, a.T, a.border);
    float* prow = a.power + size_t(gid) * a.T;
    float* lrow = a.log_power ? a.log_power + size_t(gid) * a.T : nullptr;
    const TT prev = TT(a.prev_smooth), cur = TT(a.cur_smooth);
    const TT off = TT(a.log_off), lmin = TT(a.log_min);
    TT carry = TT(0);
    for (int s = 0; s < nlive; ++s) {
        TT p = TT(prow[s]);
        if (s > 0) p = prev * carry + cur * p;  // dft.go:67-69
        carry = p;
        prow[s] = float(p);
        if (lrow) {
            const TT v = p + off;
            lrow[s] = a.comp_log_pow ? float(v == TT(0) ? lmin : dev_log(v)) : 0.f;
        }
    }
}

// dft.Params.Power for ONE step with the caller's carry (dft/dft.go:62-85): raw[k] is this step's
// re^2+im^2, carry[k] the caller's `power` tensor from the previous step
template <typename TT>
__global__ __launch_bounds__(256) void k_frame_blend(const float* raw, int raw_stride, const double* carry, int H,
                                                      int step, double prev, double cur, int comp_log_pow,
                                                      double log_off, double log_min, double* out_p, double* out_lp) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= H) return;
    TT p = TT(raw[size_t(k) * raw_stride]);
    if (step > 0) p = TT(prev) * TT(carry[k]) + TT(cur) * p;  // dft.go:67-69
    out_p[k] = double(p);
    if (comp_log_pow) {
        const TT v = p + TT(log_off);
        out_lp[k] = double(v == TT(0) ? TT(log_min) : dev_log(v));
    } else {
        out_lp[k] = 0.0;
    }
}

// dft.Params.Power's first line for ONE step (dft/dft.go:63-66): raw[k] = re^2 + im^2 of the caller's complex128
// coefficients, in the compute type, stored like the FFT kernels store their raw power (float32)
template <typename TT>
__global__ __launch_bounds__(256) void k_power_from_coefs(const double* coefs, int H, float* raw) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= H) return;
    const TT rl = TT(coefs[2 * k]), im = TT(coefs[2 * k + 1]);
    raw[k] = float(rl * rl + im * im);
}

// mel.Params.FilterDft (mel/mel.go:120-153) applied to a stored power tensor [item, H, T]
template <typename TT>
__global__ __launch_bounds__(256) void k_mel_from_power(const MelspecArgs a) {
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int per_item = a.nf * a.T;
    if (gid >= int64_t(a.n_items) * per_item) return;
    const int item = int(gid / per_item);
    const int r = int(gid - int64_t(item) * per_item);
    const int flt = r / a.T, s = r - flt * a.T;  // consecutive threads = consecutive steps: coalesced
    const aud_item it = a.items[item];
    const int64_t start = int64_t(it.start0) + int64_t(a.S) * (s - a.border);
    const bool live = start + a.N <= int64_t(it.sig_len);
    float res = 0.f;
    if (live) {
        const TT* __restrict__ filt = static_cast<const TT*>(a.filt);
        const int lo = a.bin_pts[flt], hi = a.bin_pts[flt + 2];
        const TT* wrow = filt + size_t(flt) * (a.nf + 2);
        const float* pcol = a.power + size_t(item) * a.H * a.T + s;
        TT sum = TT(0);
        for (int bin = lo; bin <= hi; ++bin) sum += wrow[bin - lo] * TT(pcol[size_t(bin) * a.T]);
        sum += TT(a.mel_log_off);
        TT val = (sum == TT(0)) ? TT(a.mel_log_min) : dev_log(sum);
        if (a.renorm) {
            val -= TT(a.renorm_min);
            if (val < TT(0)) val = TT(0);
            val *= TT(a.renorm_scale);
            if (val > TT(1)) val = TT(1);
        }
        res = float(val);
    }
    a.mel[gid] = res;
}

}  // namespace

hipError_t launch_power_smooth(const SmoothArgs& a, int compute_dtype, hipStream_t st) {
    const int64_t total = int64_t(a.n_items) * a.H;
    if (total == 0) return hipSuccess;
    const dim3 grid(unsigned((total + 255) / 256));
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(k_power_smooth<double>, grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(k_power_smooth<float>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_frame_blend(const float* raw, int raw_stride, const double* carry, int H, int step,
                              double prev, double cur, int comp_log_pow, double log_off, double log_min,
                              double* out_p, double* out_lp, int compute_dtype, hipStream_t st) {
    const dim3 grid(unsigned((H + 255) / 256));
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(k_frame_blend<double>, grid, dim3(256), 0, st, raw, raw_stride, carry, H, step, prev, cur,
                           comp_log_pow, log_off, log_min, out_p, out_lp);
    else
        hipLaunchKernelGGL(k_frame_blend<float>, grid, dim3(256), 0, st, raw, raw_stride, carry, H, step, prev, cur,
                           comp_log_pow, log_off, log_min, out_p, out_lp);
    return hipGetLastError();
}

hipError_t launch_power_from_coefs(const double* coefs, int H, float* raw, int compute_dtype, hipStream_t st) {
    const dim3 grid(unsigned((H + 255) / 256));
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(k_power_from_coefs<double>, grid, dim3(256), 0, st, coefs, H, raw);
    else
        hipLaunchKernelGGL(k_power_from_coefs<float>, grid, dim3(256), 0, st, coefs, H, raw);
    return hipGetLastError();
}

hipError_t launch_mel_from_power(const MelspecArgs& a, int compute_dtype, hipStream_t st) {
    const int64_t total = int64_t(a.n_items) * a.nf * a.T;
    if (total == 0) return hipSuccess;
    const dim3 grid(unsigned((total + 255) / 256));
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(k_mel_from_power<double>, grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(k_mel_from_power<float>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

namespace {
// four values per thread and trip: one 16-byte load, two 16-byte stores -- whole 64-byte pieces per four lanes on the link
__global__ __launch_bounds__(256) void k_widen_to_host(const float* __restrict__ src, double* __restrict__ dst, size_t n) {
    const size_t stride = size_t(gridDim.x) * 256 * 4;
    for (size_t i = (size_t(blockIdx.x) * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n && ((reinterpret_cast<uintptr_t>(src + i) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dst + i) & 15) == 0)) {
            const float4 v = *reinterpret_cast<const float4*>(src + i);
            *reinterpret_cast<double2*>(dst + i) = double2{double(v.x), double(v.y)};
            *reinterpret_cast<double2*>(dst + i + 2) = double2{double(v.z), double(v.w)};
        } else {
            for (size_t u = i; u < n && u < i + 4; ++u) dst[u] = double(src[u]);
        }
    }
}
}  // namespace

hipError_t launch_widen_to_host(const float* src, double* dst_host, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const size_t wgs = std::min<size_t>(1024, (n + 1023) / 1024);
    hipLaunchKernelGGL(k_widen_to_host, dim3(unsigned(wgs)), dim3(256), 0, st, src, dst_host, n);
    return hipGetLastError();
}

}  // namespace aud
