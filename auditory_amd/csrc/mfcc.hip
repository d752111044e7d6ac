// MFCC tail of the segment loop, on stored mel / log-power tensors:
//   k_mfcc_dct    mel.Params.CepstrumDct (mel/mel.go:192-212) for every processed step: unnormalised
//                 DCT-I of the nf log-mel values (what gonum's fourier.DCT.Transform computes),
//                 c0 <- ln(1 + c0^2), first NCoefs kept; steps the loop never reached stay 0
//   k_mfcc_energy sound/sndenv.go:360-372: Energy[s] and the overwrite of MFCC row 0 with it.  The
//                 reference sums LogPowerSegment.FloatValRowCell(s, f) over f < T, i.e. the log-power
//                 of BIN s over the steps of the segment -- reproduced as is
//   k_mfcc_deltas sound/sndenv.go:378-431: deltas / delta-deltas over +-2 steps with clamped edges;
//                 the running sums `prv` / `nxt` are carried from one coefficient to the next and the
//                 n = 2 denominator is the one that sticks -- reproduced as is
#include "kernels.h"

namespace aud {
namespace {

__device__ __forceinline__ float dev_log1p_sq(float c) { return logf(1.0f + c * c); }
__device__ __forceinline__ double dev_log1p_sq(double c) { return log(1.0 + c * c); }

template <typename TT>
__global__ __launch_bounds__(256) void k_mfcc_dct(const MfccArgs a) {
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int per_item = a.n_coefs * a.T;
    if (gid >= int64_t(a.n_items) * per_item) return;
    const int item = int(gid / per_item);
    const int r = int(gid - int64_t(item) * per_item);
    const int coef = r / a.T, s = r - coef * a.T;  // consecutive threads = consecutive steps
    const aud_item it = a.items[item];
    const int64_t start = int64_t(it.start0) + int64_t(a.S) * (s - a.border);
    float res = 0.f;
    if (start + a.N <= int64_t(it.sig_len)) {
        const TT* __restrict__ C = static_cast<const TT*>(a.dct) + size_t(coef) * a.nf;  // [n_coefs][nf]
        const float* col = a.mel + size_t(item) * a.nf * a.T + s;
        TT acc = TT(0);
        for (int j = 0; j < a.nf; ++j) acc += C[j] * TT(col[size_t(j) * a.T]);
        if (coef == 0) acc = dev_log1p_sq(acc);
        res = float(acc);
    }
    a.mfcc[gid] = res;
}

template <typename TT>
__global__ __launch_bounds__(256) void k_mfcc_energy(const MfccArgs a) {
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= int64_t(a.n_items) * a.T) return;
    const int item = int(gid / a.T), s = int(gid - int64_t(item) * a.T);
    const float* row = a.log_power + (size_t(item) * a.H + s) * a.T;  // bin s, all steps
    TT e = TT(0);
    for (int f = 0; f < a.T; ++f) e += TT(row[f]);
    if (a.energy) a.energy[gid] = float(e);
    a.mfcc[size_t(item) * a.n_coefs * a.T + s] = float(e);  // SetFloatRowCell(0, s, Energy[s])
}

template <typename TT>
__global__ __launch_bounds__(256) void k_mfcc_deltas(const MfccArgs a, const float* src, float* dst) {
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= int64_t(a.n_items) * a.T) return;
    const int item = int(gid / a.T), s = int(gid - int64_t(item) * a.T);
    const float* in = src + size_t(item) * a.n_coefs * a.T;
    float* out = dst + size_t(item) * a.n_coefs * a.T;
    TT prv = TT(0), nxt = TT(0);
    for (int i = 0; i < a.n_coefs; ++i) {
        TT nume = TT(0), d = TT(0);
        for (int n = 1; n <= 2; ++n) {
            const int sprv = max(s - n, 0), snxt = min(s + n, a.T - 1);
            prv += TT(in[size_t(i) * a.T + sprv]);
            nxt += TT(in[size_t(i) * a.T + snxt]);
            nume += TT(n) * (nxt - prv);
            d = nume / TT(2 * n * n);
        }
        out[size_t(i) * a.T + s] = float(d);
    }
}

}  // namespace

// mel.Params.CepstrumDct alone (the reference's per-step entry point)
hipError_t launch_mfcc_dct(const MfccArgs& a, int compute_dtype, hipStream_t st) {
    const int64_t n1 = int64_t(a.n_items) * a.n_coefs * a.T;
    if (n1 == 0) return hipSuccess;
    const dim3 g1(unsigned((n1 + 255) / 256)), b(256);
    if (compute_dtype == AUD_F64) hipLaunchKernelGGL(k_mfcc_dct<double>, g1, b, 0, st, a);
    else hipLaunchKernelGGL(k_mfcc_dct<float>, g1, b, 0, st, a);
    return hipGetLastError();
}

hipError_t launch_mfcc(const MfccArgs& a, int compute_dtype, hipStream_t st) {
    const bool f64 = compute_dtype == AUD_F64;
    const int64_t n1 = int64_t(a.n_items) * a.n_coefs * a.T, n2 = int64_t(a.n_items) * a.T;
    if (n1 == 0) return hipSuccess;
    const dim3 g1(unsigned((n1 + 255) / 256)), g2(unsigned((n2 + 255) / 256)), b(256);
    if (f64) hipLaunchKernelGGL(k_mfcc_dct<double>, g1, b, 0, st, a);
    else hipLaunchKernelGGL(k_mfcc_dct<float>, g1, b, 0, st, a);
    if (f64) hipLaunchKernelGGL(k_mfcc_energy<double>, g2, b, 0, st, a);
    else hipLaunchKernelGGL(k_mfcc_energy<float>, g2, b, 0, st, a);
    if (a.deltas) {
        if (f64) hipLaunchKernelGGL(k_mfcc_deltas<double>, g2, b, 0, st, a, (const float*)a.mfcc, a.deltas);
        else hipLaunchKernelGGL(k_mfcc_deltas<float>, g2, b, 0, st, a, (const float*)a.mfcc, a.deltas);
        if (a.delta_deltas) {
            if (f64) hipLaunchKernelGGL(k_mfcc_deltas<double>, g2, b, 0, st, a, (const float*)a.deltas, a.delta_deltas);
            else hipLaunchKernelGGL(k_mfcc_deltas<float>, g2, b, 0, st, a, (const float*)a.deltas, a.delta_deltas);
        }
    }
    return hipGetLastError();
}

}  // namespace aud
