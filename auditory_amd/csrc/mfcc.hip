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

// The whole tail for one item in one workgroup (what launch_mfcc runs when the item's tensors fit LDS): the item's mel
// matrix and the DCT rows are staged in LDS once (coalesced), the DCT reads them from there (the three-kernel form re-read
// every mel value 13 times through L1), the Energy rows are summed by a wave each with lanes along the steps (coalesced;
// the one-thread-per-row form walked 104 cache lines per load instruction), and the deltas / delta-deltas read the
// float32-ROUNDED coefficients back from LDS.  That rounding is THIS library's, not the reference's: MelFBankSegment,
// LogPowerSegment, MFCCSegment and MFCCDeltas are etensor.Float64 in the reference (sound/sndenv.go:106-136), the boundary
// here carries float32 tensors (north star: "1e-5 relative on the float32 mel / gabor tensors"), so the tail's inputs
// differ from the reference's by their float32 rounding (deviations) -- what the tail's tolerances cover.
// Same arithmetic and the same summation order as the kernels above, except that an Energy row is a tree sum of its
// float64-widened values.
constexpr int kFusedThreads = 512;

template <typename TT>
__global__ __launch_bounds__(kFusedThreads) void k_mfcc_fused(const MfccArgs a) {
    unsigned char* mfcc_lds = dyn_lds();
    const int T = a.T, nf = a.nf, nc = a.n_coefs;
    float* melL = reinterpret_cast<float*>(mfcc_lds);                 // [nf][T]
    float* cofL = melL + nf * T;                                      // [nc][T]  mfcc, float32-rounded
    float* dltL = cofL + nc * T;                                      // [nc][T]  deltas, float32-rounded
    TT* dctL = reinterpret_cast<TT*>(dltL + nc * T + ((nf * T + 2 * nc * T) & 1));  // [nc][nf], 8-byte aligned
    const int tid = int(threadIdx.x), item = int(blockIdx.x);
    const aud_item it = a.items[item];
    const float* __restrict__ mel = a.mel + size_t(item) * nf * T;
    for (int i = tid; i < nf * T; i += kFusedThreads) melL[i] = mel[i];
    const TT* __restrict__ C = static_cast<const TT*>(a.dct);
    for (int i = tid; i < nc * nf; i += kFusedThreads) dctL[i] = C[i];
    __syncthreads();
    // CepstrumDct per processed step (mel/mel.go:192-212)
    for (int idx = tid; idx < nc * T; idx += kFusedThreads) {
        const int coef = idx / T, s = idx - coef * T;
        const int64_t start = int64_t(it.start0) + int64_t(a.S) * (s - a.border);
        float res = 0.f;
        if (start + a.N <= int64_t(it.sig_len)) {
            TT acc = TT(0);
#pragma unroll 8
            for (int j = 0; j < nf; ++j) acc += dctL[coef * nf + j] * TT(melL[j * T + s]);
            if (coef == 0) acc = dev_log1p_sq(acc);
            res = float(acc);
        }
        cofL[idx] = res;
    }
    __syncthreads();
    // Energy[s] = sum over f < T of LogPowerSegment(s, f) (sound/sndenv.go:360-372) -> MFCC row 0
    {
        const int wave = tid >> 6, lane = tid & 63;
        const float* __restrict__ lp = a.log_power + size_t(item) * a.H * T;
        constexpr int R = 13, NW = kFusedThreads / 64;  // rows in flight per wave: their loads are issued together, then reduced one by one
        for (int s0 = wave; s0 < T; s0 += NW * R) {
            TT e[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int s = s0 + NW * r;
                TT acc = TT(0);
                if (s < T)
                    for (int f = lane; f < T; f += 64) acc += TT(lp[size_t(s) * T + f]);
                e[r] = acc;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                TT v = e[r];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                const int s = s0 + NW * r;
                if (lane == 0 && s < T) {
                    if (a.energy) a.energy[size_t(item) * T + s] = float(v);
                    cofL[s] = float(v);  // SetFloatRowCell(0, s, Energy[s])
                }
            }
        }
    }
    __syncthreads();
    float* mf = a.mfcc + size_t(item) * nc * T;
    for (int i = tid; i < nc * T; i += kFusedThreads) mf[i] = cofL[i];
    if (!a.deltas) return;  // uniform
    // deltas, then delta-deltas (sound/sndenv.go:378-431): one thread per step, coefficients in order with carried sums
    for (int pass = 0; pass < 2; ++pass) {
        const float* in = pass ? dltL : cofL;
        float* outL = pass ? nullptr : dltL;
        float* out = (pass ? a.delta_deltas : a.deltas) + size_t(item) * nc * T;
        for (int s = tid; s < T; s += kFusedThreads) {
            TT prv = TT(0), nxt = TT(0);
            for (int i = 0; i < nc; ++i) {
                TT nume = TT(0), d = TT(0);
                for (int n = 1; n <= 2; ++n) {
                    const int sprv = max(s - n, 0), snxt = min(s + n, T - 1);
                    prv += TT(in[i * T + sprv]);
                    nxt += TT(in[i * T + snxt]);
                    nume += TT(n) * (nxt - prv);
                    d = nume / TT(2 * n * n);
                }
                const float df = float(d);
                out[i * T + s] = df;
                if (outL) outL[i * T + s] = df;
            }
        }
        if (!a.delta_deltas) return;  // uniform
        __syncthreads();
    }
}

// What is left of the tail behind a mel launch that carried MelspecArgs::mfcc_acc / energy_part (aud_segment_batch_dev): one
// workgroup per item.  The coefficients arrive UNROUNDED in the compute type and stay so in LDS: deltas and delta-deltas
// are computed from them as the reference computes them from its float64 tensors (sndenv.go:378-431); float32 only at the
// stores.  Energy[s] = the per-tile sums added in tile order (sndenv.go:360-366), MFCC row 0 <- Energy (:368-372).
// (threads per item: 64 / 128 / 256 measured 29.3 / 28.9 / 28.8 us per whole-ProcessSegment step of 256 items on four streams --
// the step is bound by the mel kernel's spectrum outputs, not by this launch: profiles/round4_segment_finish_threads.txt)
constexpr int kFinishThreads = 256;

template <typename TT>
__global__ __launch_bounds__(kFinishThreads) void k_segment_finish(const SegmentFinishArgs a) {
    TT* cofL = reinterpret_cast<TT*>(dyn_lds());  // [nc][T]
    TT* dltL = cofL + a.n_coefs * a.T;            // [nc][T]
    const int T = a.T, nc = a.n_coefs, tid = int(threadIdx.x), item = int(blockIdx.x);
    const TT* __restrict__ acc = static_cast<const TT*>(a.mfcc_acc) + size_t(item) * nc * T;
    for (int i0 = tid + T; i0 < nc * T; i0 += 8 * kFinishThreads) {  // rows 1..: the mel kernel's DCT (eight loads in flight)
        TT part[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) part[u] = i0 + u * kFinishThreads < nc * T ? acc[i0 + u * kFinishThreads] : TT(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * kFinishThreads < nc * T) cofL[i0 + u * kFinishThreads] = part[u];
    }
    const TT* __restrict__ ep = static_cast<const TT*>(a.energy_part) + size_t(item) * a.tiles * T;
    for (int s = tid; s < T; s += kFinishThreads) {
        TT e = TT(0);
        for (int t0 = 0; t0 < a.tiles; t0 += 8) {  // eight loads in flight, added in tile order
            TT part[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) part[u] = t0 + u < a.tiles ? ep[size_t(t0 + u) * T + s] : TT(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) e += part[u];
        }
        if (a.energy) a.energy[size_t(item) * T + s] = float(e);
        cofL[s] = e;  // SetFloatRowCell(0, s, Energy[s])
    }
    __syncthreads();
    float* mf = a.mfcc + size_t(item) * nc * T;
    for (int i = tid; i < nc * T; i += kFinishThreads) mf[i] = float(cofL[i]);
    if (!a.deltas) return;  // uniform
    for (int pass = 0; pass < 2; ++pass) {
        const TT* in = pass ? dltL : cofL;
        float* out = (pass ? a.delta_deltas : a.deltas) + size_t(item) * nc * T;
        for (int s = tid; s < T; s += kFinishThreads) {
            TT prv = TT(0), nxt = TT(0);
            const int p1 = max(s - 1, 0), p2 = max(s - 2, 0), n1 = min(s + 1, T - 1), n2 = min(s + 2, T - 1);
#pragma unroll 4
            for (int i = 0; i < nc; ++i) {
                const TT a1 = in[i * T + p1], b1 = in[i * T + n1], a2 = in[i * T + p2], b2 = in[i * T + n2];
                TT nume = TT(0);
                prv += a1;  // n = 1 (its quotient nume / 2 is overwritten by the n = 2 one, sndenv.go:399-402)
                nxt += b1;
                nume += nxt - prv;
                prv += a2;  // n = 2
                nxt += b2;
                nume += TT(2) * (nxt - prv);
                const TT d = nume / TT(8);
                out[i * T + s] = float(d);
                if (!pass) dltL[i * T + s] = d;
            }
        }
        if (!a.delta_deltas) return;  // uniform
        __syncthreads();
    }
}

}  // namespace

size_t segment_finish_lds_bytes(int n_coefs, int T, int compute_dtype) {
    return 2 * size_t(n_coefs) * T * (compute_dtype == AUD_F64 ? 8 : 4);
}

hipError_t launch_segment_finish(const SegmentFinishArgs& a, int compute_dtype, hipStream_t st) {
    if (a.n_items == 0) return hipSuccess;
    const size_t lds = segment_finish_lds_bytes(a.n_coefs, a.T, compute_dtype);
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(k_segment_finish<double>, dim3(unsigned(a.n_items)), dim3(kFinishThreads), lds, st, a);
    else hipLaunchKernelGGL(k_segment_finish<float>, dim3(unsigned(a.n_items)), dim3(kFinishThreads), lds, st, a);
    return hipGetLastError();
}

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
    const size_t lds = (size_t(a.nf) * a.T + 2 * size_t(a.n_coefs) * a.T + 1) * sizeof(float) +
                       size_t(a.n_coefs) * a.nf * (f64 ? 8 : 4) + 8;
    if (lds <= 64 * 1024) {  // the item's tensors fit LDS: one launch, one workgroup per item
        if (f64) hipLaunchKernelGGL(k_mfcc_fused<double>, dim3(unsigned(a.n_items)), dim3(kFusedThreads), lds, st, a);
        else hipLaunchKernelGGL(k_mfcc_fused<float>, dim3(unsigned(a.n_items)), dim3(kFusedThreads), lds, st, a);
        return hipGetLastError();
    }
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
