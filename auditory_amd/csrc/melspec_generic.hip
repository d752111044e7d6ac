// Any-N frame -> FFT -> power -> mel -> log kernel.
//
// One workgroup (256 threads) owns F consecutive frames of one work item.  The frames are
// gathered into LDS, transformed with an autosort Stockham FFT whose stages ping-pong
// between two LDS buffers, turned into a power spectrum, and reduced through the mel
// triangle table.  Even N uses the packed-real trick (an N/2-point complex FFT plus one
// split pass); odd N (e.g. the prime 1103 that 25 ms @ 44.1 kHz produces) runs a full
// N-point complex FFT.  Radix 2, 3, 4, 5, 8, 16 and 25 stages run whole butterflies in registers
// (the host factorises M into as few of them as possible).  A length with any other prime factor
// p takes Bluestein's chirp convolution (two power-of-two FFTs of length L >= 2 M - 1, one frame per
// workgroup) where its two buffers fit LDS, and an O(p) per-output pass otherwise, so every N is
// supported.  This is the universal path; the common sizes have faster specialised kernels.
//
// Reference semantics implemented here: sound/sndenv.go:438-478 (window extraction,
// left zero pad, short-signal masking), dft/dft.go:53-85 (DFT of the raw window, power,
// log(power+offset)), mel/mel.go:120-153 (triangle sums, log, renorm).
#include "device_common.h"

namespace aud {
namespace {

// one autosort Stockham stage of radix P: every thread takes whole butterflies
template <typename TT, int P>
__device__ __forceinline__ void stage(const C2<TT>* src, C2<TT>* dst, const C2<TT>* __restrict__ tw, int F,
                                      int M, int N, int ratio, int ncur, int s, int tid) {
    const int m = ncur / P, nb = M / P, sm = s * m;
    for (int w = tid; w < F * nb; w += blockDim.x) {
        const int f = w / nb, b = w - f * nb;
        const int q = b / s, k = b - q * s;
        const C2<TT>* x = src + size_t(f) * M + k + s * q;
        C2<TT>* y = dst + size_t(f) * M + k + s * P * q;
        C2<TT> v[P];
#pragma unroll
        for (int i = 0; i < P; ++i) v[i] = x[i * sm];
        SmallDft<TT, P>::run(v, tw, N);
        const int tq = q * s * ratio;
        y[0] = v[0];
#pragma unroll
        for (int j = 1; j < P; ++j) y[j * s] = cmul<TT>(v[j], tw[j * tq]);
    }
}

// the power-of-two FFT of the Bluestein route: radix 16 / 8 / 4 / 2 stages of one frame of length L, result in `src`
template <typename TT>
__device__ __forceinline__ void pow2_fft(C2<TT>*& src, C2<TT>*& dst, const MelspecArgs& a, int tid) {
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.bl_tw);
    const int L = a.bl_L;
    int ncur = L, s = 1;
    for (int stg = 0; stg < a.bl_nfac; ++stg) {
        const int p = a.bl_fac[stg];
        switch (p) {
            case 16: stage<TT, 16>(src, dst, tw, 1, L, L, 1, ncur, s, tid); break;
            case 8: stage<TT, 8>(src, dst, tw, 1, L, L, 1, ncur, s, tid); break;
            case 4: stage<TT, 4>(src, dst, tw, 1, L, L, 1, ncur, s, tid); break;
            default: stage<TT, 2>(src, dst, tw, 1, L, L, 1, ncur, s, tid); break;
        }
        __syncthreads();
        C2<TT>* tmp = src;
        src = dst;
        dst = tmp;
        ncur /= p;
        s *= p;
    }
}

template <typename TT>
__global__ __launch_bounds__(256) void k_melspec_generic(const MelspecArgs a) {
    unsigned char* smem = dyn_lds();
    const int tid = threadIdx.x;
    const int F = a.F, M = a.M, N = a.N, H = a.H, T = a.T;
    C2<TT>* src = reinterpret_cast<C2<TT>*>(smem);
    C2<TT>* dst = src + (a.bl_L ? size_t(a.bl_L) : size_t(F) * M);
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.tw);

    const int tiles = (T + F - 1) / F;
    const int wg = int(tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap));
    const int item = wg / tiles;
    const int t0 = (wg - item * tiles) * F;
    const aud_item it = a.items[item];
    const bool even = (a.ratio == 2);

    // ---- gather the F windows (sndenv.go:455-478) -------------------------------------
    for (int i = tid; i < F * N; i += blockDim.x) {
        const int f = i / N, n = i - f * N;
        const int s = t0 + f;
        const int64_t start = int64_t(it.start0) + int64_t(a.S) * (s - a.border);
        const bool live = s < T && start + N <= int64_t(it.sig_len);
        const int64_t pos = start + n;
        TT v = TT(0);
        if (live && pos >= 0) v = load_sample<TT>(a.sig, a.sig_dtype, it.sig_off + pos * (it.sig_stride > 1 ? it.sig_stride : 1));
        if (even) {
            reinterpret_cast<TT*>(src)[size_t(f) * N + n] = v;  // z[n/2] = (x[2j], x[2j+1])
        } else {
            src[size_t(f) * M + n] = {v, TT(0)};
        }
    }
    __syncthreads();

    // ---- Bluestein route (M has a prime factor the register radices do not cover; F = 1): the length-M DFT as a
    // circular convolution with a chirp, done with two power-of-two FFTs of length L >= 2 M - 1:
    //   n k = (n^2 + k^2 - (k - n)^2) / 2  =>  Z[k] = w[k] sum_n (z[n] w[n]) conj(w)[k - n],  w[n] = exp(-i pi n^2 / M)
    // The inverse FFT is a forward FFT of the conjugate; 1 / L is in bhat.
    if (a.bl_L) {
        const int L = a.bl_L;
        const C2<TT>* __restrict__ chirp = static_cast<const C2<TT>*>(a.bl_chirp);
        const C2<TT>* __restrict__ bhat = static_cast<const C2<TT>*>(a.bl_bhat);
        for (int i = tid; i < L; i += blockDim.x) src[i] = i < M ? cmul<TT>(src[i], chirp[i]) : C2<TT>{TT(0), TT(0)};
        __syncthreads();
        pow2_fft<TT>(src, dst, a, tid);
        for (int i = tid; i < L; i += blockDim.x) {
            const C2<TT> c = cmul<TT>(src[i], bhat[i]);
            src[i] = C2<TT>{c.x, -c.y};
        }
        __syncthreads();
        pow2_fft<TT>(src, dst, a, tid);
        for (int k = tid; k < M; k += blockDim.x) src[k] = cmul<TT>(chirp[k], C2<TT>{src[k].x, -src[k].y});
        __syncthreads();
    }

    // ---- Stockham stages: x[k + s(q + m i)] -> y[k + s(p q + j)] * W_ncur^(q j) ---------
    int ncur = M, s = 1;
    for (int stg = 0; stg < (a.bl_L ? 0 : a.nfac); ++stg) {
        const int p = a.fac[stg];
        const int m = ncur / p;
        const int nb = M / p;  // butterflies per frame
        switch (p) {
            case 2: stage<TT, 2>(src, dst, tw, F, M, N, a.ratio, ncur, s, tid); break;
            case 3: stage<TT, 3>(src, dst, tw, F, M, N, a.ratio, ncur, s, tid); break;
            case 4: stage<TT, 4>(src, dst, tw, F, M, N, a.ratio, ncur, s, tid); break;
            case 5: stage<TT, 5>(src, dst, tw, F, M, N, a.ratio, ncur, s, tid); break;
            case 8: stage<TT, 8>(src, dst, tw, F, M, N, a.ratio, ncur, s, tid); break;
            case 16: stage<TT, 16>(src, dst, tw, F, M, N, a.ratio, ncur, s, tid); break;
            case 25: stage<TT, 25>(src, dst, tw, F, M, N, a.ratio, ncur, s, tid); break;
            default: {
                // any other prime p: one thread per output j of each radix-p butterfly
                const bool prune = (!even) && (a.nfac == 1);  // single prime stage: only k < H is used
                const int jn = prune ? H : p;
                const int per = jn * nb;
                const int wp = N / p;
                for (int w = tid; w < F * per; w += blockDim.x) {
                    const int f = w / per, r = w - f * per;
                    const int j = r / nb, b = r - j * nb;
                    const int q = b / s, k = b - q * s;
                    const C2<TT>* x = src + size_t(f) * M + k + s * q;
                    const int sm = s * m;
                    // p-term direct sum: accumulate in float64 even in the f32 build, so a long
                    // prime pass (p = 1103) does not pile up sqrt(p) f32 roundings
                    double ar = 0.0, ai = 0.0;
                    int e = 0;
                    for (int i = 0; i < p; ++i) {
                        const C2<TT> v = x[i * sm];
                        const C2<TT> c = tw[e * wp];
                        ar += double(v.x) * double(c.x) - double(v.y) * double(c.y);
                        ai += double(v.x) * double(c.y) + double(v.y) * double(c.x);
                        e += j;
                        if (e >= p) e -= p;
                    }
                    const C2<TT> t = tw[int((int64_t(q) * j * s * a.ratio) % N)];
                    dst[size_t(f) * M + k + s * (p * q + j)] = cmul<TT>({TT(ar), TT(ai)}, t);
                }
            }
        }
        __syncthreads();
        C2<TT>* tmp = src;
        src = dst;
        dst = tmp;
        ncur = m;
        s *= p;
    }

    // ---- power spectrum into the free buffer: P[f][k], row pitch odd -----------------
    TT* P = reinterpret_cast<TT*>(dst);
    const int Hp = H | 1;
    for (int w = tid; w < F * H; w += blockDim.x) {
        const int f = w / H, k = w - f * H;
        const C2<TT>* Z = src + size_t(f) * M;
        TT re, im;
        if (even) {
            // X[k] = (Z[k] + conj Z[M-k])/2 - i W_N^k (Z[k] - conj Z[M-k])/2, Z[M] == Z[0]
            const C2<TT> A = Z[k == M ? 0 : k];
            const C2<TT> Bc = Z[k == 0 ? 0 : M - k];
            const TT er = (A.x + Bc.x) * TT(0.5), ei = (A.y - Bc.y) * TT(0.5);
            const TT dr = (A.x - Bc.x) * TT(0.5), di = (A.y + Bc.y) * TT(0.5);
            const C2<TT> wk = tw[k];
            // -i * (dr + i di) = di - i dr
            re = er + (di * wk.x + dr * wk.y);
            im = ei + (di * wk.y - dr * wk.x);
        } else {
            re = Z[k].x;
            im = Z[k].y;
        }
        P[size_t(f) * Hp + k] = re * re + im * im;  // dft.go:64-66
    }
    __syncthreads();

    // ---- optional PowerSegment / LogPowerSegment (dft.go:70-83) -----------------------
    if (a.power || a.log_power) {
        const TT off = TT(a.dft_log_off), lmin = TT(a.dft_log_min);
        for (int w = tid; w < F * H; w += blockDim.x) {
            const int k = w / F, f = w - k * F;
            const int sstep = t0 + f;
            if (sstep >= T) continue;
            const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
            const bool live = start + N <= int64_t(it.sig_len);
            const TT pw = P[size_t(f) * Hp + k];
            const size_t o = (size_t(item) * H + k) * T + sstep;
            if (a.power) a.power[o] = live ? float(pw) : 0.f;
            if (a.log_power) {
                float lp = 0.f;
                if (live && a.comp_log_pow) {
                    const TT v = pw + off;
                    lp = float(v == TT(0) ? lmin : dev_log(v));
                }
                a.log_power[o] = lp;
            }
        }
    }

    // ---- mel triangles + log (mel.go:120-153) ---------------------------------------
    {
        const TT* __restrict__ filt = static_cast<const TT*>(a.filt);
        const int cols = a.nf + 2;
        const TT loff = TT(a.mel_log_off), lmin = TT(a.mel_log_min);
        for (int w = tid; w < F * a.nf; w += blockDim.x) {
            const int flt = w / F, f = w - flt * F;
            const int sstep = t0 + f;
            if (sstep >= T) continue;
            const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
            const bool live = start + N <= int64_t(it.sig_len);
            float res = 0.f;
            if (live) {
                const int lo = a.bin_pts[flt], hi = a.bin_pts[flt + 2];
                const TT* wrow = filt + size_t(flt) * cols;
                const TT* prow = P + size_t(f) * Hp;
                TT sum = TT(0);
                for (int bin = lo; bin <= hi; ++bin) sum += wrow[bin - lo] * prow[bin];
                sum += loff;
                TT val = (sum == TT(0)) ? lmin : dev_log(sum);
                if (a.renorm) {
                    val -= TT(a.renorm_min);
                    if (val < TT(0)) val = TT(0);
                    val *= TT(a.renorm_scale);
                    if (val > TT(1)) val = TT(1);
                }
                res = float(val);
            }
            a.mel[(size_t(item) * a.nf + flt) * T + sstep] = res;
        }
    }
}

}  // namespace

size_t melspec_generic_lds_bytes(int M, int F, int compute_dtype) {
    const size_t c = compute_dtype == AUD_F64 ? 16 : 8;
    return size_t(2) * F * M * c;
}

// Bluestein: L = the power of two >= 2 M - 1 if two complex buffers of that length fit LDS, else 0
int melspec_generic_bluestein_L(int M, int compute_dtype) {
    int L = 1;
    while (L < 2 * M - 1) L <<= 1;
    return melspec_generic_lds_bytes(L, 1, compute_dtype) <= 160 * 1024 ? L : 0;
}

hipError_t melspec_generic_prepare(size_t lds_bytes) {
    const void* fns[2] = {reinterpret_cast<const void*>(&k_melspec_generic<double>),
                          reinterpret_cast<const void*>(&k_melspec_generic<float>)};
    for (const void* fn : fns) {
        // the attribute belongs to the kernel, not to a plan: only ever raised, to the device's limit
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes > 0 ? 160 * 1024 : 0);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int melspec_generic_pick_F(int M, int compute_dtype) {
    for (int F = 16; F >= 1; F >>= 1)
        if (melspec_generic_lds_bytes(M, F, compute_dtype) <= 64 * 1024) return F;
    return 0;
}

hipError_t launch_melspec_generic(const MelspecArgs& a, int compute_dtype, hipStream_t st) {
    const int tiles = (a.T + a.F - 1) / a.F;
    const dim3 grid(unsigned(a.n_items) * unsigned(tiles));
    const size_t lds = a.bl_L ? melspec_generic_lds_bytes(a.bl_L, 1, compute_dtype) : melspec_generic_lds_bytes(a.M, a.F, compute_dtype);
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(k_melspec_generic<double>, grid, dim3(256), lds, st, a);
    else
        hipLaunchKernelGGL(k_melspec_generic<float>, grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace aud
