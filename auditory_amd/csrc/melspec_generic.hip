// Any-N frame -> FFT -> power -> mel -> log kernel.
//
// One workgroup (256 threads) owns F consecutive frames of one work item.  The frames are
// gathered into LDS, transformed with an autosort Stockham FFT whose stages ping-pong
// between two LDS buffers, turned into a power spectrum, and reduced through the mel
// triangle table.  Even N uses the packed-real trick (an N/2-point complex FFT plus one
// split pass); odd N (e.g. the prime 1103 that 25 ms @ 44.1 kHz produces) runs a full
// N-point complex FFT.  Radix 2, 3, 4, 5, 8, 16 and 25 stages run whole butterflies in registers
// (the host factorises M into as few of them as possible).  A length with any other prime factor
// p takes Bluestein's chirp convolution (two FFTs of a 2-3-5-smooth length L >= 2 M - 1 per workgroup; on odd window
// lengths they carry TWO real frames) where its two buffers fit LDS, and an O(p) per-output pass otherwise, so every N is
// supported.  This is the universal path; the common sizes have faster specialised kernels.
//
// Reference semantics implemented here: sound/sndenv.go:438-478 (window extraction,
// left zero pad, short-signal masking), dft/dft.go:53-85 (DFT of the raw window, power,
// log(power+offset)), mel/mel.go:120-153 (triangle sums, log, renorm).
#include <cstdlib>

#include "device_common.h"

namespace aud {
namespace {

constexpr int kNonFinite = 1 << 20;  // pair route: sentinel exponent of a frame that holds an Inf / NaN sample
__device__ __forceinline__ float scale2(float v, int e) { return ldexpf(v, e); }
__device__ __forceinline__ double scale2(double v, int e) { return ldexp(v, e); }

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

// the FFT of the Bluestein route: radix 16 / 8 / 4 / 2 / 25 / 5 / 3 stages of one transform of length L (2-3-5-smooth), result in `src`
template <typename TT>
__device__ __forceinline__ void smooth_fft(C2<TT>*& src, C2<TT>*& dst, const MelspecArgs& a, int tid) {
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.bl_tw);
    const int L = a.bl_L;
    int ncur = L, s = 1;
    for (int stg = 0; stg < a.bl_nfac; ++stg) {
        const int p = a.bl_fac[stg];
        switch (p) {
            case 16: stage<TT, 16>(src, dst, tw, 1, L, L, 1, ncur, s, tid); break;
            case 8: stage<TT, 8>(src, dst, tw, 1, L, L, 1, ncur, s, tid); break;
            case 4: stage<TT, 4>(src, dst, tw, 1, L, L, 1, ncur, s, tid); break;
            case 25: stage<TT, 25>(src, dst, tw, 1, L, L, 1, ncur, s, tid); break;
            case 5: stage<TT, 5>(src, dst, tw, 1, L, L, 1, ncur, s, tid); break;
            case 3: stage<TT, 3>(src, dst, tw, 1, L, L, 1, ncur, s, tid); break;
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
    // Bluestein route on an odd window length: TWO real frames ride one complex transform, z[n] = x_0[n] + i x_1[n]; they are
    // separated behind it: X_0[k] = (Z[k] + conj Z[M - k]) / 2, X_1[k] = (Z[k] - conj Z[M - k]) / 2i
    const bool pair = a.bl_L != 0 && !even && F == 2;
    // ... each frame first divided by 2^(exponent of its largest sample), so that both components of z are O(1): what leaks
    // from one frame into the other through rounding is then 2^-53 of the frame's OWN peak, as in a transform of its own (a
    // quiet frame beside a loud one would otherwise inherit the loud one's floor); the powers are scaled back exactly, and a
    // frame of exact zeros keeps an exactly zero spectrum (LogMin rule, mel.go:135-137)
    int* pair_exp = reinterpret_cast<int*>(smem + 2 * size_t(a.bl_L ? a.bl_L : F * M) * sizeof(C2<TT>));  // [2]
    if (pair) {
        if (tid < 2) pair_exp[tid] = kNoSignal;
        __syncthreads();
    }

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
        } else if (pair) {
            reinterpret_cast<TT*>(src)[2 * size_t(n) + f] = v;   // frame 0: real parts, frame 1: imaginary parts
            // an Inf / NaN sample takes its frame OUT of the pair (sentinel exponent): the frame's bins are NaN, as a transform of
            // its own would leave them, and its partner -- an independent frame in the reference (dft.go:42-50) -- runs alone
            const int ex = (v - v == TT(0)) ? amax_exponent<TT>(v < TT(0) ? -v : v) : kNonFinite;
            if (ex != kNoSignal) atomicMax(pair_exp + f, ex);
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
        const int x0 = pair ? pair_exp[0] : 0, x1 = pair ? pair_exp[1] : 0;
        const int e0 = (x0 == kNoSignal || x0 == kNonFinite) ? 0 : x0, e1 = (x1 == kNoSignal || x1 == kNonFinite) ? 0 : x1;
        for (int i = tid; i < L; i += blockDim.x) {
            C2<TT> z = i < M ? src[i] : C2<TT>{TT(0), TT(0)};
            if (pair) z = C2<TT>{x0 == kNonFinite ? TT(0) : scale2(z.x, -e0), x1 == kNonFinite ? TT(0) : scale2(z.y, -e1)};
            src[i] = i < M ? cmul<TT>(z, chirp[i]) : z;
        }
        __syncthreads();
        smooth_fft<TT>(src, dst, a, tid);
        for (int i = tid; i < L; i += blockDim.x) {
            const C2<TT> c = cmul<TT>(src[i], bhat[i]);
            src[i] = C2<TT>{c.x, -c.y};
        }
        __syncthreads();
        smooth_fft<TT>(src, dst, a, tid);
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
        } else if (pair) {
            const C2<TT> A = src[k], B = src[k == 0 ? 0 : M - k];
            if (f == 0) {
                re = (A.x + B.x) * TT(0.5);
                im = (A.y - B.y) * TT(0.5);
            } else {
                re = (A.y + B.y) * TT(0.5);
                im = (B.x - A.x) * TT(0.5);
            }
            const int ex = pair_exp[f];
            const TT p = ex == kNoSignal ? TT(0) : ex == kNonFinite ? TT(__builtin_nan("")) : scale2(re * re + im * im, 2 * ex);
            P[size_t(f) * Hp + k] = p;
            continue;
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

size_t melspec_generic_lds_bytes(int M, int F, int compute_dtype, bool bluestein) {
    const size_t c = compute_dtype == AUD_F64 ? 16 : 8;
    // two complex buffers; the Bluestein route (M = its transform length L, F = 1) adds the frame pair's two exponent words.
    // Plain Stockham plans stay at exactly 2 F M c: M = 2048 in float64 / 4096 in float32 fill the 64 KB to the byte.
    return size_t(2) * F * M * c + (bluestein ? 16 : 0);
}

// Bluestein: the transform length L >= 2 M - 1.  Any 2-3-5-smooth L the stage radices (16, 8, 4, 2, 25, 5, 3) cover will do;
// the cheapest by (length x number of stages) whose two complex buffers fit LDS is taken -- for M = 1103 that is
// 2304 = 16 x 16 x 3 x 3 (four stages over 2304 points, 74 KB in float64: two workgroups per CU) rather than the next
// power of two 4096 (three stages over 4096 points, 128 KB: one).  0: no length fits.
int melspec_generic_bluestein_L(int M, int compute_dtype) {
    const int64_t need = 2 * int64_t(M) - 1;
#ifdef AUD_TUNE_BLUESTEIN  // (tuning builds only: pick the length by hand)
    if (const char* env = getenv("AUD_BLUESTEIN_L")) {
        const int forced = atoi(env);
        if (forced >= need && melspec_generic_lds_bytes(forced, 1, compute_dtype, true) <= 160 * 1024) return forced;
    }
#endif
    int best = 0;
    double best_cost = 0.0;
    for (int64_t p5 = 1; p5 <= 4 * need; p5 *= 5)
        for (int64_t p3 = p5; p3 <= 4 * need; p3 *= 3)
            for (int64_t L = p3; L <= 4 * need; L *= 2) {
                if (L < need || L > (int64_t(1) << 20)) continue;
                if (melspec_generic_lds_bytes(int(L), 1, compute_dtype, true) > 160 * 1024) continue;
                double stages = 0;  // as capi.hip's factorize() will cut it; a radix-25 stage weighs 1.75 of the others
                int64_t m = L;
                for (int r : {16, 8, 4, 2, 25, 5, 3})
                    while (m % r == 0) {
                        m /= r;
                        stages += r == 25 ? 1.75 : 1.0;
                    }
                // two workgroups per CU when the buffers fit twice.  Fitted to N = 1103 in float64 on an MI355X (us per 256
                // segments of 14 frames): L 2304 101, 2560 106, 2400 120, 2500 122, 3072 191, 4096 194
                const double fit2 = 2 * melspec_generic_lds_bytes(int(L), 1, compute_dtype, true) <= 160 * 1024 ? 0.55 : 1.0;
                const double cost = double(L) * stages * fit2;
                if (best == 0 || cost < best_cost) {
                    best = int(L);
                    best_cost = cost;
                }
            }
    return best;
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
        if (melspec_generic_lds_bytes(M, F, compute_dtype, false) <= 64 * 1024) return F;
    return 0;
}

hipError_t launch_melspec_generic(const MelspecArgs& a, int compute_dtype, hipStream_t st) {
    const int tiles = (a.T + a.F - 1) / a.F;
    const dim3 grid(unsigned(a.n_items) * unsigned(tiles));
    const size_t lds = a.bl_L ? melspec_generic_lds_bytes(a.bl_L, 1, compute_dtype, true) : melspec_generic_lds_bytes(a.M, a.F, compute_dtype, false);
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(k_melspec_generic<double>, grid, dim3(256), lds, st, a);
    else
        hipLaunchKernelGGL(k_melspec_generic<float>, grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace aud
