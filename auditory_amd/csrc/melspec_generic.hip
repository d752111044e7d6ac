// Any-N frame -> FFT -> power -> mel -> log kernel.
//
// One workgroup (256 threads) owns F consecutive frames of one work item.  The frames are
// gathered into LDS, transformed with an autosort Stockham FFT whose stages ping-pong
// between two LDS buffers, turned into a power spectrum, and reduced through the mel
// triangle table.  Even N uses the packed-real trick (an N/2-point complex FFT plus one
// split pass); odd N (e.g. the prime 1103 that 25 ms @ 44.1 kHz produces) runs a full
// N-point complex FFT.  Radix 2, 3, 4, 5, 8, 16 and 25 stages run whole butterflies in registers
// (the host factorises M into as few of them as possible); any other prime factor p goes
// through an O(p) per-output pass, so every N is supported.  This is the
// universal path; the common power-of-two sizes have faster specialised kernels.
//
// Reference semantics implemented here: sound/sndenv.go:438-478 (window extraction,
// left zero pad, short-signal masking), dft/dft.go:53-85 (DFT of the raw window, power,
// log(power+offset)), mel/mel.go:120-153 (triangle sums, log, renorm).
#include "kernels.h"

namespace aud {
namespace {

template <typename TT>
struct C2 {
    TT x, y;
};

template <typename TT>
__device__ __forceinline__ C2<TT> cmul(C2<TT> a, C2<TT> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

template <typename TT>
__device__ __forceinline__ C2<TT> cadd(C2<TT> a, C2<TT> b) { return {a.x + b.x, a.y + b.y}; }
template <typename TT>
__device__ __forceinline__ C2<TT> csub(C2<TT> a, C2<TT> b) { return {a.x - b.x, a.y - b.y}; }
template <typename TT>
__device__ __forceinline__ C2<TT> mul_mi(C2<TT> a) { return {a.y, -a.x}; }  // * (-i)
template <typename TT>
__device__ __forceinline__ C2<TT> mul_pi(C2<TT> a) { return {-a.y, a.x}; }  // * (+i)

// ---- forward DFTs of small length on registers: u[k] = sum_n u[n] exp(-2 pi i n k / P) --------
template <typename TT>
__device__ __forceinline__ void dft2(C2<TT>& u0, C2<TT>& u1) {
    const C2<TT> t = u0;
    u0 = cadd(t, u1);
    u1 = csub(t, u1);
}
template <typename TT>
__device__ __forceinline__ void dft3(C2<TT>& u0, C2<TT>& u1, C2<TT>& u2) {
    const TT hs = TT(0.86602540378443864676L);  // sin(2 pi / 3)
    const C2<TT> t = cadd(u1, u2);
    const C2<TT> mm = {u0.x - TT(0.5) * t.x, u0.y - TT(0.5) * t.y};
    const C2<TT> n = {hs * (u1.x - u2.x), hs * (u1.y - u2.y)};
    u0 = cadd(u0, t);
    u1 = cadd(mm, mul_mi(n));
    u2 = cadd(mm, mul_pi(n));
}
template <typename TT>
__device__ __forceinline__ void dft4(C2<TT>& u0, C2<TT>& u1, C2<TT>& u2, C2<TT>& u3) {
    const C2<TT> s0 = cadd(u0, u2), d0 = csub(u0, u2);
    const C2<TT> s1 = cadd(u1, u3), d1 = csub(u1, u3);
    u0 = cadd(s0, s1);
    u2 = csub(s0, s1);
    u1 = cadd(d0, mul_mi(d1));
    u3 = cadd(d0, mul_pi(d1));
}
template <typename TT>
__device__ __forceinline__ void dft5(C2<TT>& u0, C2<TT>& u1, C2<TT>& u2, C2<TT>& u3, C2<TT>& u4) {
    const TT c1 = TT(0.30901699437494742410L), c2 = TT(-0.80901699437494742410L);
    const TT s1 = TT(0.95105651629515357212L), s2 = TT(0.58778525229247312917L);
    const C2<TT> t1 = cadd(u1, u4), t2 = cadd(u2, u3), t3 = csub(u1, u4), t4 = csub(u2, u3);
    const C2<TT> m1 = {u0.x + c1 * t1.x + c2 * t2.x, u0.y + c1 * t1.y + c2 * t2.y};
    const C2<TT> m2 = {u0.x + c2 * t1.x + c1 * t2.x, u0.y + c2 * t1.y + c1 * t2.y};
    const C2<TT> n1 = {s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y};
    const C2<TT> n2 = {s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y};
    u0 = {u0.x + t1.x + t2.x, u0.y + t1.y + t2.y};
    u1 = cadd(m1, mul_mi(n1));
    u4 = cadd(m1, mul_pi(n1));
    u2 = cadd(m2, mul_mi(n2));
    u3 = cadd(m2, mul_pi(n2));
}

// P-point DFT of v[0..P-1], natural order in and out.  tw / N give access to W_N^k for the
// composite sizes whose inner twiddles are not worth spelling out as literals (P = 25).
template <typename TT, int P>
struct SmallDft;
template <typename TT>
struct SmallDft<TT, 2> {
    static __device__ __forceinline__ void run(C2<TT> (&v)[2], const C2<TT>*, int) { dft2(v[0], v[1]); }
};
template <typename TT>
struct SmallDft<TT, 3> {
    static __device__ __forceinline__ void run(C2<TT> (&v)[3], const C2<TT>*, int) { dft3(v[0], v[1], v[2]); }
};
template <typename TT>
struct SmallDft<TT, 4> {
    static __device__ __forceinline__ void run(C2<TT> (&v)[4], const C2<TT>*, int) { dft4(v[0], v[1], v[2], v[3]); }
};
template <typename TT>
struct SmallDft<TT, 5> {
    static __device__ __forceinline__ void run(C2<TT> (&v)[5], const C2<TT>*, int) {
        dft5(v[0], v[1], v[2], v[3], v[4]);
    }
};
template <typename TT>
struct SmallDft<TT, 8> {  // 8 = 4 x 2: two 4-point DFTs (even / odd samples), then one radix-2 layer
    static __device__ __forceinline__ void run(C2<TT> (&v)[8], const C2<TT>*, int) {
        const TT r2 = TT(0.70710678118654752440L);
        dft4(v[0], v[2], v[4], v[6]);
        dft4(v[1], v[3], v[5], v[7]);
        const C2<TT> o1 = {(v[3].x + v[3].y) * r2, (v[3].y - v[3].x) * r2};   // * W8^1
        const C2<TT> o2 = mul_mi(v[5]);                                        // * W8^2
        const C2<TT> o3 = {(v[7].y - v[7].x) * r2, -(v[7].x + v[7].y) * r2};  // * W8^3
        const C2<TT> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
        v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
        v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
        v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
        v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
    }
};
template <typename TT>
struct SmallDft<TT, 16> {  // 16 = 4 x 4
    static __device__ __forceinline__ void run(C2<TT> (&v)[16], const C2<TT>*, int) {
        const TT c1 = TT(0.92387953251128675613L), s1 = TT(0.38268343236508977173L);
        const TT r2 = TT(0.70710678118654752440L);
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);
        v[5] = cmul(v[5], C2<TT>{c1, -s1});
        v[9] = C2<TT>{(v[9].x + v[9].y) * r2, (v[9].y - v[9].x) * r2};
        v[13] = cmul(v[13], C2<TT>{s1, -c1});
        v[6] = C2<TT>{(v[6].x + v[6].y) * r2, (v[6].y - v[6].x) * r2};
        v[10] = mul_mi(v[10]);
        v[14] = C2<TT>{(v[14].y - v[14].x) * r2, -(v[14].x + v[14].y) * r2};
        v[7] = cmul(v[7], C2<TT>{s1, -c1});
        v[11] = C2<TT>{(v[11].y - v[11].x) * r2, -(v[11].x + v[11].y) * r2};
        v[15] = cmul(v[15], C2<TT>{-c1, s1});
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j) {
                const C2<TT> t = v[4 * i + j];
                v[4 * i + j] = v[4 * j + i];
                v[4 * j + i] = t;
            }
    }
};
template <typename TT>
struct SmallDft<TT, 25> {  // 25 = 5 x 5, inner twiddles W25^(b k1) = W_N^(b k1 N / 25) from the table
    static __device__ __forceinline__ void run(C2<TT> (&v)[25], const C2<TT>* tw, int N) {
        const int w25 = N / 25;
#pragma unroll
        for (int b = 0; b < 5; ++b) dft5(v[b], v[5 + b], v[10 + b], v[15 + b], v[20 + b]);
#pragma unroll
        for (int k1 = 1; k1 < 5; ++k1)
#pragma unroll
            for (int b = 1; b < 5; ++b) v[5 * k1 + b] = cmul(v[5 * k1 + b], tw[b * k1 * w25]);
#pragma unroll
        for (int k1 = 0; k1 < 5; ++k1) dft5(v[5 * k1], v[5 * k1 + 1], v[5 * k1 + 2], v[5 * k1 + 3], v[5 * k1 + 4]);
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = i + 1; j < 5; ++j) {
                const C2<TT> t = v[5 * i + j];
                v[5 * i + j] = v[5 * j + i];
                v[5 * j + i] = t;
            }
    }
};

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

template <typename TT>
__device__ __forceinline__ TT load_sample(const void* sig, int dtype, int64_t i) {
    if (dtype == AUD_F32) return TT(static_cast<const float*>(sig)[i]);
    if (dtype == AUD_F64) return TT(static_cast<const double*>(sig)[i]);
    return TT(static_cast<const int16_t*>(sig)[i]) / TT(0x7FFF);  // sound.go:138
}

__device__ __forceinline__ float dev_log(float v) { return logf(v); }
__device__ __forceinline__ double dev_log(double v) { return log(v); }

template <typename TT>
__global__ __launch_bounds__(256) void k_melspec_generic(const MelspecArgs a) {
    unsigned char* smem = dyn_lds();
    const int tid = threadIdx.x;
    const int F = a.F, M = a.M, N = a.N, H = a.H, T = a.T;
    C2<TT>* src = reinterpret_cast<C2<TT>*>(smem);
    C2<TT>* dst = src + size_t(F) * M;
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.tw);

    const int tiles = (T + F - 1) / F;
    const int item = blockIdx.x / tiles;
    const int t0 = (blockIdx.x - item * tiles) * F;
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
        if (live && pos >= 0) v = load_sample<TT>(a.sig, a.sig_dtype, it.sig_off + pos);
        if (even) {
            reinterpret_cast<TT*>(src)[size_t(f) * N + n] = v;  // z[n/2] = (x[2j], x[2j+1])
        } else {
            src[size_t(f) * M + n] = {v, TT(0)};
        }
    }
    __syncthreads();

    // ---- Stockham stages: x[k + s(q + m i)] -> y[k + s(p q + j)] * W_ncur^(q j) ---------
    int ncur = M, s = 1;
    for (int stg = 0; stg < a.nfac; ++stg) {
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

int melspec_generic_pick_F(int M, int compute_dtype) {
    for (int F = 16; F >= 1; F >>= 1)
        if (melspec_generic_lds_bytes(M, F, compute_dtype) <= 64 * 1024) return F;
    return 0;
}

hipError_t launch_melspec_generic(const MelspecArgs& a, int compute_dtype, hipStream_t st) {
    const int tiles = (a.T + a.F - 1) / a.F;
    const dim3 grid(unsigned(a.n_items) * unsigned(tiles));
    const size_t lds = melspec_generic_lds_bytes(a.M, a.F, compute_dtype);
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(k_melspec_generic<double>, grid, dim3(256), lds, st, a);
    else
        hipLaunchKernelGGL(k_melspec_generic<float>, grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace aud
