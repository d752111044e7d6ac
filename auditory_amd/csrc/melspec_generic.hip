// Any-N frame -> FFT -> power -> mel -> log kernel.
//
// One workgroup (256 threads) owns F consecutive frames of one work item.  The frames are
// gathered into LDS, transformed with an autosort Stockham FFT whose stages ping-pong
// between two LDS buffers, turned into a power spectrum, and reduced through the mel
// triangle table.  Even N uses the packed-real trick (an N/2-point complex FFT plus one
// split pass); odd N (e.g. the prime 1103 that 25 ms @ 44.1 kHz produces) runs a full
// N-point complex FFT.  Radix 2, 3, 4 and 5 have dedicated butterflies; any other prime
// factor p goes through an O(p) per-output pass, so every N is supported.  This is the
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
        if (p == 2) {
            for (int w = tid; w < F * nb; w += blockDim.x) {
                const int f = w / nb, b = w - f * nb;
                const int q = b / s, k = b - q * s;
                const C2<TT>* x = src + size_t(f) * M + k + s * q;
                C2<TT>* y = dst + size_t(f) * M + k + s * 2 * q;
                const C2<TT> a0 = x[0], a1 = x[s * m];
                const C2<TT> w1 = tw[q * s * a.ratio];
                y[0] = {a0.x + a1.x, a0.y + a1.y};
                y[s] = cmul<TT>({a0.x - a1.x, a0.y - a1.y}, w1);
            }
        } else if (p == 4) {
            for (int w = tid; w < F * nb; w += blockDim.x) {
                const int f = w / nb, b = w - f * nb;
                const int q = b / s, k = b - q * s;
                const C2<TT>* x = src + size_t(f) * M + k + s * q;
                C2<TT>* y = dst + size_t(f) * M + k + s * 4 * q;
                const int sm = s * m;
                const C2<TT> a0 = x[0], a1 = x[sm], a2 = x[2 * sm], a3 = x[3 * sm];
                const C2<TT> e0 = {a0.x + a2.x, a0.y + a2.y}, e1 = {a0.x - a2.x, a0.y - a2.y};
                const C2<TT> o0 = {a1.x + a3.x, a1.y + a3.y}, o1 = {a1.x - a3.x, a1.y - a3.y};
                const int tq = q * s * a.ratio;
                y[0] = {e0.x + o0.x, e0.y + o0.y};
                y[s] = cmul<TT>({e1.x + o1.y, e1.y - o1.x}, tw[tq]);  // a0 - i a1 - a2 + i a3
                y[2 * s] = cmul<TT>({e0.x - o0.x, e0.y - o0.y}, tw[2 * tq]);
                y[3 * s] = cmul<TT>({e1.x - o1.y, e1.y + o1.x}, tw[3 * tq]);
            }
        } else if (p == 3) {
            const TT hs = TT(0.86602540378443864676L);  // sin(2 pi / 3)
            for (int w = tid; w < F * nb; w += blockDim.x) {
                const int f = w / nb, b = w - f * nb;
                const int q = b / s, k = b - q * s;
                const C2<TT>* x = src + size_t(f) * M + k + s * q;
                C2<TT>* y = dst + size_t(f) * M + k + s * 3 * q;
                const int sm = s * m;
                const C2<TT> a0 = x[0], a1 = x[sm], a2 = x[2 * sm];
                const C2<TT> t = {a1.x + a2.x, a1.y + a2.y};
                const C2<TT> mm = {a0.x - TT(0.5) * t.x, a0.y - TT(0.5) * t.y};
                const C2<TT> n = {hs * (a1.x - a2.x), hs * (a1.y - a2.y)};
                const int tq = q * s * a.ratio;
                y[0] = {a0.x + t.x, a0.y + t.y};
                y[s] = cmul<TT>({mm.x + n.y, mm.y - n.x}, tw[tq]);          // m - i n
                y[2 * s] = cmul<TT>({mm.x - n.y, mm.y + n.x}, tw[2 * tq]);  // m + i n
            }
        } else if (p == 5) {
            const TT c1 = TT(0.30901699437494742410L), c2 = TT(-0.80901699437494742410L);
            const TT s1 = TT(0.95105651629515357212L), s2 = TT(0.58778525229247312917L);
            for (int w = tid; w < F * nb; w += blockDim.x) {
                const int f = w / nb, b = w - f * nb;
                const int q = b / s, k = b - q * s;
                const C2<TT>* x = src + size_t(f) * M + k + s * q;
                C2<TT>* y = dst + size_t(f) * M + k + s * 5 * q;
                const int sm = s * m;
                const C2<TT> a0 = x[0], a1 = x[sm], a2 = x[2 * sm], a3 = x[3 * sm], a4 = x[4 * sm];
                const C2<TT> t1 = {a1.x + a4.x, a1.y + a4.y}, t2 = {a2.x + a3.x, a2.y + a3.y};
                const C2<TT> t3 = {a1.x - a4.x, a1.y - a4.y}, t4 = {a2.x - a3.x, a2.y - a3.y};
                const C2<TT> m1 = {a0.x + c1 * t1.x + c2 * t2.x, a0.y + c1 * t1.y + c2 * t2.y};
                const C2<TT> m2 = {a0.x + c2 * t1.x + c1 * t2.x, a0.y + c2 * t1.y + c1 * t2.y};
                const C2<TT> n1 = {s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y};
                const C2<TT> n2 = {s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y};
                const int tq = q * s * a.ratio;
                y[0] = {a0.x + t1.x + t2.x, a0.y + t1.y + t2.y};
                y[s] = cmul<TT>({m1.x + n1.y, m1.y - n1.x}, tw[tq]);          // m1 - i n1
                y[2 * s] = cmul<TT>({m2.x + n2.y, m2.y - n2.x}, tw[2 * tq]);  // m2 - i n2
                y[3 * s] = cmul<TT>({m2.x - n2.y, m2.y + n2.x}, tw[3 * tq]);  // m2 + i n2
                y[4 * s] = cmul<TT>({m1.x - n1.y, m1.y + n1.x}, tw[4 * tq]);  // m1 + i n1
            }
        } else {
            // one thread per output j of each radix-p butterfly
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
