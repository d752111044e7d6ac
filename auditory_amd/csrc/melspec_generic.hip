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
#include "frames_epilogue.h"

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
            case 9: stage<TT, 9>(src, dst, tw, 1, L, L, 1, ncur, s, tid); break;
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

// ---- the Bluestein route IN PLACE (round 5) ---------------------------------------------------------------------------------
// One complex buffer of L points instead of the autosort's two: a thread reads the inputs of ALL its butterflies of a stage into
// registers, the workgroup meets at a barrier, and the outputs go back into the same buffer at the autosort's positions -- twice
// the barriers, half the LDS: the 2304-point transform of N = 1103 in float64 takes 39 KB per workgroup, FOUR workgroups (16
// waves) per CU where the two-buffer form had two -- and this kernel waits (barriers, LDS round trips) for most of its life.
// The buffer is padded by one element per 16 (padx): the first stages store with a stride of 16 elements = 256 bytes, i.e.
// sixteen lanes of a 16-byte store on the same four banks; with the pad the stride is 17 elements and the lanes fan out.
__host__ __device__ __forceinline__ int padx(int i) { return i + (i >> 4); }

// `pre(i, z)`: what element i of the buffer stands for when a stage LOADS it -- the identity, or (first stage of a transform) the
// chirp / bhat multiplication that would otherwise be a pass of its own over the buffer with a barrier behind it
struct PreNone {
    template <typename Z>
    __device__ __forceinline__ Z operator()(int, Z z) const { return z; }
};
// LIN (chosen per stage by stage_inplace_any): equally spaced padded positions on both sides
template <typename TT, int P, int R, bool LIN, typename PRE>
__device__ __forceinline__ void stage_inplace(C2<TT>* buf, const C2<TT>* __restrict__ tw, int L, int ncur, int s, int tws, int tid, PRE pre) {
    // (tws: the stage's twiddle W_ncur^(q j) is tw[j q tws] -- s for one transform of length L over the table W_L; the batched
    //  plain route passes ratio x (product of the radices so far) over the table W_N)
    const int m = ncur / P, nb = L / P, sm = s * m;
    // wave-uniform facts that keep the per-element index arithmetic off the vector unit: s is a power of two in every stage
    // but those behind a radix 3 / 5 / 9 one (shift instead of a division); with a stride that is a multiple of 16 the padded
    // positions of a butterfly's elements are equally spaced (padx(x + i d) = padx(x) + i (d + d / 16))
    const int s_log = (s & (s - 1)) == 0 ? 31 - __builtin_clz(unsigned(s)) : -1;
    constexpr bool lin_in = LIN, lin_out = LIN;
    const int din = sm + (sm >> 4), dout = s == 1 ? 1 : s + (s >> 4);
    C2<TT> v[R][P];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int b = tid + 256 * r;
        if (b < nb) {
            const int q = s_log >= 0 ? b >> s_log : b / s, k = b - q * s;
            const int x0 = k + s * q;
            const C2<TT>* in = buf + padx(x0);
            // (in batches of eight: `pre` may bring a table value per element -- sixteen of them in flight beside sixteen
            //  float64 values do not fit 128 registers)
            constexpr int kB = sizeof(TT) == 8 ? 8 : (P > 8 ? P : 8);  // (float32: everything at once, 79 registers)
#pragma unroll
            for (int i0 = 0; i0 < P; i0 += kB) {
#pragma unroll
                for (int u = 0; u < kB; ++u)
                    if (i0 + u < P) v[r][i0 + u] = pre(x0 + (i0 + u) * sm, lin_in ? in[(i0 + u) * din] : buf[padx(x0 + (i0 + u) * sm)]);
#if defined(__HIP_DEVICE_COMPILE__)
                if constexpr (sizeof(TT) == 8 && P > 8) __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
    }
    __syncthreads();  // every input of the stage is in registers
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int b = tid + 256 * r;
        if (b < nb) {
            const int q = s_log >= 0 ? b >> s_log : b / s, k = b - q * s;
            SmallDft<TT, P>::run(v[r], tw, L);
            const unsigned tq = unsigned(q) * unsigned(tws);
            const int y0 = k + s * P * q;
            C2<TT>* out = buf + padx(y0);
            out[0] = v[r][0];
            if (s * P == L) {  // the last stage (wave-uniform): q = 0 everywhere, no twiddles
#pragma unroll
                for (int j = 1; j < P; ++j) (lin_out ? out[j * dout] : buf[padx(y0 + j * s)]) = v[r][j];
            } else {
                // (twiddles in batches of eight: all fifteen of a radix-16 butterfly in flight beside its sixteen values do
                //  not fit the 128 registers four waves per SIMD leave a thread)
                if constexpr (sizeof(TT) == 4) {
#pragma unroll
                    for (int j = 1; j < P; ++j) (lin_out ? out[j * dout] : buf[padx(y0 + j * s)]) = cmul<TT>(v[r][j], tw[unsigned(j) * tq]);
                }
                constexpr int kB = 8;
#pragma unroll
                for (int j0 = 1; j0 < (sizeof(TT) == 8 ? P : 0); j0 += kB) {
                    C2<TT> w8[kB];
#pragma unroll
                    for (int u = 0; u < kB; ++u)
                        if (j0 + u < P) w8[u] = tw[unsigned(j0 + u) * tq];
#pragma unroll
                    for (int u = 0; u < kB; ++u)
                        if (j0 + u < P) (lin_out ? out[(j0 + u) * dout] : buf[padx(y0 + (j0 + u) * s)]) = cmul<TT>(v[r][j0 + u], w8[u]);
#if defined(__HIP_DEVICE_COMPILE__)
                    if constexpr (sizeof(TT) == 8 && P > 8) __builtin_amdgcn_sched_barrier(0);
#endif
                }
            }
        }
    }
    __syncthreads();
}

// a stage's butterflies L / P over 256 threads: R rounds held in registers at once, P x R <= 16 complex values
__host__ __device__ inline int inplace_rounds(int L, int p) { return (L / p + 255) / 256; }
__host__ __device__ inline bool inplace_radix_ok(int L, int p) {
    const int r = inplace_rounds(L, p);
    switch (p) {
        case 16: return r <= 1;
        case 9: return r <= 1;
        case 8: return r <= 2;
        case 7: return r <= 2;
        case 5: return r <= 3;
        case 4: return r <= 4;
        case 3: return r <= 5;
        case 2: return r <= 8;
        default: return false;  // (25 and the O(p) pass: the two-buffer route)
    }
}

template <typename TT, typename PRE>
__device__ __forceinline__ void stage_inplace_any(int p, C2<TT>* buf, const C2<TT>* __restrict__ tw, int L, int ncur, int s, int tws, int tid, PRE pre) {
    // equally spaced padded positions: inputs sm apart with sm a multiple of 16; outputs s apart with s a multiple of 16, or
    // s = 1 under radix 16 (y0 = 16 q: the 16 outputs share one pad group)
    const int sm = s * (ncur / p);
    const bool lin = (sm & 15) == 0 && ((s & 15) == 0 || (s == 1 && p == 16));
    switch (p) {
        case 16:
            if (lin) stage_inplace<TT, 16, 1, true>(buf, tw, L, ncur, s, tws, tid, pre);
            else stage_inplace<TT, 16, 1, false>(buf, tw, L, ncur, s, tws, tid, pre);
            break;
        case 9:
            if (lin) stage_inplace<TT, 9, 1, true>(buf, tw, L, ncur, s, tws, tid, pre);
            else stage_inplace<TT, 9, 1, false>(buf, tw, L, ncur, s, tws, tid, pre);
            break;
        case 8: stage_inplace<TT, 8, 2, false>(buf, tw, L, ncur, s, tws, tid, pre); break;
        case 7: stage_inplace<TT, 7, 2, false>(buf, tw, L, ncur, s, tws, tid, pre); break;
        case 5: stage_inplace<TT, 5, 3, false>(buf, tw, L, ncur, s, tws, tid, pre); break;
        case 4: stage_inplace<TT, 4, 4, false>(buf, tw, L, ncur, s, tws, tid, pre); break;
        case 3: stage_inplace<TT, 3, 5, false>(buf, tw, L, ncur, s, tws, tid, pre); break;
        default: stage_inplace<TT, 2, 8, false>(buf, tw, L, ncur, s, tws, tid, pre); break;
    }
}
// the transform in place; `pre` rides on the FIRST stage's loads (the caller's barrier stands before the call)
template <typename TT, typename PRE>
__device__ __forceinline__ void smooth_fft_inplace(C2<TT>* buf, const MelspecArgs& a, int tid, PRE pre) {
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.bl_tw);
    const int L = a.bl_L;
    int ncur = L, s = 1;
    stage_inplace_any<TT>(a.bl_fac[0], buf, tw, L, ncur, s, s, tid, pre);
    ncur /= a.bl_fac[0];
    s *= a.bl_fac[0];
    for (int stg = 1; stg < a.bl_nfac; ++stg) {
        const int p = a.bl_fac[stg];
        stage_inplace_any<TT>(p, buf, tw, L, ncur, s, s, tid, PreNone());
        ncur /= p;
        s *= p;
    }
}

// ---- smooth window lengths IN PLACE (round 6): F frames per workgroup as ONE batched transform ---------------------------------
// The autosort stage x[k + s (q + m i)] -> y[k + s (P q + j)] W_ncur^(q j) treats k < s as a batch index: started at s = F
// (instead of 1) over a buffer that holds element n of frame f at position f + F n, the stages above run F independent M-point
// transforms as if they were the tail of one transform of length F M -- same butterflies per thread, same padded buffer, same
// batched loads as the Bluestein route, one buffer instead of the two-buffer route's two.  X_f[k] ends at position f + F k.
// Twiddles from the plan's W_N table: W_ncur^(q j) = W_N^(ratio prod q j), prod = the radices so far.
template <typename TT>
__device__ __forceinline__ void plain_fft_inplace(C2<TT>* buf, const MelspecArgs& a, int tid) {
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.tw);
    const int L = a.F * a.M;
    int ncur = a.M, s = a.F, tws = a.ratio;
    for (int stg = 0; stg < a.ip_nfac; ++stg) {
        const int p = a.ip_fac[stg];
        stage_inplace_any<TT>(p, buf, tw, L, ncur, s, tws, tid, PreNone());
        ncur /= p;
        s *= p;
        tws *= p;
    }
}

// INPL: the instantiation of the in-place Bluestein route alone -- without the two-buffer stages (whose radix-25 butterfly
// holds 25 complex values per thread: 156 registers in float64, three waves per SIMD) it stays within 128 registers, and
// four workgroups per CU is what the one-buffer layout is for
template <typename TT, bool INPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1 + 3 * int(INPL))))
void k_melspec_generic(const MelspecArgs a) {
    unsigned char* smem = dyn_lds();
    const int tid = threadIdx.x;
    const int F = a.F, M = a.M, N = a.N, H = a.H, T = a.T;
    constexpr bool inpl = INPL;  // Bluestein in ONE padded buffer (element i at padx(i)); the host launches it for bl_inplace plans
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
    int* pair_exp = reinterpret_cast<int*>(smem + (inpl ? size_t(padx(a.bl_L)) : 2 * size_t(a.bl_L ? a.bl_L : F * M)) * sizeof(C2<TT>));  // [2]
    if (pair) {
        if (tid < 2) pair_exp[tid] = kNoSignal;
        __syncthreads();
    }

    // ---- gather the F windows (sndenv.go:455-478) -------------------------------------
    // frame by frame (what a frame is -- live, where it starts -- is workgroup-uniform), a thread's samples 256 apart; the pair
    // route's per-frame exponent is the maximum over the thread's own samples first, then over the wave (DPP), then ONE LDS
    // atomic per wave and frame (round 6: an atomic per sample was 2 206 serialised LDS operations per workgroup, and with
    // the division of the flat index by N a quarter of the kernel's time: profiles/round6_cfg1_ablation.txt)
    const int64_t stride = it.sig_stride > 1 ? it.sig_stride : 1;
    for (int f = 0; f < F; ++f) {
        const int s = t0 + f;
        const int64_t start = int64_t(it.start0) + int64_t(a.S) * (s - a.border);
        const bool live = s < T && start + N <= int64_t(it.sig_len);
        int ex_max = kNoSignal;
        // what a sample is turned into and where it goes (uniform route flags; positions of frame f)
        auto put = [&](int n, TT v) {
            if (inpl && !pair) {  // z[n / 2] = (x[2j], x[2j+1]) for even N, z[n] = (x[n], 0) for odd N; element j of frame f at f + F j
                TT* cell = &src[padx(f + F * (even ? n >> 1 : n))].x;  // (Bluestein without a pair: F = 1)
                if (even) cell[n & 1] = v;
                else { cell[0] = v; cell[1] = TT(0); }
            } else if (even) {
                reinterpret_cast<TT*>(src)[size_t(f) * N + n] = v;  // z[n/2] = (x[2j], x[2j+1])
            } else if (pair) {
                (&src[inpl ? padx(n) : n].x)[f] = v;   // frame 0: real parts, frame 1: imaginary parts
                // an Inf / NaN sample takes its frame OUT of the pair (sentinel exponent): the frame's bins are NaN, as a transform of
                // its own would leave them, and its partner -- an independent frame in the reference (dft.go:42-50) -- runs alone
                const int ex = (v - v == TT(0)) ? amax_exponent<TT>(v < TT(0) ? -v : v) : kNonFinite;
                ex_max = ex > ex_max ? ex : ex_max;
            } else {
                src[size_t(f) * M + n] = {v, TT(0)};
            }
        };
        // one loop per sample type (a launch constant), the frame's first sample as a pointer, 32-bit positions inside the frame:
        // the loop had carried a 64-bit position, a 64-bit product with the stride and a three-way type switch per sample
        const int lo = !live ? N : start < 0 ? int(-start < int64_t(N) ? -start : int64_t(N)) : 0;  // first sample that is not pad (N: none)
        auto run = [&](auto* stream) {
            auto* first = stream + it.sig_off + start * stride;   // (dereferenced for lo <= n < N only: samples of the stream)
            for (int n = tid; n < N; n += blockDim.x) {
                TT v = TT(0);
                if (n >= lo) {
                    if constexpr (sizeof(*stream) == 2) v = pcm16_to<TT>(int(first[int64_t(n) * stride]));
                    else v = TT(first[int64_t(n) * stride]);
                }
                put(n, v);
            }
        };
        if (a.sig_dtype == AUD_F32) run(static_cast<const float*>(a.sig));
        else if (a.sig_dtype == AUD_F64) run(static_cast<const double*>(a.sig));
        else run(static_cast<const int16_t*>(a.sig));
        if (pair) {  // (uniform)
            ex_max = wave_max_i32(ex_max);
            if ((tid & 63) == 0 && ex_max != kNoSignal) atomicMax(pair_exp + f, ex_max);
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
        if constexpr (inpl) {
            // the three multiplications ride on loads: z . chirp and (.) . bhat, conjugated, on the first stage of the two
            // transforms; the last one, chirp . conj(.), on the power pass below (zat)
            smooth_fft_inplace<TT>(src, a, tid, [&](int i, C2<TT> z) {
                if (i >= M) return C2<TT>{TT(0), TT(0)};  // (whatever the buffer holds behind the window)
                if (pair) z = C2<TT>{x0 == kNonFinite ? TT(0) : scale2(z.x, -e0), x1 == kNonFinite ? TT(0) : scale2(z.y, -e1)};
                return cmul<TT>(z, chirp[i]);
            });
            smooth_fft_inplace<TT>(src, a, tid, [&](int i, C2<TT> z) {
                const C2<TT> c = cmul<TT>(z, bhat[i]);
                return C2<TT>{c.x, -c.y};
            });
        } else {
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
    }

    // ---- smooth lengths in place: the F frames as one batched transform (plain_fft_inplace) ----
    const bool plain_ip = inpl && a.ip_nfac > 0;  // (uniform; the host sets ip_nfac only on plans without a Bluestein length)
    if constexpr (inpl) {
        if (plain_ip) plain_fft_inplace<TT>(src, a, tid);
    }

    // ---- Stockham stages: x[k + s(q + m i)] -> y[k + s(p q + j)] * W_ncur^(q j) ---------
    int ncur = M, s = 1;
    for (int stg = 0; stg < ((inpl || a.bl_L) ? 0 : a.nfac); ++stg) {
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
            case 9: stage<TT, 9>(src, dst, tw, F, M, N, a.ratio, ncur, s, tid); break;
            case 7: stage<TT, 7>(src, dst, tw, F, M, N, a.ratio, ncur, s, tid); break;
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
    // (in place: the spectrum Z[0 .. M) sits at the padded positions below padx(M); everything of the buffer behind it is free
    //  once the last transform is done, and L >= 2 M - 1 leaves room for F rows of H values there)
    // (smooth lengths in place: the buffer is full of spectra; the launch's LDS has room for the F rows behind it)
    TT* P = inpl ? reinterpret_cast<TT*>(src + padx(plain_ip ? F * M : M) + 1) : reinterpret_cast<TT*>(dst);
    const int Hp = H | 1;
    // Z[k] of frame f.  In place (one frame, or a pair in one transform): the buffer still holds the second transform's raw
    // output -- the convolution's last step, chirp . conj(.), is applied here, on the load
    const C2<TT>* __restrict__ chirp_z = static_cast<const C2<TT>*>(a.bl_chirp);
    auto zat = [&](int f, int k) {
        if constexpr (inpl) {
            if (plain_ip) return src[padx(f + F * k)];
            const C2<TT> r = src[padx(k)];
            return cmul<TT>(chirp_z[k], C2<TT>{r.x, -r.y});
        } else {
            return src[size_t(f) * M + k];
        }
    };
    if (pair) {  // (uniform) both frames' bins from ONE read of Z[k] and Z[M - k]
        const int ex0 = pair_exp[0], ex1 = pair_exp[1];
        for (int k = tid; k < H; k += blockDim.x) {
            const C2<TT> A = zat(0, k), B = zat(0, k == 0 ? 0 : M - k);
            {
                const TT re = (A.x + B.x) * TT(0.5), im = (A.y - B.y) * TT(0.5);
                P[k] = ex0 == kNoSignal ? TT(0) : ex0 == kNonFinite ? TT(__builtin_nan("")) : scale2(re * re + im * im, 2 * ex0);
            }
            {
                const TT re = (A.y + B.y) * TT(0.5), im = (B.x - A.x) * TT(0.5);
                P[size_t(Hp) + k] = ex1 == kNoSignal ? TT(0) : ex1 == kNonFinite ? TT(__builtin_nan("")) : scale2(re * re + im * im, 2 * ex1);
            }
        }
    }
    const float h_inv = 1.0f / float(H);  // (w < F H <= 2^17: (w + 0.5) / H is at least 0.5 / H from an integer, the product's error below 16 x 2^-22)
    for (int w = tid; w < (pair ? 0 : F * H); w += blockDim.x) {
        const int f = int((float(w) + 0.5f) * h_inv), k = w - f * H;
        TT re, im;
        if (even) {
            // X[k] = (Z[k] + conj Z[M-k])/2 - i W_N^k (Z[k] - conj Z[M-k])/2, Z[M] == Z[0]
            const C2<TT> A = zat(f, k == M ? 0 : k);
            const C2<TT> Bc = zat(f, k == 0 ? 0 : M - k);
            const TT er = (A.x + Bc.x) * TT(0.5), ei = (A.y - Bc.y) * TT(0.5);
            const TT dr = (A.x - Bc.x) * TT(0.5), di = (A.y + Bc.y) * TT(0.5);
            const C2<TT> wk = tw[k];
            // -i * (dr + i di) = di - i dr
            re = er + (di * wk.x + dr * wk.y);
            im = ei + (di * wk.y - dr * wk.x);
        } else {
            const C2<TT> z = zat(f, k);
            re = z.x;
            im = z.y;
        }
        P[size_t(f) * Hp + k] = re * re + im * im;  // dft.go:64-66
    }
    __syncthreads();

    if (a.bl_L) frames_epilogue<TT, false>(a, it, item, tiles, t0, P, tid);  // (uniform) one or two frames: a slot per (frame, filter)
    else frames_epilogue<TT, true>(a, it, item, tiles, t0, P, tid);           // up to sixteen: a slot per filter walks the frames
}

}  // namespace

// the Bluestein route runs in ONE padded buffer where every stage of L fits a thread's registers (stage_inplace)
bool melspec_generic_bluestein_inplace(int L) {
    int m = L;
    for (int p : {16, 8, 4, 2, 25, 5, 9, 3})
        while (m % p == 0) {
            if (!inplace_radix_ok(L, p)) return false;
            m /= p;
        }
    return m == 1;
}

// the fused tail parks F x nf log-mel values behind the F power spectra (F x (H | 1) values) in the buffer the spectra live in:
// the second Stockham buffer (F M complex values; L for a two-buffer Bluestein plan), or -- in place -- what the padded buffer
// has left behind Z[0 .. M) and one element
bool melspec_generic_tail_fits(int M, int F, int H, int nf, int compute_dtype, int bl_L, bool bl_inplace) {
    const size_t tsz = compute_dtype == AUD_F64 ? 8 : 4, c = 2 * tsz;
    const size_t need = (size_t(F) * size_t(H | 1) + size_t(F) * size_t(nf)) * tsz;
    if (bl_L && bl_inplace) return size_t(padx(M) + 1) * c + need <= size_t(padx(bl_L)) * c;
    return need <= (bl_L ? size_t(bl_L) : size_t(F) * size_t(M)) * c;
}

size_t melspec_generic_lds_bytes(int M, int F, int compute_dtype, bool bluestein) {
    const size_t c = compute_dtype == AUD_F64 ? 16 : 8;
    if (bluestein && melspec_generic_bluestein_inplace(M)) return size_t(padx(M)) * c + 16;
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
                for (int r : {16, 8, 4, 2, 25, 5, 9, 3})
                    while (m % r == 0) {
                        m /= r;
                        stages += r == 25 ? 1.75 : 1.0;
                    }
                // the kernel waits for most of its life (barriers, LDS round trips): what counts beside length x stages is how
                // many workgroups a CU's 160 KB of LDS hold at once.  Round 4's fit to N = 1103 in float64 (two buffers, us per
                // 256 segments of 14 frames: L 2304 101, 2560 106, 2400 120, 2500 122, 3072 191, 4096 194) gave 0.55 for two
                // against one; the in-place route fits four of L = 2304 (tools/tune_bluestein.sh, profiles/round5_bluestein_*)
                const size_t lds_l = melspec_generic_lds_bytes(int(L), 1, compute_dtype, true);
                const int per_cu = int(std::min<size_t>(4, size_t(160 * 1024) / lds_l));
                const double fit2 = per_cu >= 4 ? 0.35 : per_cu == 3 ? 0.42 : per_cu == 2 ? 0.55 : 1.0;
                const double cost = double(L) * stages * fit2;
                if (best == 0 || cost < best_cost) {
                    best = int(L);
                    best_cost = cost;
                }
            }
    return best;
}

// Smooth window lengths in place (plain_fft_inplace): which F, which radices, how much LDS.  The buffer holds F M complex values
// (padded), the F power spectra and the fused tail's F x nf log-mel values live behind it.  F is a power of two (the batch index
// is the low part of every position: shifts) -- the largest that keeps the workgroup at <= 40 KB (four workgroups per CU: this
// kernel waits for barriers and LDS round trips for most of its life), with fewer workgroups per CU where one
// frame alone needs more, and never more than the segment has steps (rounded up to a power of two).
int melspec_generic_plain_inplace(int M, int H, int nf, int T, int compute_dtype, int forced_F, int* fac, int* nfac, size_t* lds) {
    int n = 0, m = M;
    for (int p : {16, 8, 4, 2, 9, 7, 5, 3})
        while (m % p == 0) {
            if (n >= kMaxFactors) return 0;
            fac[n++] = p;
            m /= p;
        }
    if (m != 1 || n == 0) return 0;  // (a prime factor above 7: the two-buffer route's O(p) pass, or Bluestein)
    const size_t tsz = compute_dtype == AUD_F64 ? 8 : 4, c = 2 * tsz;
    auto bytes = [&](int F) { return size_t(padx(F * M) + 1) * c + (size_t(F) * size_t(H | 1) + size_t(F) * size_t(nf)) * tsz + 16; };
    auto runs = [&](int F) {
        for (int i = 0; i < n; ++i)
            if (!inplace_radix_ok(F * M, fac[i])) return false;
        return true;
    };
    int tcap = 1;
    while (tcap < T) tcap <<= 1;
    int best = 0;
    for (int F = 16; F >= 1; F >>= 1) {
        if (forced_F > 0 && F != forced_F) continue;
        if (F > tcap || !runs(F)) continue;
        const size_t b = bytes(F);
        if (b > size_t(160) * 1024) continue;
        best = F;  // (descending: without a break this ends at the smallest F that fits at all -- taken when none fits 40 KB)
        if (forced_F > 0 || b <= size_t(40) * 1024) break;
    }
    if (best == 0) return 0;
    *nfac = n;
    *lds = bytes(best);
    return best;
}

hipError_t melspec_generic_prepare(size_t lds_bytes) {
    const void* fns[4] = {reinterpret_cast<const void*>(&k_melspec_generic<double, false>),
                          reinterpret_cast<const void*>(&k_melspec_generic<float, false>),
                          reinterpret_cast<const void*>(&k_melspec_generic<double, true>),
                          reinterpret_cast<const void*>(&k_melspec_generic<float, true>)};
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
    // (one frame per workgroup in up to the CU's whole LDS: the plan raises the kernel's limit, melspec_generic_prepare)
    return melspec_generic_lds_bytes(M, 1, compute_dtype, false) <= 160 * 1024 ? 1 : 0;
}

hipError_t launch_melspec_generic(const MelspecArgs& a, int compute_dtype, hipStream_t st) {
    const int tiles = (a.T + a.F - 1) / a.F;
    const dim3 grid(unsigned(a.n_items) * unsigned(tiles));
    const bool plain_ip = a.bl_L == 0 && a.ip_nfac > 0;
    const size_t tsz = compute_dtype == AUD_F64 ? 8 : 4;
    const size_t lds = plain_ip ? size_t(padx(a.F * a.M) + 1) * 2 * tsz + (size_t(a.F) * size_t(a.H | 1) + size_t(a.F) * size_t(a.nf)) * tsz + 16
                       : a.bl_L ? melspec_generic_lds_bytes(a.bl_L, 1, compute_dtype, true) : melspec_generic_lds_bytes(a.M, a.F, compute_dtype, false);
    const bool inpl = plain_ip || (a.bl_L != 0 && a.bl_inplace != 0);
    if (compute_dtype == AUD_F64) {
        if (inpl) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_generic<double, true>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_generic<double, false>), grid, dim3(256), lds, st, a);
    } else {
        if (inpl) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_generic<float, true>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_generic<float, false>), grid, dim3(256), lds, st, a);
    }
    return hipGetLastError();
}

}  // namespace aud
