// Device-side building blocks shared by the frame->mel kernel families: complex helpers, small
// in-register DFTs (2, 3, 4, 5, 8, 16, 25 points), sample loads, and the tile epilogue (optional
// power / log-power outputs, mel triangle reduction + log) that runs once the tile's power
// spectrum is parked in LDS.
#pragma once
#include <type_traits>
#include "kernels.h"

namespace aud {

template <typename TT>
struct alignas(2 * sizeof(TT)) C2 {
    TT x, y;
};
// four consecutive bins / weights moved as one 16-byte (f32) or 32-byte (f64) access
template <typename TT>
struct alignas(4 * sizeof(TT)) Q4 {
    TT x, y, z, w;
};
// the same four values where only 16-byte alignment is promised (weight rows of the wave kernels: their row stride is an
// odd number of 16-byte pieces); LDS moves 16 bytes per instruction anyway
template <typename TT>
struct alignas(16) Q4a {
    TT x, y, z, w;
};
// two complex values moved as one 16-byte (f32) LDS access
template <typename TT>
struct alignas(16) C2x2 {
    C2<TT> a, b;
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
// exp(-2 pi i m / 25), m = 0..16 (the inner twiddles of the 5 x 5 factorisation, m = b k1).  Literals rather than reads
// of the plan's W_N table: sixteen table values kept alive through the butterflies were what pushed the float64
// N = 400 kernel past its register budget (43 spilled registers, 38 MB of scratch traffic per launch).
template <typename TT>
__device__ __forceinline__ C2<TT> w25_literal(int m) {
    constexpr long double c[17] = {1.0L, 0.9685831611286311194902L, 0.8763066800438635873081L, 0.7289686274214115231467L,
                                   0.5358267949789966182713L, 0.3090169943749474241023L, 0.06279051952931337607618L,
                                   -0.1873813145857246305426L, -0.4257792915650726488625L, -0.6374239897486897101767L,
                                   -0.8090169943749474241023L, -0.9297764858882514036609L, -0.9921147013144778310498L,
                                   -0.9921147013144778310498L, -0.9297764858882514036609L, -0.8090169943749474241023L,
                                   -0.6374239897486897101767L};
    constexpr long double sn[17] = {0.0L, -0.2486898871648547882423L, -0.4817536741017152749872L, -0.6845471059286886737323L,
                                    -0.8443279255020150785486L, -0.9510565162951535721164L, -0.9980267284282715619523L,
                                    -0.9822872507286886810856L, -0.9048270524660195277137L, -0.770513242775789230803L,
                                    -0.5877852522924731291687L, -0.3681245526846779591569L, -0.1253332335643042453731L,
                                    0.1253332335643042453731L, 0.3681245526846779591569L, 0.5877852522924731291687L,
                                    0.770513242775789230803L};
    return C2<TT>{TT(c[m]), TT(sn[m])};
}
template <typename TT>
struct SmallDft<TT, 25> {  // 25 = 5 x 5, inner twiddles W25^(b k1)
    static __device__ __forceinline__ void run(C2<TT> (&v)[25], const C2<TT>*, int) {
#pragma unroll
        for (int b = 0; b < 5; ++b) dft5(v[b], v[5 + b], v[10 + b], v[15 + b], v[20 + b]);
#pragma unroll
        for (int k1 = 1; k1 < 5; ++k1)
#pragma unroll
            for (int b = 1; b < 5; ++b) v[5 * k1 + b] = cmul(v[5 * k1 + b], w25_literal<TT>(b * k1));
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

// 20 = 4 x 5 and 10 = 2 x 5 have coprime factors: the prime-factor (Good-Thomas) index maps
//   n = (N2 n1 + N1 n2) mod N,   k = (N2 (N2^-1 mod N1) k1 + N1 (N1^-1 mod N2) k2) mod N
// turn the DFT into N2 DFTs of length N1 and N1 of length N2 with NO twiddles between them; every index below is a
// compile-time constant, so the maps cost nothing (the unrolled arrays live in registers).
template <typename TT>
struct SmallDft<TT, 20> {  // n = (5 n1 + 4 n2) mod 20, k = (5 k1 + 16 k2) mod 20
    static __device__ __forceinline__ void run(C2<TT> (&v)[20], const C2<TT>*, int) {
#pragma unroll
        for (int n2 = 0; n2 < 5; ++n2)
            dft4(v[(4 * n2) % 20], v[(5 + 4 * n2) % 20], v[(10 + 4 * n2) % 20], v[(15 + 4 * n2) % 20]);  // n1 -> k1 in place
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1)
            dft5(v[(5 * k1) % 20], v[(5 * k1 + 4) % 20], v[(5 * k1 + 8) % 20], v[(5 * k1 + 12) % 20], v[(5 * k1 + 16) % 20]);
        C2<TT> t[20];
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 5; ++k2) t[(5 * k1 + 16 * k2) % 20] = v[(5 * k1 + 4 * k2) % 20];
#pragma unroll
        for (int k = 0; k < 20; ++k) v[k] = t[k];
    }
};
template <typename TT>
struct SmallDft<TT, 10> {  // n = (5 n1 + 2 n2) mod 10, k = (5 k1 + 6 k2) mod 10
    static __device__ __forceinline__ void run(C2<TT> (&v)[10], const C2<TT>*, int) {
#pragma unroll
        for (int n2 = 0; n2 < 5; ++n2) dft2(v[(2 * n2) % 10], v[(5 + 2 * n2) % 10]);
#pragma unroll
        for (int k1 = 0; k1 < 2; ++k1)
            dft5(v[(5 * k1) % 10], v[(5 * k1 + 2) % 10], v[(5 * k1 + 4) % 10], v[(5 * k1 + 6) % 10], v[(5 * k1 + 8) % 10]);
        C2<TT> t[10];
#pragma unroll
        for (int k1 = 0; k1 < 2; ++k1)
#pragma unroll
            for (int k2 = 0; k2 < 5; ++k2) t[(5 * k1 + 6 * k2) % 10] = v[(5 * k1 + 2 * k2) % 10];
#pragma unroll
        for (int k = 0; k < 10; ++k) v[k] = t[k];
    }
};

template <typename TT>
__device__ __forceinline__ TT load_sample(const void* sig, int dtype, int64_t i) {
    if (dtype == AUD_F32) return TT(static_cast<const float*>(sig)[i]);
    if (dtype == AUD_F64) return TT(static_cast<const double*>(sig)[i]);
    return TT(static_cast<const int16_t*>(sig)[i]) / TT(0x7FFF);  // sound.go:138
}

// int16 PCM / 0x7FFF (sound.go:138) in float32 without the full division sequence: q = x * RN(1/32767), one
// residual step.  Equal to the correctly rounded quotient for every int16 value (Markstein's correction step;
// tests/test_gpu_parity.py::test_int16_normalisation_exhaustive runs all 65536), i.e. to what load_sample returns.
__device__ __forceinline__ float pcm16_to_float(int v) {
    const float x = float(v);
    constexpr float r = 1.0f / 32767.0f;
    const float q = x * r;
    return fmaf(fmaf(-q, 32767.0f, x), r, q);
}

// the same quotient in float64 (what the reference computes: float64(v) / float64(0x7FFF)), again without the
// division sequence; exact for every int16 value (checked exhaustively on the host: tests/test_capi_host.py)
__device__ __forceinline__ double pcm16_to_double(int v) {
    const double x = double(v);
    constexpr double r = 1.0 / 32767.0;
    const double q = x * r;
    return fma(fma(-q, 32767.0, x), r, q);
}
template <typename TT>
__device__ __forceinline__ TT pcm16_to(int v) {
    if constexpr (sizeof(TT) == 4) return pcm16_to_float(v);
    else return pcm16_to_double(v);
}

// First-pass operands of one frame for the register-resident kernels: the frame's N samples as N/2 packed
// pairs z[n] = (x[2n], x[2n+1]); this lane takes z[lane + STRIDE n1], n1 = 0..NV-1.  Three routes:
//   1 float32 samples, frame inside the stream: one 8-byte load per pair (two 4-byte loads where the frame starts on
//     an odd sample), widened to the compute type in registers;
//   2 int16 samples under the same conditions: one 4-byte load per pair (two 2-byte loads on odd starts), normalised
//     on the fly (half the input bytes of the float route);
//   0 anything else (stream edges, left zero pad, float64 samples, strided streams): guarded element loads.
// Split in two so that a persistent kernel can request the next tile's operands before it computes the current
// one: frame_pairs_issue() only issues the loads of routes 1 / 2 (raw words, no wait), frame_pairs_take() converts
// them -- or runs route 0 on the spot.
template <int NV>
struct FrameRaw {
    uint2 w[NV];  // route 1: the two float bit patterns; route 2: .x = the packed int16 pair
    int route;
};
template <int NV, int STRIDE, int N, bool PCM16_ROUTE = true, bool ODD_ROUTE = false>
__device__ __forceinline__ void frame_pairs_issue(const MelspecArgs& a, const aud_item& it, int sstep, int lane,
                                                  FrameRaw<NV>& r) {
    const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
    const int64_t pos0 = start + 2 * lane;
    // a frame that starts on an odd sample (odd step lengths: 441 at 44.1 kHz) has pairs that straddle the 8-byte
    // (int16: 4-byte) grid.  ODD_ROUTE (the one-frame-per-wave kernel, where the parity is wave-uniform): its pairs are
    // fetched as two element loads each -- same raw words, same conversion; otherwise such frames take route 0 (in the
    // kernels with several frames per wave the second flavour of loads under a lane condition cost 5-10 % on even frames)
    // A strided stream (one channel of interleaved stereo, sig_stride = 2) is the same case: element loads, stride apart.
    const int64_t str = it.sig_stride > 1 ? it.sig_stride : 1;
    const bool even = ((it.sig_off + start) & 1) == 0 && str == 1;
    const bool inside = sstep < a.T && start >= 0 && start + N <= int64_t(it.sig_len) && (ODD_ROUTE || even);
    r.route = 0;
    if (inside && a.sig_dtype == AUD_F32 && (reinterpret_cast<uintptr_t>(a.sig) & 7) == 0) {
        if (!ODD_ROUTE || even) {
            const uint2* __restrict__ src =
                reinterpret_cast<const uint2*>(static_cast<const float*>(a.sig) + it.sig_off + pos0);
#pragma unroll
            for (int n1 = 0; n1 < NV; ++n1) r.w[n1] = src[STRIDE * n1];
        } else {
            const uint32_t* __restrict__ src =
                reinterpret_cast<const uint32_t*>(static_cast<const float*>(a.sig) + it.sig_off + pos0 * str);
#pragma unroll
            for (int n1 = 0; n1 < NV; ++n1) r.w[n1] = uint2{src[(2 * STRIDE * n1) * str], src[(2 * STRIDE * n1 + 1) * str]};
        }
        r.route = 1;
    }
    if constexpr (PCM16_ROUTE) {
        if (inside && a.sig_dtype == AUD_I16 && (reinterpret_cast<uintptr_t>(a.sig) & 3) == 0) {
            if (!ODD_ROUTE || even) {
                const uint32_t* __restrict__ src =
                    reinterpret_cast<const uint32_t*>(static_cast<const int16_t*>(a.sig) + it.sig_off + pos0);
#pragma unroll
                for (int n1 = 0; n1 < NV; ++n1) r.w[n1].x = src[STRIDE * n1];
            } else {
                const unsigned short* __restrict__ src =
                    reinterpret_cast<const unsigned short*>(static_cast<const int16_t*>(a.sig) + it.sig_off + pos0 * str);
#pragma unroll
                for (int n1 = 0; n1 < NV; ++n1)
                    r.w[n1].x = uint32_t(src[(2 * STRIDE * n1) * str]) | (uint32_t(src[(2 * STRIDE * n1 + 1) * str]) << 16);
            }
            r.route = 2;
        }
    }
}
template <typename TT, int NV, int STRIDE, int N, bool PCM16_ROUTE = true>
__device__ __forceinline__ void frame_pairs_take(const MelspecArgs& a, const aud_item& it, int sstep, int lane,
                                                 const FrameRaw<NV>& r, C2<TT> (&v)[NV]) {
    if (r.route == 1) {
#pragma unroll
        for (int n1 = 0; n1 < NV; ++n1)
            v[n1] = C2<TT>{TT(__uint_as_float(r.w[n1].x)), TT(__uint_as_float(r.w[n1].y))};
        return;
    }
    if constexpr (PCM16_ROUTE) {
        if (r.route == 2) {
#pragma unroll
            for (int n1 = 0; n1 < NV; ++n1) {
                v[n1].x = pcm16_to<TT>(int(int16_t(r.w[n1].x & 0xFFFFu)));
                v[n1].y = pcm16_to<TT>(int(int16_t(r.w[n1].x >> 16)));
            }
            return;
        }
    }
    const int64_t lim = it.sig_len;
    const int64_t pos0 = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border) + 2 * lane;
    const bool frame_on = sstep < a.T;
    const int64_t str = it.sig_stride > 1 ? it.sig_stride : 1;
#pragma unroll
    for (int n1 = 0; n1 < NV; ++n1) {
        const int64_t p = pos0 + 2 * STRIDE * n1;
        v[n1].x = (frame_on && p >= 0 && p < lim) ? load_sample<TT>(a.sig, a.sig_dtype, it.sig_off + p * str) : TT(0);
        v[n1].y = (frame_on && p + 1 >= 0 && p + 1 < lim)
                      ? load_sample<TT>(a.sig, a.sig_dtype, it.sig_off + (p + 1) * str) : TT(0);
    }
}
template <typename TT, int NV, int STRIDE, int N, bool PCM16_ROUTE = true, bool ODD_ROUTE = false>
__device__ __forceinline__ void load_frame_pairs(const MelspecArgs& a, const aud_item& it, int sstep, int lane,
                                                 C2<TT> (&v)[NV]) {
    FrameRaw<NV> r;
    frame_pairs_issue<NV, STRIDE, N, PCM16_ROUTE, ODD_ROUTE>(a, it, sstep, lane, r);
    frame_pairs_take<TT, NV, STRIDE, N, PCM16_ROUTE>(a, it, sstep, lane, r, v);
}

__device__ __forceinline__ float dev_log(float v) { return logf(v); }
__device__ __forceinline__ double dev_log(double v) { return log(v); }

// ln of a float64 band power whose result is stored as float32 (mel.go:133-139, dft.go:76-82): mantissa and exponent
// are split in float64 (two instructions), the logarithm of the mantissa is taken in float32 and the exponent's
// share added with one fused multiply-add.  Error: the float32 rounding of the mantissa (6e-8), logf's last place
// on a value in [-0.7, 0] and the rounding of the result -- all below the float32 spacing of the stored value --
// instead of ~100 float64 instructions and their constants.  Any magnitude a double can hold takes this route
// (no range branch); zero, negative and NaN inputs behave as in log().
// v = m 2^ex with 0.5 <= m < 1 (frexp), ln v = ex ln 2 + log2(m) ln 2 with log2 from the hardware's v_log_f32 (1 ulp):
// no range or denormal fix-ups are needed because m is always in [0.5, 1) -- 8 instructions instead of logf's 25.
__device__ __forceinline__ float mantissa_log(float m, int ex) {
    const float ln2 = 0.693147180559945309417f;
    return fmaf(float(ex), ln2, __builtin_amdgcn_logf(m) * ln2);
}
__device__ __forceinline__ float feature_log(float v) {
    int ex = 0;
    const float m = frexpf(v, &ex);  // m = v for 0, inf and NaN, with ex = 0
    return mantissa_log(m, ex);
}
__device__ __forceinline__ double feature_log(double v) {
    int ex = 0;
    const double m = frexp(v, &ex);  // v = m 2^ex, 0.5 <= |m| < 1 (m = v for 0, inf and NaN, with ex = 0)
    return double(mantissa_log(float(m), ex));
}

// Same-value stores by two lanes of one wave to one LDS address (the 20 x 10 kernel's shadow lanes) are harmless on the
// GPU; the emulator's ThreadSanitizer build is told so, everywhere else the macros are empty.
#ifndef AUD_BENIGN_RACE_BEGIN
#define AUD_BENIGN_RACE_BEGIN()
#define AUD_BENIGN_RACE_END()
#endif

// Orders the LDS traffic of ONE wave: stores issued before it are visible to loads issued after it by any lane
// of the same wave.  The hardware executes a wave's LDS instructions in order, so this emits no instruction; it
// stops the compiler from moving a lane's loads above other lanes' stores (the lanes of a wave exchanging data
// through LDS without a workgroup barrier -- what the wave-autonomous kernels do).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// ---- tile epilogue ------------------------------------------------------------------------------
// P: LDS, [F][Hp] power spectrum of the tile's F frames (Hp a multiple of 4, pad bins zeroed);
// w4: LDS copy of the chunked mel weights.  NT threads; F frames x (NT / F) filter groups.
//   * dft/dft.go:70-83: PowerSegment / LogPowerSegment [item, H, T] (optional)
//   * mel/mel.go:120-153: triangle sums (as aligned 4-bin chunks, bin order kept), + LogOff,
//     ln / LogMin, optional renorm, MelFBankSegment [item, nf, T]
// With lds_out set, mel values go to lds_out[flt * lds_pitch + lds_col0 + frame] instead of global memory
// (kernels whose tiles are narrower than a 64-byte output run collect several tiles there first).
template <typename TT, int NT, int F>
__device__ __forceinline__ void tile_spectrum_outputs(const MelspecArgs& a, const TT* P, int Hp, const aud_item& it,
                                                      int item, int t0, int tid, TT pscale = TT(1)) {
    const int T = a.T, H = a.H, N = a.N;
    const int64_t lim = it.sig_len;
    if constexpr (NT == 64) {
        // one wave: a lane keeps ONE frame (tid % F) and walks the bins k = tid / F, + 64 / F, ... -- step, liveness and the
        // output addresses are per-lane constants and every iteration is one LDS read, the logarithm and two stores (the
        // general form below re-derives frame, liveness and two 64-bit addresses for every element).  Lanes beyond
        // F * (64 / F) repeat the last group and store nothing; reads past the last bin are clamped, not skipped.
        if (a.power || a.log_power) {
            constexpr int G = 64 / F;
            const int ff = tid % F, g0 = tid / F;
            const bool has = g0 < G;
            const int g = has ? g0 : G - 1;
            const int sstep = t0 + ff;
            const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
            const bool col_on = has && sstep < T;
            const bool live = start + N <= lim;
            const TT off = TT(a.dft_log_off), lmin = TT(a.dft_log_min);
            const TT* prow = P + ff * Hp;
            size_t o = (size_t(item) * H + g) * T + (col_on ? sstep : 0);
            const size_t ostep = size_t(G) * T;
            const bool want_lp = live && a.comp_log_pow;
            const int n_it = (H + G - 1) / G;  // the same trip count for every lane
            for (int i = 0; i < n_it; ++i) {
                const int k = g + G * i;
                const TT pw = pscale * prow[k < H ? k : H - 1];
                if (col_on && k < H) {
                    if (a.power) a.power[o] = live ? float(pw) : 0.f;
                    if (a.log_power) {
                        float lp = 0.f;
                        if (want_lp) {
                            const TT vv = pw + off;
                            lp = float(vv == TT(0) ? lmin : feature_log(vv));
                        }
                        a.log_power[o] = lp;
                    }
                }
                o += ostep;
            }
        }
        return;
    }
    if (a.power || a.log_power) {
        const TT off = TT(a.dft_log_off), lmin = TT(a.dft_log_min);
        for (int w = tid; w < F * H; w += NT) {
            const int k = w / F, ff = w - k * F;
            const int sstep = t0 + ff;
            if (sstep >= T) continue;
            const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
            const bool live = start + N <= lim;
            const TT pw = pscale * P[ff * Hp + k];  // pscale: a power of two (exact)
            const size_t o = (size_t(item) * H + k) * T + sstep;
            if (a.power) a.power[o] = live ? float(pw) : 0.f;
            if (a.log_power) {
                float lp = 0.f;
                if (live && a.comp_log_pow) {
                    const TT vv = pw + off;
                    lp = float(vv == TT(0) ? lmin : feature_log(vv));
                }
                a.log_power[o] = lp;
            }
        }
    }
}

template <typename TT, int NT, int F, bool SCHED_LDS = true>
__device__ __forceinline__ void tile_epilogue(const MelspecArgs& a, const FastArgs& e, const TT* P, int Hp,
                                              const unsigned char* smem, const aud_item& it, int item, int t0,
                                              int tid, float* lds_out = nullptr, int lds_pitch = 0,
                                              int lds_col0 = 0) {
    const int T = a.T, N = a.N;
    const int64_t lim = it.sig_len;
    tile_spectrum_outputs<TT, NT, F>(a, P, Hp, it, item, t0, tid);
    const int ff = tid % F, grp = tid / F;
    const int sstep = t0 + ff;
    if (sstep >= T) return;
    const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
    const bool live = start + N <= lim;
    typedef Q4<TT> quad_t;
    const quad_t* w4 = reinterpret_cast<const quad_t*>(smem + e.w4_off);
    const quad_t* prow = reinterpret_cast<const quad_t*>(P + ff * Hp);
    const TT loff = TT(a.mel_log_off), lmin = TT(a.mel_log_min);
    float* mel_col = a.mel + (size_t(item) * a.nf * T + sstep);  // one 64-bit base, then flt * T per filter
    // the filter-group schedule: the LDS copy made by stage_mel_weights (16-bit entries), or the plan's table in
    // global memory for the kernel that has no LDS to spare for it
    typedef typename std::conditional<SCHED_LDS, unsigned short, int>::type sched_t;
    const sched_t* s_off;
    const sched_t* s_flt;
    const sched_t* s_chunk;
    if constexpr (SCHED_LDS) {
        s_off = reinterpret_cast<const sched_t*>(smem + e.sched_off);
        s_flt = s_off + e.n_groups + 1;
        s_chunk = s_flt + a.nf;
    } else {
        s_off = reinterpret_cast<const sched_t*>(e.grp_off);
        s_flt = reinterpret_cast<const sched_t*>(e.grp_flt);
        s_chunk = reinterpret_cast<const sched_t*>(e.chunk);
    }
    for (int idx = s_off[grp]; idx < int(s_off[grp + 1]); ++idx) {
        const int flt = s_flt[idx];
        float res = 0.f;
        if (live) {
            const int c0 = s_chunk[3 * flt], nc = s_chunk[3 * flt + 1], wo = s_chunk[3 * flt + 2];
            TT sum = TT(0);
#pragma unroll 4
            for (int c = 0; c < nc; ++c) {
                const quad_t pw = prow[c0 + c];
                const quad_t ww = w4[wo + c];
                sum += ww.x * pw.x;
                sum += ww.y * pw.y;
                sum += ww.z * pw.z;
                sum += ww.w * pw.w;
            }
            sum += loff;
            TT val = (sum == TT(0)) ? lmin : feature_log(sum);
            if (a.renorm) {
                val -= TT(a.renorm_min);
                if (val < TT(0)) val = TT(0);
                val *= TT(a.renorm_scale);
                if (val > TT(1)) val = TT(1);
            }
            res = float(val);
        }
        if (lds_out)
            lds_out[flt * lds_pitch + lds_col0 + ff] = res;
        else
            mel_col[size_t(flt) * T] = res;  // MelFBankSegment[item][flt][sstep]
    }
}

// ---- epilogue of the wave-autonomous kernels ---------------------------------------------------------------
// P: this wave's [FPW][Hp] power spectrum in LDS.  64 lanes = FPW frames x (64 / FPW) filter groups.  Optional
// spectrum outputs as tile_epilogue.  The mel reduction (mel/mel.go:120-153) walks the group's padded list of chunk
// steps (kernels.h FastArgs): every step is one 4-bin chunk of one filter, dot(w4 chunk, P chunk) added to a running
// sum that a step flagged `first` restarts; a step flagged `last` parks the sum in its filter's slot.  The loop has
// no data-dependent branch, so the LDS reads of four steps are in flight together (the first version walked one
// dependent chain per filter and took 27 % of a wave's life, profiles/r02c_stamps_*); logarithms and stores follow
// for all slots at once.  Products are the reference's; the additions are pairwise inside a chunk and in bin order
// across chunks (float64: far below the float32 spacing of the stored value).
template <typename TT, int FPW, int MAXS, bool COMPACT>
__device__ __forceinline__ void wave_mel_steps_impl(const MelspecArgs& a, const FastArgs& e, const TT* P, int Hp,
                                                    const unsigned char* smem, int item, int sstep, bool col_on,
                                                    bool live, int ff, int grp) {
    const int T = a.T;
    // no LDS access below sits under a lane condition (a masked frame still reads its -- valid -- row and drops the sums)
    // COMPACT (one filter group per lane, w64x16): a filter's row holds only its own chunks, steps past its end read the
    // table's shared zero chunk; otherwise every group's row has the slot's full length
    const Q4a<TT>* wrow = reinterpret_cast<const Q4a<TT>*>(smem + e.w4_off + (COMPACT ? 0 : grp * e.w_stride));
    const Q4a<TT>* prow = reinterpret_cast<const Q4a<TT>*>(P + ff * Hp);  // rows are 16-byte aligned (float64 pitch: Hp = 2 mod 4)
    const unsigned* recs = reinterpret_cast<const unsigned*>(smem + e.slots_off) + grp * e.n_slots * (COMPACT ? 2 : 1);
    const TT loff = TT(a.mel_log_off), lmin = TT(a.mel_log_min);
    float* mel_col = a.mel + (size_t(item) * a.nf * T + (col_on ? sstep : 0));
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
        if (k < e.n_slots) {  // wave-uniform
            const unsigned rec = recs[COMPACT ? 2 * k : k];
            const int ns = e.slot_steps[k];  // wave-uniform trip count: slot k is equally long in every group
            const Q4a<TT>* pp = prow + (rec & 0xFFFFu);
            TT s0 = TT(0), s1 = TT(0);
            if constexpr (COMPACT) {
                const unsigned row = recs[2 * k + 1];
                const int own = int(row >> 16);
                const unsigned char* wown = reinterpret_cast<const unsigned char*>(wrow) + 16 * (row & 0xFFFFu);  // rows start on any 16-byte piece
#pragma unroll 2
                for (int s = 0; s < ns; ++s) {
                    const Q4a<TT> pw = pp[s];
                    const Q4a<TT> ww = *(s < own ? reinterpret_cast<const Q4a<TT>*>(wown) + s : wrow);  // wrow[0] = the zero chunk
                    s0 += ww.x * pw.x;
                    s1 += ww.y * pw.y;
                    s0 += ww.z * pw.z;
                    s1 += ww.w * pw.w;
                }
            } else {
#pragma unroll 4
                for (int s = 0; s < ns; ++s) {
                    const Q4a<TT> pw = pp[s], ww = wrow[s];
                    s0 += ww.x * pw.x;
                    s1 += ww.y * pw.y;
                    s0 += ww.z * pw.z;
                    s1 += ww.w * pw.w;
                }
                wrow += ns;
            }
            const int flt = int(rec >> 16);
            const TT sum = (s0 + s1) + loff;
            TT val = (sum == TT(0)) ? lmin : feature_log(sum);
            if (a.renorm) {
                val -= TT(a.renorm_min);
                if (val < TT(0)) val = TT(0);
                val *= TT(a.renorm_scale);
                if (val > TT(1)) val = TT(1);
            }
            const float res = live ? float(val) : 0.f;
            if (col_on && flt != 0xFFFF) mel_col[size_t(flt) * T] = res;  // MelFBankSegment[item][flt][sstep]
        }
    }
}

template <typename TT, int FPW, int MAXS, bool COMPACT = false>
__device__ __forceinline__ void wave_mel_steps(const MelspecArgs& a, const FastArgs& e, const TT* P, int Hp,
                                               const unsigned char* smem, const aud_item& it, int item, int t0,
                                               int lane) {
    tile_spectrum_outputs<TT, 64, FPW>(a, P, Hp, it, item, t0, lane, TT(0.25));  // the wave kernels keep 4 x power in LDS
    // 64 is not a multiple of FPW = 6: lanes 60..63 have no filter group; they run group n_groups - 1 again and store nothing
    const int ff = lane % FPW, g0 = lane / FPW;
    const bool has = g0 < e.n_groups;
    const int grp = has ? g0 : e.n_groups - 1;
    const int sstep = t0 + ff;
    const bool col_on = has && sstep < a.T;
    const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
    const bool live = col_on && start + a.N <= int64_t(it.sig_len);
    wave_mel_steps_impl<TT, FPW, MAXS, COMPACT>(a, e, P, Hp, smem, item, sstep, col_on, live, ff, grp);  // e.n_slots <= MAXS (host)
}

// ---- mel on the matrix pipe (float32 only; an experiment the plan can switch on) -------------------
// D[16 filters x 16 frames] += A[16 filters x 4 bins] * B[4 bins x 16 frames] per v_mfma_f32_16x16x4_f32:
//   A: lane l holds W[16 b + (l & 15)][bin0 + (l >> 4)]   (pre-arranged on the host, 64 floats per K-step)
//   B: lane l holds P[frame l & 15][bin0 + (l >> 4)]       (one 4-byte LDS read)
//   D: register r of lane l = filter 16 b + 4 (l >> 4) + r, frame l & 15
// The accumulation is a k-ordered fmaf chain (bins ascending), like the reference's sequential sum; bins
// outside a triangle carry zero weights.  Filter block b runs on wave b mod 4.
typedef float aud_f32x4 __attribute__((vector_size(16)));

template <int NT, int F>
__device__ __forceinline__ void tile_mel_mfma(const MelspecArgs& a, const FastArgs& e, const float* P, int Hp,
                                              const aud_item& it, int item, int t0, int tid) {
    static_assert(F == 16, "one MFMA column per frame of the tile");
    const int wave = tid >> 6, lane = tid & 63;
    const int frame = lane & 15, kq = lane >> 4;
    const int T = a.T;
    const int sstep = t0 + frame;
    const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
    const bool live = start + a.N <= int64_t(it.sig_len);
    const float loff = float(a.mel_log_off), lmin = float(a.mel_log_min);
    for (int b = wave; b < e.n_blocks; b += NT / 64) {
        const int c0 = e.blk[3 * b], ns = e.blk[3 * b + 1], off = e.blk[3 * b + 2];
        const float* __restrict__ arow = e.atab + size_t(off) * 64 + lane;
        const float* prow = P + frame * Hp + 4 * c0 + kq;
        aud_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 4 <= ns; s += 4) {  // operands of four steps in flight, then the dependent accumulate chain
            float wa[4], pb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                wa[u] = arow[64 * (s + u)];
                pb[u] = prow[4 * (s + u)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u], pb[u], acc, 0, 0, 0);
        }
        for (; s < ns; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[64 * s], prow[4 * s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int flt = 16 * b + 4 * kq + r;
            if (flt < a.nf && sstep < T) {
                float res = 0.f;
                if (live) {
                    const float sum = acc[r] + loff;
                    float val = (sum == 0.f) ? lmin : dev_log(sum);
                    if (a.renorm) {
                        val -= float(a.renorm_min);
                        if (val < 0.f) val = 0.f;
                        val *= float(a.renorm_scale);
                        if (val > 1.f) val = 1.f;
                    }
                    res = val;
                }
                a.mel[(size_t(item) * a.nf + flt) * T + sstep] = res;
            }
        }
    }
}

// copy the chunked mel weights into LDS (visible after the next barrier)
template <typename TT, int NT>
__device__ __forceinline__ void stage_mel_weights(const FastArgs& e, unsigned char* smem, int tid) {
    typedef Q4<TT> quad_t;
    const quad_t* __restrict__ gw = static_cast<const quad_t*>(e.w4);
    quad_t* lw = reinterpret_cast<quad_t*>(smem + e.w4_off);
    for (int c = tid; c < e.n_chunks; c += NT) lw[c] = gw[c];
}

// The filter-group schedule ([groups + 1] offsets | [nf] filter ids | [nf][3] chunk info) goes into LDS as 16-bit
// entries (filter ids, chunk indices and offsets into w4 all stay far below 65536: LDS bounds them).  The epilogue
// walks it once per filter; from global memory every step of that walk is a dependent load behind an
// s_waitcnt vmcnt(0) that also waits for the previous filter's store.  Two steps so that the kernel can put the
// operand loads between them: the fetch is issued first and costs two registers, the store waits only for it
// (loads return in order), and the operand loads are neither delayed nor squeezed for registers.
struct SchedRegs {
    int v0, v1;
};
template <int NT>
__device__ __forceinline__ SchedRegs mel_schedule_fetch(const FastArgs& e, int tid) {
    SchedRegs r;
    r.v0 = tid < e.n_sched ? e.grp_off[tid] : 0;
    r.v1 = tid + NT < e.n_sched ? e.grp_off[tid + NT] : 0;
    return r;
}
template <int NT>
__device__ __forceinline__ void mel_schedule_store(const FastArgs& e, unsigned char* smem, int tid, const SchedRegs& r) {
    unsigned short* ls = reinterpret_cast<unsigned short*>(smem + e.sched_off);
    if (tid < e.n_sched) ls[tid] = static_cast<unsigned short>(r.v0);
    if (tid + NT < e.n_sched) ls[tid + NT] = static_cast<unsigned short>(r.v1);
#pragma unroll 1
    for (int c = tid + 2 * NT; c < e.n_sched; c += NT) ls[c] = static_cast<unsigned short>(e.grp_off[c]);  // nf > ~120
}

}  // namespace aud
