// Device-side building blocks shared by the frame->mel kernel families: complex helpers, small in-register DFTs (2, 3, 4,
// 5, 8, 10, 16, 20, 25 points), the sample routes (buffer-descriptor pairs with the frame's largest magnitude taken on
// the way), the per-frame power-of-two scale behind the wave kernels' float32 spectrum, and their epilogue: optional
// power / log-power outputs, slot-uniform mel reduction + log, and -- as an instantiation of its own -- the fused
// segment tail (CepstrumDct of the unrounded log-mel values, per-tile Energy sums).
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

// Every fused multiply-add of the frame->mel kernels is written out (mad), and the compiler's own contraction is off
// for these files: which of two products a contraction fuses is the compiler's choice per instantiation, and a kernel
// is instantiated per sample type and compute type -- results must not depend on which instantiation ran (int16 PCM
// and the same samples as float32 give the same bits), nor differ between the GPU and the CPU thread emulator.
#pragma clang fp contract(off)
__device__ __forceinline__ float mad(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double mad(double a, double b, double c) { return fma(a, b, c); }

template <typename TT>
__device__ __forceinline__ C2<TT> cmul(C2<TT> a, C2<TT> b) {
    return {mad(a.x, b.x, -(a.y * b.y)), mad(a.x, b.y, a.y * b.x)};
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
    const C2<TT> mm = {mad(TT(-0.5), t.x, u0.x), mad(TT(-0.5), t.y, u0.y)};
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
    const C2<TT> m1 = {mad(c2, t2.x, mad(c1, t1.x, u0.x)), mad(c2, t2.y, mad(c1, t1.y, u0.y))};
    const C2<TT> m2 = {mad(c1, t2.x, mad(c2, t1.x, u0.x)), mad(c1, t2.y, mad(c2, t1.y, u0.y))};
    const C2<TT> n1 = {mad(s2, t4.x, s1 * t3.x), mad(s2, t4.y, s1 * t3.y)};
    const C2<TT> n2 = {mad(s2, t3.x, -(s1 * t4.x)), mad(s2, t3.y, -(s1 * t4.y))};
    u0 = {u0.x + t1.x + t2.x, u0.y + t1.y + t2.y};
    u1 = cadd(m1, mul_mi(n1));
    u4 = cadd(m1, mul_pi(n1));
    u2 = cadd(m2, mul_mi(n2));
    u3 = cadd(m2, mul_pi(n2));
}

// 7-point DFT (44.1 kHz = 2^2 3^2 5^2 7^2 Hz: its 10 / 20 / 30 / 40 / 50 ms windows are 441, 882, 1323, 1764, 2205 samples): sums and
// differences of the pairs (j, 7 - j), X[k] = m_k - i n_k, X[7 - k] = m_k + i n_k
template <typename TT>
__device__ __forceinline__ void dft7(C2<TT>& u0, C2<TT>& u1, C2<TT>& u2, C2<TT>& u3, C2<TT>& u4, C2<TT>& u5, C2<TT>& u6) {
    const TT c1 = TT(0.62348980185873353053L), c2 = TT(-0.22252093395631440429L), c3 = TT(-0.90096886790241912624L);
    const TT s1 = TT(0.78183148246802980871L), s2 = TT(0.97492791218182360702L), s3 = TT(0.43388373911755812048L);
    const C2<TT> t1 = cadd(u1, u6), t2 = cadd(u2, u5), t3 = cadd(u3, u4), d1 = csub(u1, u6), d2 = csub(u2, u5), d3 = csub(u3, u4);
    const C2<TT> m1 = {mad(c3, t3.x, mad(c2, t2.x, mad(c1, t1.x, u0.x))), mad(c3, t3.y, mad(c2, t2.y, mad(c1, t1.y, u0.y)))};
    const C2<TT> m2 = {mad(c1, t3.x, mad(c3, t2.x, mad(c2, t1.x, u0.x))), mad(c1, t3.y, mad(c3, t2.y, mad(c2, t1.y, u0.y)))};
    const C2<TT> m3 = {mad(c2, t3.x, mad(c1, t2.x, mad(c3, t1.x, u0.x))), mad(c2, t3.y, mad(c1, t2.y, mad(c3, t1.y, u0.y)))};
    const C2<TT> n1 = {mad(s3, d3.x, mad(s2, d2.x, s1 * d1.x)), mad(s3, d3.y, mad(s2, d2.y, s1 * d1.y))};
    const C2<TT> n2 = {mad(-s1, d3.x, mad(-s3, d2.x, s2 * d1.x)), mad(-s1, d3.y, mad(-s3, d2.y, s2 * d1.y))};
    const C2<TT> n3 = {mad(s2, d3.x, mad(-s1, d2.x, s3 * d1.x)), mad(s2, d3.y, mad(-s1, d2.y, s3 * d1.y))};
    u0 = {u0.x + t1.x + t2.x + t3.x, u0.y + t1.y + t2.y + t3.y};
    u1 = cadd(m1, mul_mi(n1));
    u6 = cadd(m1, mul_pi(n1));
    u2 = cadd(m2, mul_mi(n2));
    u5 = cadd(m2, mul_pi(n2));
    u3 = cadd(m3, mul_mi(n3));
    u4 = cadd(m3, mul_pi(n3));
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
struct SmallDft<TT, 7> {
    static __device__ __forceinline__ void run(C2<TT> (&v)[7], const C2<TT>*, int) {
        dft7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
    }
};
// exp(-2 pi i m / 9), m = 1, 2, 4 (the inner twiddles of the 3 x 3 factorisation, m = b k1)
template <typename TT>
__device__ __forceinline__ C2<TT> w9_literal(int m) {
    return m == 1 ? C2<TT>{TT(0.76604444311897803520L), TT(-0.64278760968653932632L)}
         : m == 2 ? C2<TT>{TT(0.17364817766693034885L), TT(-0.98480775301220805937L)}
                  : C2<TT>{TT(-0.93969262078590838405L), TT(-0.34202014332566873304L)};
}
template <typename TT>
struct SmallDft<TT, 9> {  // 9 = 3 x 3, inner twiddles W9^(b k1)
    static __device__ __forceinline__ void run(C2<TT> (&v)[9], const C2<TT>*, int) {
#pragma unroll
        for (int b = 0; b < 3; ++b) dft3(v[b], v[3 + b], v[6 + b]);
        v[4] = cmul(v[4], w9_literal<TT>(1));
        v[5] = cmul(v[5], w9_literal<TT>(2));
        v[7] = cmul(v[7], w9_literal<TT>(2));
        v[8] = cmul(v[8], w9_literal<TT>(4));
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1) dft3(v[3 * k1], v[3 * k1 + 1], v[3 * k1 + 2]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i + 1; j < 3; ++j) {
                const C2<TT> t = v[3 * i + j];
                v[3 * i + j] = v[3 * j + i];
                v[3 * j + i] = t;
            }
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
// 16 = 4 x 4.  First layer: 4-point DFTs over a of the inputs n = 4 a + b (result k1 at v[4 k1 + b]); then the inner twiddles
// W16^(b k1); second layer: 4-point DFTs over b (result k2 at v[4 k1 + k2], frequency k = k1 + 4 k2); then the transposition
// that puts frequency k at v[k].  The layers are separate functions because the fixed-geometry chirp kernel
// (melspec_chirp.hip) prunes them: a first layer whose inputs n >= 8 are zero, a second layer of which only k < 8 is used.
template <typename TT>
__device__ __forceinline__ void dft16_twiddle(C2<TT> (&v)[16]) {
    const TT c1 = TT(0.92387953251128675613L), s1 = TT(0.38268343236508977173L);
    const TT r2 = TT(0.70710678118654752440L);
    v[5] = cmul(v[5], C2<TT>{c1, -s1});
    v[9] = C2<TT>{(v[9].x + v[9].y) * r2, (v[9].y - v[9].x) * r2};
    v[13] = cmul(v[13], C2<TT>{s1, -c1});
    v[6] = C2<TT>{(v[6].x + v[6].y) * r2, (v[6].y - v[6].x) * r2};
    v[10] = mul_mi(v[10]);
    v[14] = C2<TT>{(v[14].y - v[14].x) * r2, -(v[14].x + v[14].y) * r2};
    v[7] = cmul(v[7], C2<TT>{s1, -c1});
    v[11] = C2<TT>{(v[11].y - v[11].x) * r2, -(v[11].x + v[11].y) * r2};
    v[15] = cmul(v[15], C2<TT>{-c1, s1});
}
template <typename TT>
__device__ __forceinline__ void dft16_transpose(C2<TT> (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i + 1; j < 4; ++j) {
            const C2<TT> t = v[4 * i + j];
            v[4 * i + j] = v[4 * j + i];
            v[4 * j + i] = t;
        }
}
template <typename TT>
struct SmallDft<TT, 16> {
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

// ---- first-pass operands of the wave kernels --------------------------------------------------------------------
// A frame's N samples as N/2 packed pairs z[n] = (x[2n], x[2n+1]); a lane takes z[lane + STRIDE n1], n1 = 0..NV-1.  The
// kernels are instantiated per SAMPLE TYPE (SRC = the launch's sig_dtype), and where a tile's samples come from is
// wave-uniform (one item per wave tile), so the route is chosen on the scalar unit and no load sits under a lane condition:
//   float32 / int16 PCM (SRC = AUD_F32 / AUD_I16): buffer loads through a descriptor over the part of the ITEM's stream the
//     tile can touch -- the hardware returns 0 for any offset outside it, which is exactly the left zero pad of SndToWindow
//     (sndenv.go:461-468); frames that run off the END are masked later (sndenv.go:458-460), whatever was loaded for them.
//       pairs  contiguous stream, pairs on a 4-byte grid: ONE buffer_load_dwordx2 (int16: dword) per pair.  The hardware
//              takes 8-byte loads at 4-byte alignment, so float32 frames that start on an odd sample (odd step lengths: 441
//              at 44.1 kHz) need no second flavour;
//       elems  otherwise (one channel of interleaved stereo: sig_stride 2; int16 frames that start on an odd sample; a
//              left zero pad that would cut a pair in two): two element loads per pair;
//   float64 (SRC = AUD_F64, what the host-buffer entry points upload): guarded element loads.
enum { kRoutePairs = 1, kRouteElems = 2 };
template <int SRC>
struct SampleWindow {
    __amdgpu_buffer_rsrc_t rsrc;
    int route;
    int step;      // bytes between consecutive samples of the stream
    int64_t lo;    // stream position of the descriptor's first sample
};
template <>
struct SampleWindow<AUD_F64> {};
// first_start: start sample of the tile's FIRST frame; span: samples from there to the end of its last frame; every
// argument wave-uniform
template <int SRC>
__device__ __forceinline__ SampleWindow<SRC> sample_window(const MelspecArgs& a, const aud_item& it, int64_t first_start,
                                                           int span) {
    SampleWindow<SRC> w;
    if constexpr (SRC != AUD_F64) {
        constexpr int esize = SRC == AUD_F32 ? 4 : 2;
        const int str = it.sig_stride > 1 ? it.sig_stride : 1;
        const int64_t lo = first_start > 0 ? first_start : 0;
        int64_t n = int64_t(it.sig_len) - lo;  // samples of the stream from lo on, as far as the tile can reach
        if (n > int64_t(span) + (first_start < 0 ? first_start : 0)) n = int64_t(span) + (first_start < 0 ? first_start : 0);
        const int64_t bytes = n > 0 ? ((n - 1) * str + 1) * esize : 0;
        const char* base = static_cast<const char*>(a.sig) + (int64_t(it.sig_off) + lo * str) * esize;
        // a pair load needs the pair inside one 4-byte-aligned unit and whole pairs on either side of sample 0
        const bool all_even = ((it.start0 | a.S) & 1) == 0;
        const bool pairs = str == 1 && (reinterpret_cast<uintptr_t>(base) & 3) == 0 &&
                           (SRC == AUD_F32 ? (first_start >= 0 || all_even) : (a.S & 1) == 0 && (first_start >= 0 || all_even));
        w.route = pairs ? kRoutePairs : kRouteElems;
        w.step = str * esize;
        w.lo = lo;
        w.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, bytes < (int64_t(1) << 30) ? int(bytes) : 0,
                                                   0x00020000);
    }
    return w;
}
template <int NV>
struct PairRaw {
    unsigned lo[NV], hi[NV];  // float32: the two bit patterns; int16: lo = the packed pair
};
// issue only (no wait): pos0 = the lane's first sample (frame start + 2 x lane-in-frame) as a stream position, may be negative
template <int SRC, int NV, int STRIDE>
__device__ __forceinline__ void pairs_issue(const SampleWindow<SRC>& w, int64_t pos0, PairRaw<NV>& r) {
    if constexpr (SRC != AUD_F64) {
        constexpr int esize = SRC == AUD_F32 ? 4 : 2;
        const int rel = int(pos0 - w.lo);  // a negative position is a huge unsigned offset: out of range, reads 0
        if (w.route == kRoutePairs) {
            const int off = rel * esize;
#pragma unroll
            for (int n1 = 0; n1 < NV; ++n1) {
                if constexpr (SRC == AUD_F32) {
                    const auto q = __builtin_amdgcn_raw_buffer_load_b64(w.rsrc, off + 8 * STRIDE * n1, 0, 0);
                    r.lo[n1] = q[0];
                    r.hi[n1] = q[1];
                } else {
                    r.lo[n1] = __builtin_amdgcn_raw_buffer_load_b32(w.rsrc, off + 4 * STRIDE * n1, 0, 0);
                    r.hi[n1] = 0u;  // (both routes write both words: otherwise hipcc merges the routes' last stores through
                }                   //  a pointer select and the word ends up in scratch memory)
            }
        } else {
            const int off = rel * w.step;
#pragma unroll
            for (int n1 = 0; n1 < NV; ++n1) {
                if constexpr (SRC == AUD_F32) {
                    r.lo[n1] = __builtin_amdgcn_raw_buffer_load_b32(w.rsrc, off + 2 * STRIDE * n1 * w.step, 0, 0);
                    r.hi[n1] = __builtin_amdgcn_raw_buffer_load_b32(w.rsrc, off + (2 * STRIDE * n1 + 1) * w.step, 0, 0);
                } else {
                    const unsigned lo = __builtin_amdgcn_raw_buffer_load_b16(w.rsrc, off + 2 * STRIDE * n1 * w.step, 0, 0);
                    const unsigned hi = __builtin_amdgcn_raw_buffer_load_b16(w.rsrc, off + (2 * STRIDE * n1 + 1) * w.step, 0, 0);
                    r.lo[n1] = (lo & 0xFFFFu) | (hi << 16);
                    r.hi[n1] = 0u;
                }
            }
        }
    }
}
// max(m, |x|, |y|) as ONE instruction (v_max3_f32 with the abs source modifiers; a NaN operand is ignored).  hipcc turns
// fmaxf(m, fmaxf(fabsf(x), fabsf(y))) into 3.5 instructions per pair (canonicalising self-maxes): 70 instead of 20 per
// tile of the N = 400 kernel.  (tests/emul/hip/hip_runtime.h defines AUD_EMUL_ABSMAX3 with the same semantics for the CPU
// thread emulator, which cannot assemble it.)
#ifndef AUD_EMUL_ABSMAX3
__device__ __forceinline__ float absmax3(float m, float x, float y) {
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(x), "v"(y));
    return r;
}
#endif

// convert what pairs_issue requested -- or, for float64 samples, load them here.  `amax` receives the largest sample
// magnitude the lane holds (the frame's scale comes from it, frame_scale below).
template <typename TT, int SRC, int NV, int STRIDE>
__device__ __forceinline__ void pairs_take(const MelspecArgs& a, const aud_item& it, int64_t pos0, bool frame_on,
                                           const PairRaw<NV>& r, C2<TT> (&v)[NV], TT& amax) {
    if constexpr (SRC == AUD_F32) {
        float m = 0.f;
#pragma unroll
        for (int n1 = 0; n1 < NV; ++n1) {
            const float x = __uint_as_float(r.lo[n1]), y = __uint_as_float(r.hi[n1]);
            v[n1] = C2<TT>{TT(x), TT(y)};
            m = absmax3(m, x, y);
        }
        amax = TT(m);
    } else if constexpr (SRC == AUD_I16) {
        int m = 0;
#pragma unroll
        for (int n1 = 0; n1 < NV; ++n1) {
            const int x = int(int16_t(r.lo[n1] & 0xFFFFu)), y = int(int16_t(r.lo[n1] >> 16));
            v[n1].x = pcm16_to<TT>(x);
            v[n1].y = pcm16_to<TT>(y);
            m = max(m, max(x < 0 ? -x : x, y < 0 ? -y : y));
        }
        amax = pcm16_to<TT>(m);
    } else {
        const int64_t lim = it.sig_len;
        const int64_t str = it.sig_stride > 1 ? it.sig_stride : 1;
        const double* __restrict__ sig = static_cast<const double*>(a.sig) + it.sig_off;
        TT m = TT(0);
#pragma unroll
        for (int n1 = 0; n1 < NV; ++n1) {
            const int64_t p = pos0 + 2 * STRIDE * n1;
            v[n1].x = (frame_on && p >= 0 && p < lim) ? TT(sig[p * str]) : TT(0);
            v[n1].y = (frame_on && p + 1 >= 0 && p + 1 < lim) ? TT(sig[(p + 1) * str]) : TT(0);
            const TT ax = v[n1].x < TT(0) ? -v[n1].x : v[n1].x, ay = v[n1].y < TT(0) ? -v[n1].y : v[n1].y;
            m = ax > m ? ax : m;  // (a NaN sample never wins: the frame's outputs are NaN whatever the scale)
            m = ay > m ? ay : m;
        }
        amax = m;
    }
}

__device__ __forceinline__ float dev_log(float v) { return logf(v); }
__device__ __forceinline__ double dev_log(double v) { return log(v); }

// ln of a band power whose result is stored as float32 (mel.go:133-139, dft.go:76-82): v = m 2^ex with 0.5 <= m < 1
// (frexp; exact in either type), ln v = ex ln 2 + log2(m) ln 2 with log2 from the hardware's v_log_f32 (1 ulp) and one
// fused multiply-add -- 8 instructions instead of logf's 25 or log()'s ~100; no range or denormal fix-ups because m is
// always in [0.5, 1).  Error: ABSOLUTE, about 1e-7 (the float32 rounding of m, of log2(m) ln 2 in [-0.7, 0] and of the
// result) -- below the float32 spacing of the stored value once |ln v| >= 1, but a relative error of up to ~1e-7 / |ln v|
// for v close to 1 (where ex ln 2 and log2(m) ln 2 cancel): inside the 1e-5 max(1, |ref|) criterion everywhere.
// Zero, negative and NaN inputs behave as in log().
__device__ __forceinline__ float mantissa_log(float m, int ex) {
    const float ln2 = 0.693147180559945309417f;
    return fmaf(float(ex), ln2, __builtin_amdgcn_logf(m) * ln2);
}
// the same logarithm before its float32 rounding (what a float64 consumer of the value reads): absolute error ~4e-8
__device__ __forceinline__ double mantissa_log_wide(float m, int ex) {
    const double ln2 = 0.693147180559945309417;
    return mad(double(ex), ln2, double(__builtin_amdgcn_logf(m)) * ln2);
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


// Largest value of an int over the wave (every lane gets it, as a wave-uniform scalar) and over each aligned 16-lane row,
// by data-parallel-primitive moves: no LDS traffic, six / four vector instructions.  max() is idempotent, so mirrored
// lanes meeting twice does no harm.
__device__ __forceinline__ int row16_max_i32(int x) {
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));  // row_half_mirror
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));  // row_mirror: all 16 lanes of a row agree
    return x;
}
__device__ __forceinline__ int wave_max_i32(int x) {
    x = row16_max_i32(x);
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1 and 3
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(x, 63);
}

// ---- a frame's power-of-two scale ------------------------------------------------------------------------------------
// The wave kernels park the power spectrum in LDS as FLOAT32 for both compute types (half the epilogue's LDS traffic, its
// multiply-adds at the float32 rate).  float64 plans must hold for every input a double can carry, so each frame's
// spectrum is first divided by 2^sc, sc = 2 x the binary exponent of the frame's largest sample magnitude: with
// 2^(e-1) <= max |x| < 2^e, 4 |X_k|^2 <= 4 N^2 2^(2e) and (Parseval) the largest of them >= 2^(2e), so the scaled peak lies in [1, 2^24] for
// N <= 2048 and every bin down to 2^-126 of it (-370 dB; a float64 FFT's own floor is -320 dB) stays a NORMAL float32:
// relative error 2^-24 per stored bin, no dependence on the input's overall level.  All terms of a mel sum are
// non-negative, so the sum inherits that bound; exact zeros stay exact (LogMin rule, mel.go:135-137).  The scale goes
// back in after the sum: as an addition to the logarithm's exponent (LogOff == 0) or by ldexp in float64.
constexpr int kNoSignal = -(1 << 20);  // exponent of an all-zero lane
template <typename TT>
__device__ __forceinline__ int amax_exponent(TT amax) {
    int ex = 0;
    if constexpr (sizeof(TT) == 4) (void)frexpf(amax, &ex);
    else (void)frexp(amax, &ex);
    return amax > TT(0) ? ex : kNoSignal;  // (inf / NaN: exponent 0 -- the frame's outputs are inf / NaN anyway)
}
// the lanes of one frame agree on the frame's exponent through an LDS word: `slot` points at the frame's word (the same
// for all its lanes).  float32 plans keep their spectrum unscaled (sc = 0).
// the scale of the frame whose word `slot` is (the epilogue's lanes own other frames than the FFT's lanes)
// Frames at ordinary levels -- largest sample between 2^-20 and 2^30, i.e. scaled or not the spectrum's peak lies within
// [2^-40, 2^86] and every bin down to 2^-86 of it is a normal float32 (a float64 FFT's own floor is 2^-100 of the peak) --
// keep scale 0: a wave whose frames all do skips the ldexp of every bin (split_pair<TT, false>).
__device__ __forceinline__ int scale_of_exponent(int ex) { return (ex == kNoSignal || (ex >= -20 && ex <= 30)) ? 0 : 2 * ex; }
__device__ __forceinline__ int frame_scale_of(const int* slot) { return scale_of_exponent(*slot); }
template <typename TT>
__device__ __forceinline__ int frame_scale(int* slot, TT amax) {
    if constexpr (sizeof(TT) == 4) return 0;
    AUD_BENIGN_RACE_BEGIN();
    *slot = kNoSignal;  // every lane of the frame stores the same word
    AUD_BENIGN_RACE_END();
    wave_lds_fence();
    atomicMax(slot, amax_exponent<TT>(amax));  // ds_max_i32, no return
    wave_lds_fence();
    return frame_scale_of(slot);
}
// one bin of 4 x power into the float32 spectrum
template <bool SCALED = true>
__device__ __forceinline__ float scaled_power(double p4, int sc) { return SCALED ? float(ldexp(p4, -sc)) : float(p4); }
template <bool SCALED = true>
__device__ __forceinline__ float scaled_power(float p4, int) { return p4; }

// ---- epilogue of the wave-autonomous kernels ---------------------------------------------------------------------
// P: this wave's [FPW][Hp] spectrum in LDS, float32, FOUR times the power divided by 2^sc of its frame (frame_scale; `sc`
// = the scale of the frame THIS lane reduces, lane % FPW; the 1/4 lives in the mel weights).  64 lanes = FPW frames x (64 / FPW) filter groups.
//   * dft/dft.go:70-83: PowerSegment / LogPowerSegment [item, H, T] (optional outputs)
//   * mel/mel.go:120-153: triangle sums over aligned 4-bin chunks, + LogOff, ln / LogMin, optional renorm,
//     MelFBankSegment [item, nf, T]
// The mel reduction walks slot-uniform chunk steps (kernels.h WaveArgs): slot k takes the same number of steps in every
// group, so the loop bounds are scalar and a step is two 16-byte LDS reads and four multiply-adds into four running sums
// (one per position in the chunk: at most 2 x the slot's steps additions each, pairwise at the end).  No LDS access sits
// under a lane condition: a masked frame still reads its -- valid -- row and drops the sums.
// Sum of a value over an ALIGNED block of W = 4 or 8 lanes by data-parallel-primitive moves (no LDS traffic); every lane of
// the block gets it.  Order ((0+1)+(2+3)) + ((4+5)+(6+7)): the same on every run.
template <int W>
__device__ __forceinline__ float block_sum(float v) {
    static_assert(W == 4 || W == 8, "aligned lane block");
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    if constexpr (W == 8)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
    return v;
}
__device__ __forceinline__ double dpp_move(double v, int ctrl_b1_4e_141) {
    const long long bits = __builtin_bit_cast(long long, v);
    int lo = int(bits), hi = int(bits >> 32);
    if (ctrl_b1_4e_141 == 0) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
    } else if (ctrl_b1_4e_141 == 1) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false);
    }
    return __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
template <int W>
__device__ __forceinline__ double block_sum(double v) {
    static_assert(W == 4 || W == 8, "aligned lane block");
    v += dpp_move(v, 0);
    v += dpp_move(v, 1);
    if constexpr (W == 8) v += dpp_move(v, 2);
    return v;
}

// ---- optional spectrum outputs of the wave kernels (dft/dft.go:70-83: PowerSegment / LogPowerSegment [item, H, T]) ----
// Bins [k_lo, k_hi) of the wave's FPW frames.  A lane keeps ONE frame and walks the bins of its group: lane = group * W +
// frame with W lanes per group -- W = FPW (every lane busy; FPW = 6: ten groups, lanes 60..63 repeat the last one and
// store nothing) or, for the bins the fused segment tail sums over the frames, the next power of two (W = 8 for FPW = 6:
// eight groups, two idle lanes each), so that the frames of a bin sit in one aligned lane block and block_sum adds them
// without LDS traffic.  Step, liveness and the output addresses are per-lane constants; every iteration is one LDS read,
// the logarithm and two stores; reads past the last bin are clamped, not skipped.
// MODE (wave-uniform, chosen once per tile): 0 = LogOffSet == 0: float32 throughout (p4 2^(sc-2) rounds once either way,
// and its logarithm is the mantissa's plus the shifted exponent); 1 = an ordinary positive LogOffSet (the reference's
// default is 1.0, dft.go:37) and every frame's scale small enough for the sum to be a normal float32: one float64
// multiply-add, then the float32 logarithm; 2 = anything else: float64 ldexp / add / frexp as the definition reads.
// esum (fused segment tail, k_hi <= T): Energy[s] sums LogPowerSegment(s, f) over the steps f -- bin s < T of every frame
// (sndenv.go:360-366), a float64 tensor in the reference: the same logarithm BEFORE its float32 rounding,
// summed over the tile's frames into this tile's row of energy_part.
template <typename TT, int FPW, int W, int MODE>
__device__ __forceinline__ void wave_spectrum_range(const MelspecArgs& a, const float* P, int Hp, const int* exps, int sc1,
                                                    const aud_item& it, int item, int t0, int lane, int k_lo, int k_hi,
                                                    TT* esum) {
    constexpr int G = 64 / W;
    const int T = a.T, H = a.H;
    const int f0 = lane % W, g0 = lane / W;
    const bool has = g0 < G && f0 < FPW;
    const int ff = f0 < FPW ? f0 : FPW - 1, g = g0 < G ? g0 : G - 1;
    // the scale of THIS lane's frame: the frames' words (exps), or the one frame's scale (sc1: one frame per wave)
    const int sc = sizeof(TT) == 8 ? (exps ? frame_scale_of(exps + ff) : sc1) : 0;
    const int sstep = t0 + ff;
    const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
    const bool col_on = has && sstep < T;
    const bool live = start + a.N <= int64_t(it.sig_len);
    const float* prow = P + ff * Hp;
    size_t o = (size_t(item) * H + k_lo + g) * T + (col_on ? sstep : 0);
    const size_t ostep = size_t(G) * T;
    const bool want_lp = live && a.comp_log_pow;
    const int n_it = (k_hi - k_lo + G - 1) / G;  // the same trip count for every lane
    const double scale_d = MODE == 1 ? ldexp(1.0, sc - 2) : 1.0;
    for (int i = 0; i < n_it; ++i) {
        const int k = k_lo + g + G * i;
        const float p4 = prow[k < H ? k : H - 1];
        // the logarithm's argument as mantissa and exponent (`none`: the exact-zero case, LogMin)
        float pw, m = 1.f;
        int ex = 0;
        bool none = false;
        if constexpr (sizeof(TT) == 8 && MODE == 0) {
            pw = ldexpf(p4, sc - 2);
            m = frexpf(p4, &ex);
            ex += sc - 2;
            none = p4 == 0.f;
        } else if constexpr (sizeof(TT) == 8 && MODE == 1) {
            pw = ldexpf(p4, sc - 2);
            m = frexpf(float(mad(double(p4), scale_d, a.dft_log_off)), &ex);  // > 0: never the LogMin case
        } else if constexpr (sizeof(TT) == 8) {
            const double pd = ldexp(double(p4), sc - 2);
            pw = float(pd);
            const double vv = pd + a.dft_log_off;
            m = float(frexp(vv, &ex));
            none = vv == 0.0;
        } else {
            pw = 0.25f * p4;
            const float vv = pw + float(a.dft_log_off);
            m = frexpf(vv, &ex);
            none = vv == 0.f;
        }
        const float lp = !want_lp ? 0.f : none ? float(a.dft_log_min) : mantissa_log(m, ex);
        if (col_on && k < k_hi) {
            if (a.power) a.power[o] = live ? pw : 0.f;
            if (a.log_power) a.log_power[o] = lp;
        }
        o += ostep;
        if constexpr (W == 4 || W == 8) {
            if (esum) {  // wave-uniform
                TT wide;
                if constexpr (sizeof(TT) == 8) wide = none ? a.dft_log_min : mantissa_log_wide(m, ex);
                else wide = lp;
                const TT tot = block_sum<W>(want_lp && col_on ? wide : TT(0));
                if (f0 == 0 && g0 < G && k < k_hi) esum[k] = tot;
            }
        }
    }
}

// ---- the same outputs for the several-frames-per-wave kernels (FPW = 4 / 6): a lane keeps ONE BIN and HALF the tile's frames ----
// lane = 2 b + h: bin 32 i + b in pass i, frames HF h .. HF h + HF - 1 (HF = FPW / 2).  Against the one-frame-per-lane walk above:
//   * a lane's HF values of a bin are consecutive steps of one PowerSegment row -- ONE 8- / 12-byte store per tensor and pass
//     instead of HF 4-byte ones (whole-ProcessSegment kernel: 65 -> 18 store instructions per wave; the two lanes of a bin are
//     neighbours, so their halves of the 16- / 24-byte run reach the texture addresser in the same quarter-wave);
//   * one output address per pass instead of one per value, no per-value clamp / select chains: the loop body is the
//     logarithm and little else;
//   * the Energy sums of the fused segment tail (bin s < T summed over the tile's frames, sndenv.go:360-366) are the
//     lane's own HF values plus its neighbour's: one lane swap (no second lane geometry, no zero-padded 8-lane blocks).
// MODE as above.  `full` (wave-uniform): all FPW steps of the tile exist (t0 + FPW <= T) -- the vector stores; the last tile of a
// segment whose length is not a multiple of FPW takes per-value stores under its column masks.
// HF consecutive float32 values as ONE buffer store (8 or 12 bytes per lane) through a descriptor over the item's tensor:
// `voffset` in bytes; a run at or beyond the descriptor's range is dropped by the hardware (bins k >= H of the last pass need
// no branch).  A 12-byte run written as two instructions (what the compiler makes of a 4-byte-aligned struct store) costs the
// vector-memory path two passes over the same cache lines -- and that path, not vector issue, is what bounds the
// whole-ProcessSegment kernel (profiles/round5_sndenv_b4096_*: SQ_VMEM_TA_ADDR_FIFO_FULL).
template <int HF>
__device__ __forceinline__ void store_run(const __amdgpu_buffer_rsrc_t& r, int voffset, const float (&v)[HF]) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (HF == 3) {
        typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
        __builtin_amdgcn_raw_buffer_store_b96(u32x3{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2])}, r, voffset, 0, 0);
    } else {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}, r, voffset, 0, 0);
    }
#else  // (the CPU thread emulator: dword by dword, each under the descriptor's range check)
#pragma unroll
    for (int u = 0; u < HF; ++u) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[u]), r, voffset + 4 * u, 0, 0);
#endif
}
template <typename TT, int FPW, int MODE>
__device__ __forceinline__ void wave_spectrum_halves(const MelspecArgs& a, const float* P, int Hp, const int* exps, const aud_item& it,
                                                     int item, int t0, int lane, TT* esum) {
    constexpr int HF = FPW / 2;
    static_assert(FPW == 2 * HF && (HF == 2 || HF == 3), "half a tile per lane");
    const int T = a.T, H = a.H;
    const int h = lane & 1, b = lane >> 1;
    const bool full = t0 + FPW <= T;  // wave-uniform
    bool col_on[HF], live[HF];
    int sc[HF];
    float scale_f[HF];     // 2^(sc - 2) where that is a float32 (MODE 1 promises it)
    double scale_d[HF];
#pragma unroll
    for (int u = 0; u < HF; ++u) {
        const int ff = HF * h + u, sstep = t0 + ff;
        const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
        col_on[u] = sstep < T;
        live[u] = start + a.N <= int64_t(it.sig_len);
        sc[u] = sizeof(TT) == 8 ? frame_scale_of(exps + ff) : 0;
        scale_d[u] = MODE == 1 ? ldexp(1.0, sc[u] - 2) : 1.0;
        scale_f[u] = MODE == 1 ? float(scale_d[u]) : 1.f;
    }
    const float* prow = P + (HF * h) * Hp;
    // the item's [H][T] tensors behind buffer descriptors (wave-uniform: one item per tile)
    const int tensor_bytes = H * T * 4;
    const __amdgpu_buffer_rsrc_t rpw = __builtin_amdgcn_make_buffer_rsrc(a.power ? a.power + size_t(item) * H * T : nullptr, 0,
                                                                         a.power ? tensor_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rlp = __builtin_amdgcn_make_buffer_rsrc(a.log_power ? a.log_power + size_t(item) * H * T : nullptr, 0,
                                                                         a.log_power ? tensor_bytes : 0, 0x00020000);
    const int obase = (t0 + HF * h) * 4;
    const int n_e = esum ? (T + 31) >> 5 : 0;  // passes that hold a bin s < T (fused tail only)
    const int n_it = (a.power || a.log_power) ? (H + 31) >> 5 : n_e;  // (a lean segment call keeps no spectrum tensors)
    for (int i = 0; i < n_it; ++i) {
        const int k = 32 * i + b;
        const int kc = k < H ? k : H - 1;  // (reads past the last bin are clamped, not skipped)
        float pwv[HF], lpv[HF];
        TT wsum = TT(0);
#pragma unroll
        for (int u = 0; u < HF; ++u) {
            const float p4 = prow[u * Hp + kc];
            float pw, m = 1.f;
            int ex = 0;
            bool none = false;
            if constexpr (sizeof(TT) == 8 && MODE == 0) {
                pw = ldexpf(p4, sc[u] - 2);
                m = frexpf(p4, &ex);
                ex += sc[u] - 2;
                none = p4 == 0.f;
            } else if constexpr (sizeof(TT) == 8 && MODE == 1) {
                pw = p4 * scale_f[u];  // = ldexpf(p4, sc - 2): an exact power of two within float32's range
                m = frexpf(float(mad(double(p4), scale_d[u], a.dft_log_off)), &ex);  // > 0: never the LogMin case
            } else if constexpr (sizeof(TT) == 8) {
                const double pd = ldexp(double(p4), sc[u] - 2);
                pw = float(pd);
                const double vv = pd + a.dft_log_off;
                m = float(frexp(vv, &ex));
                none = vv == 0.0;
            } else {
                pw = 0.25f * p4;
                const float vv = pw + float(a.dft_log_off);
                m = frexpf(vv, &ex);
                none = vv == 0.f;
            }
            const bool want_lp = live[u] && a.comp_log_pow;
            const float lp = !want_lp ? 0.f : none ? float(a.dft_log_min) : mantissa_log(m, ex);
            pwv[u] = live[u] ? pw : 0.f;
            lpv[u] = lp;
            if (i < n_e) {  // wave-uniform
                TT wide;
                if constexpr (sizeof(TT) == 8) wide = none ? a.dft_log_min : mantissa_log_wide(m, ex);
                else wide = lp;
                wsum += (want_lp && col_on[u]) ? wide : TT(0);  // frames in step order: ((w0 + w1) + w2)
            }
        }
        {
            const int o = obase + k * T * 4;  // (k >= H: beyond the descriptor's range, dropped)
            if (full) {
                if (a.power) store_run<HF>(rpw, o, pwv);
                if (a.log_power) store_run<HF>(rlp, o, lpv);
            } else {
#pragma unroll
                for (int u = 0; u < HF; ++u) {
                    const int ou = col_on[u] ? o + 4 * u : 0x7FFFFFF0;  // a step beyond the segment: out of range
                    if (a.power) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pwv[u]), rpw, ou, 0, 0);
                    if (a.log_power) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lpv[u]), rlp, ou, 0, 0);
                }
            }
        }
        if (i < n_e) {
            // the other half's sum from the neighbouring lane (quad_perm [1,0,3,2]); first half + second half on both lanes
            TT other;
            if constexpr (sizeof(TT) == 8) other = dpp_move(wsum, 0);
            else other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, wsum), 0xB1, 0xF, 0xF, false));
            const TT tot = h == 0 ? wsum + other : other + wsum;
            if (h == 0 && k < T) esum[k] = tot;
        }
    }
}

template <typename TT, int FPW, int W>
__device__ __forceinline__ void wave_spectrum_pick(const MelspecArgs& a, const float* P, int Hp, const int* exps, int sc1,
                                                   const aud_item& it, int item, int t0, int lane, int k_lo, int k_hi,
                                                   TT* esum) {
    if constexpr (sizeof(TT) == 8) {
        const double off = a.dft_log_off;
        // the scaled peak of a frame is below 2^26 (frame_scale), so with sc < 96 the sum is below 2^123
        const int sc = exps ? frame_scale_of(exps + lane % FPW) : sc1;
        const bool small = __builtin_amdgcn_ballot_w64(sc >= 96 || sc < -900) == 0;
        if (off == 0.0) wave_spectrum_range<TT, FPW, W, 0>(a, P, Hp, exps, sc1, it, item, t0, lane, k_lo, k_hi, esum);
        else if (small && off >= 1e-30 && off <= 1e30)
            wave_spectrum_range<TT, FPW, W, 1>(a, P, Hp, exps, sc1, it, item, t0, lane, k_lo, k_hi, esum);
        else wave_spectrum_range<TT, FPW, W, 2>(a, P, Hp, exps, sc1, it, item, t0, lane, k_lo, k_hi, esum);
    } else {
        wave_spectrum_range<TT, FPW, W, 2>(a, P, Hp, exps, sc1, it, item, t0, lane, k_lo, k_hi, esum);
    }
}
template <typename TT, int FPW>
__device__ __forceinline__ void wave_spectrum_halves_pick(const MelspecArgs& a, const float* P, int Hp, const int* exps,
                                                          const aud_item& it, int item, int t0, int lane, TT* esum) {
    if constexpr (sizeof(TT) == 8) {
        const double off = a.dft_log_off;
        // MODE 1 multiplies by 2^(sc - 2) in float32 and adds in float64: every frame's scale must keep both normal
        const int sc = frame_scale_of(exps + lane % FPW);
        const bool small = __builtin_amdgcn_ballot_w64(sc >= 96 || sc < -100) == 0;
        if (off == 0.0) wave_spectrum_halves<TT, FPW, 0>(a, P, Hp, exps, it, item, t0, lane, esum);
        else if (small && off >= 1e-30 && off <= 1e30) wave_spectrum_halves<TT, FPW, 1>(a, P, Hp, exps, it, item, t0, lane, esum);
        else wave_spectrum_halves<TT, FPW, 2>(a, P, Hp, exps, it, item, t0, lane, esum);
    } else {
        wave_spectrum_halves<TT, FPW, 2>(a, P, Hp, exps, it, item, t0, lane, esum);
    }
}

template <typename TT, int FPW, bool FUSE>
__device__ __forceinline__ void wave_spectrum_outputs(const MelspecArgs& a, const float* P, int Hp, const int* exps, int sc1,
                                                      const aud_item& it, int item, int t0, int lane) {
    if (!a.power && !a.log_power && !(FUSE && a.energy_part)) return;
    if constexpr (FPW == 4 || FPW == 6) {
        TT* esum = nullptr;
        if constexpr (FUSE) {
            if (a.energy_part) {  // fused segment tail (T <= H: plan-time check): this tile's row of per-bin Energy sums
                const int tiles = (a.T + FPW - 1) / FPW;
                esum = static_cast<TT*>(a.energy_part) + (size_t(item) * tiles + t0 / FPW) * a.T;
            }
        }
        wave_spectrum_halves_pick<TT, FPW>(a, P, Hp, exps, it, item, t0, lane, esum);
    } else {
        wave_spectrum_pick<TT, FPW, FPW>(a, P, Hp, exps, sc1, it, item, t0, lane, 0, a.H, static_cast<TT*>(nullptr));
    }
}

// `sc` = the scale of the frame THIS lane reduces (lane % FPW); `exps` = the frames' scale words (null: one frame per wave).
// FUSE: the instantiation that also carries the segment tail (MelspecArgs::mfcc_acc / energy_part); the kernels branch to
// it once per wave (wave_mel_epilogue_pick), so the plain path pays nothing for it.
// TOLDS: the workgroup-per-item kernel also keeps every mel value in the item's LDS matrix mel_lds [nf][T] (what the fused
// agabor.Convolve reads: a NaN value is stored as 0.5 there, gabor.go:278-280; the tensor in memory keeps the NaN).
template <typename TT, int FPW, int MAXS, bool COMPACT = false, bool FUSE = false, bool TOLDS = false>
__device__ __forceinline__ void wave_mel_epilogue(const MelspecArgs& a, const WaveArgs& e, const float* P, int Hp,
                                                  const unsigned char* smem, int sc, const aud_item& it,
                                                  int item, int t0, int lane, const int* exps = nullptr,
                                                  float* mel_lds = nullptr, float* stash = nullptr, int stash_i = 0) {
    wave_spectrum_outputs<TT, FPW, FUSE>(a, P, Hp, exps, sc, it, item, t0, lane);
    // 64 is not a multiple of FPW = 6: lanes 60..63 have no filter group; they run group n_groups - 1 again and store nothing
    const int T = a.T;
    const int ff = lane % FPW, g0 = lane / FPW;
    const bool has = g0 < e.n_groups;
    const int grp = has ? g0 : e.n_groups - 1;
    const int sstep = t0 + ff;
    const bool col_on = has && sstep < T;
    const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
    const bool live = col_on && start + a.N <= int64_t(it.sig_len);
    // COMPACT (one filter group per lane, w64x16): a filter's row holds only its own chunks, steps past its end read the
    // table's shared zero chunk; otherwise every group's row has the slot's full length
    const Q4a<float>* wrow = reinterpret_cast<const Q4a<float>*>(smem + e.w4_off + (COMPACT ? 0 : grp * e.w_stride));
    const Q4a<float>* prow = reinterpret_cast<const Q4a<float>*>(P + ff * Hp);  // Hp is a multiple of 4: rows 16-byte aligned
    const unsigned* recs = reinterpret_cast<const unsigned*>(smem + e.slots_off) + grp * e.n_slots * (COMPACT ? 2 : 1);
    const bool plain = a.mel_log_off == 0.0 && !a.renorm;  // wave-uniform: the reference's defaults (mel.go:80, :175)
    const float lminf = float(a.mel_log_min);
    float* mel_col = a.mel + (size_t(item) * a.nf * T + (col_on ? sstep : 0));
    // fused segment tail: this lane's share of the frame's CepstrumDct (mel.go:192-212), coefficients 1 .. kDctCoefs - 1
    // (coefficient 0 is never seen: ProcessSegment overwrites MFCC row 0 with Energy, sndenv.go:368-372)
    constexpr bool kCanFuse = FUSE && !COMPACT && (FPW == 4 || FPW == 6);
    constexpr bool fuse = kCanFuse;
    TT cc[kDctCoefs];
#pragma unroll
    for (int c = 0; c < kDctCoefs; ++c) cc[c] = TT(0);
    const TT* dct_t = reinterpret_cast<const TT*>(smem + (e.dct_off >= 0 ? e.dct_off : 0));
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
        if (k < e.n_slots) {  // wave-uniform
            const unsigned rec = recs[COMPACT ? 2 * k : k];
            const int ns = e.slot_steps[k];  // wave-uniform trip count: slot k is equally long in every group
            const Q4a<float>* pp = prow + (rec & 0xFFFFu);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            if constexpr (COMPACT) {
                const unsigned row = recs[2 * k + 1];
                const int own = int(row >> 16);
                const Q4a<float>* wown = wrow + (row & 0xFFFFu);  // rows start on any 16-byte piece
#pragma unroll 2
                for (int s = 0; s < ns; ++s) {
                    const Q4a<float> pw = pp[s];
                    const Q4a<float> ww = *(s < own ? wown + s : wrow);  // wrow[0] = the zero chunk
                    s0 = fmaf(ww.x, pw.x, s0);
                    s1 = fmaf(ww.y, pw.y, s1);
                    s2 = fmaf(ww.z, pw.z, s2);
                    s3 = fmaf(ww.w, pw.w, s3);
                }
            } else {
#pragma unroll 4
                for (int s = 0; s < ns; ++s) {
                    const Q4a<float> pw = pp[s], ww = wrow[s];
                    s0 = fmaf(ww.x, pw.x, s0);
                    s1 = fmaf(ww.y, pw.y, s1);
                    s2 = fmaf(ww.z, pw.z, s2);
                    s3 = fmaf(ww.w, pw.w, s3);
                }
                wrow += ns;
            }
            const int flt = int(rec >> 16);
            const float sum = (s0 + s1) + (s2 + s3);  // = (the reference's sum) / 2^sc
            float res;
            TT wide;  // the same value before its float32 rounding (what the reference's DCT reads is a float64)
            if (plain) {
                int ex = 0;
                const float m = frexpf(sum, &ex);  // m = sum for 0 and NaN, with ex = 0
                res = sum == 0.f ? lminf : mantissa_log(m, ex + sc);
                if constexpr (sizeof(TT) == 8) wide = sum == 0.f ? a.mel_log_min : mantissa_log_wide(m, ex + sc);
                else wide = res;
            } else {
                const double sd = ldexp(double(sum), sc) + a.mel_log_off;
                double val = sd == 0.0 ? a.mel_log_min : feature_log(sd);
                if (a.renorm) {
                    val -= a.renorm_min;
                    if (val < 0.0) val = 0.0;
                    val *= a.renorm_scale;
                    if (val > 1.0) val = 1.0;
                }
                res = float(val);
                wide = TT(val);
            }
            if constexpr (COMPACT) {
                // one frame per wave (w64x16): a frame's values would be n_filters 4-byte stores into as many cache lines; with a
                // stash the wave parks them in LDS, [slot][lane][4 frames], and writes 16-byte [filter][4 steps] pieces behind
                // its fourth frame (wave_mel_flush4)
                if (stash) stash[(k * 64 + lane) * 4 + stash_i] = live ? res : 0.f;
                else if (col_on && flt != 0xFFFF) mel_col[size_t(flt) * T] = live ? res : 0.f;
            } else {
                if (col_on && flt != 0xFFFF) mel_col[size_t(flt) * T] = live ? res : 0.f;  // MelFBankSegment[item][flt][sstep]
            }
            if constexpr (TOLDS) {
                const float kept = live ? res : 0.f;
                if (col_on && flt != 0xFFFF) mel_lds[flt * T + sstep] = kept != kept ? 0.5f : kept;
            }
            if constexpr (kCanFuse) {
                if (fuse) {
                    const bool on = live && flt != 0xFFFF;
                    const TT x = on ? wide : TT(0);  // a step the loop never reached keeps MFCC = 0
                    const TT* drow = dct_t + (on ? flt : 0) * kDctPitch;
#pragma unroll
                    for (int c = 1; c < kDctCoefs; ++c) cc[c] = mad(drow[c], x, cc[c]);
                }
            }
        }
    }
    if constexpr (kCanFuse) {
        if (fuse) {
            // a frame's coefficient = the sum over its filter groups.  The spectrum has been consumed: its LDS rows take
            // the partial sums, [coefficient][frame][group], six coefficients at a time (2.9 KB in float64); lane q of
            // the first 6 FPW then adds the G consecutive values of (coefficient q / FPW, frame q % FPW), in group order.
            constexpr int G = 64 / FPW, CH = (kDctCoefs - 1) / 2, TASKS = CH * FPW;
            static_assert(2 * CH == kDctCoefs - 1 && TASKS <= 64 && G % 2 == 0, "chunking of the DCT sums");
            TT* S = reinterpret_cast<TT*>(const_cast<float*>(P));
            TT* out = static_cast<TT*>(a.mfcc_acc) + size_t(item) * a.n_coefs * T;
            const int q = lane < TASKS ? lane : 0;
            const C2<TT>* mine = reinterpret_cast<const C2<TT>*>(S + q * G);
            wave_lds_fence();  // every lane is done with the spectrum
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                // (lanes without a group -- 60..63 of FPW = 6 -- carry zeros: they store into cells of their own behind the sums)
#pragma unroll
                for (int cl = 0; cl < CH; ++cl) S[has ? (cl * FPW + ff) * G + grp : CH * FPW * G + (lane & 3)] = cc[1 + r * CH + cl];
                wave_lds_fence();
                TT tot = TT(0);
#pragma unroll
                for (int h = 0; h < G / 2; ++h) {
                    const C2<TT> two = mine[h];
                    tot += two.x;
                    tot += two.y;
                }
                const int c = 1 + r * CH + q / FPW, s2 = t0 + q % FPW;
                if (lane < TASKS && s2 < T && c < a.n_coefs) out[size_t(c) * T + s2] = tot;
                wave_lds_fence();
            }
        }
    }
}

// The stash of a one-frame-per-wave kernel after four consecutive steps t0 .. t0 + 3 of ONE item (t0 a multiple of 4, T a
// multiple of 4, the mel tensor 16-byte aligned: the caller checks): MelFBankSegment[item][flt][t0 .. t0 + 3] as one 16-byte
// store per filter.
template <int MAXS>
__device__ __forceinline__ void wave_mel_flush4(const MelspecArgs& a, const WaveArgs& e, const unsigned char* smem, const float* stash,
                                                int item, int t0, int lane) {
    const unsigned* recs = reinterpret_cast<const unsigned*>(smem + e.slots_off) + lane * e.n_slots * 2;
    float* base = a.mel + size_t(item) * a.nf * a.T + t0;
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
        if (k < e.n_slots) {  // wave-uniform
            const int flt = int(recs[2 * k] >> 16);
            const float4 v = *reinterpret_cast<const float4*>(stash + (k * 64 + lane) * 4);
            if (flt != 0xFFFF) *reinterpret_cast<float4*>(base + size_t(flt) * a.T) = v;
        }
    }
}

template <typename TT, int FPW, int MAXS, bool TOLDS = false>
__device__ __forceinline__ void wave_mel_epilogue_pick(const MelspecArgs& a, const WaveArgs& e, const float* P, int Hp,
                                                       const unsigned char* smem, int sc, const aud_item& it,
                                                       int item, int t0, int lane, const int* exps, float* mel_lds = nullptr) {
    if constexpr (TOLDS) {
        wave_mel_epilogue<TT, FPW, MAXS, false, false, true>(a, e, P, Hp, smem, sc, it, item, t0, lane, exps, mel_lds);
    } else {
        if (a.mfcc_acc != nullptr && e.dct_off >= 0)  // wave-uniform
            wave_mel_epilogue<TT, FPW, MAXS, false, true>(a, e, P, Hp, smem, sc, it, item, t0, lane, exps);
        else wave_mel_epilogue<TT, FPW, MAXS, false, false>(a, e, P, Hp, smem, sc, it, item, t0, lane, exps);
    }
}

}  // namespace aud
