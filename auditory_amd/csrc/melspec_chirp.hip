// Window lengths up to 1152 that no smooth in-place transform serves -- N = 1103 first of all, what the reference's 25 ms windows
// are on its shipped 44.1 kHz WAVs -- as a chirp (Bluestein) convolution of the FIXED length L = 2304 = 16 x 16 x 9: a pair of
// real frames (z = x0 + i x1) through ONE length-N DFT, whatever the parity or the factors of N (capi.hip decides which plans:
// every such float64 plan above 512 samples, shorter ones where the any-N routes would cost more than this kernel's fixed price).
//
// The any-N kernel (melspec_generic.hip) runs this length with run-time stage geometry: two autosort transforms of three
// stages, every stage a load-all / barrier / store-all round trip through LDS -- six round trips and twelve barriers, its
// index arithmetic on the vector unit.  Here the geometry is the compiler's, and the two transforms are a decimation-in-
// frequency / decimation-in-time PAIR:
//
//   forward  (natural order in, digit-reversed out):  radix 16 -> twiddle | radix 16 -> twiddle | radix 9
//   second   (digit-reversed in, natural order out):                       radix 9 | twiddle -> radix 16 | twiddle -> radix 16
//
// * every butterfly writes the positions it read: no barrier between a stage's loads and its stores (one barrier per stage);
// * the forward transform's last stage and the second transform's first stage work on the SAME nine elements of a thread:
//   radix 9, x bhat (the table permuted to the digit-reversed order by the host), conjugate, radix 9 -- in registers, one LDS
//   round trip instead of two;
// * the window goes from global memory straight into the first stage's registers (a thread's eight samples per frame are
//   144 apart: coalesced), the inputs 1152 <= n < 2304 of that stage are zero by construction and its butterfly's first layer
//   is pruned accordingly; of the last stage only the outputs k < 1152 are computed;
// => five LDS round trips and seven barriers per pair of frames, and the buffer's pitch (153 elements per block of 144) makes
//    every access pattern below conflict-free under the bank rules of MI355X_MICROARCH.md (LDS section).
//
// The arithmetic is the generic route's (z . chirp, FFT_L, . bhat, conjugate, FFT_L, chirp . conj; two real frames per complex
// transform on float64 plans, each divided by the power of two of its largest sample first) -- only the order of the additions
// inside the transforms differs.  Everything behind the power spectrum is frames_epilogue.h, shared with the any-N kernel.
//
// Reference semantics: sound/sndenv.go:438-478 (window extraction, left zero pad, short-signal masking), dft/dft.go:42-85 (DFT
// of the raw window, power, log), mel/mel.go:120-153.
#include "device_common.h"
#include "frames_epilogue.h"

namespace aud {
namespace {

constexpr int kL = 2304;          // transform length
constexpr int kBlk = kL / 16;     // 144: elements per block of the outer radix-16 stages
constexpr int kPitch = kBlk + 9;  // 153: a block's pitch in the LDS buffer (pitch = 9 mod 16: see the access patterns below)
constexpr int kChirpNonFinite = 1 << 20;  // sentinel exponent of a frame that holds an Inf / NaN sample (as the any-N kernel's)

// tables of the plan (one allocation, complex<TT>): tw1[j][r] = W_2304^(j r) (j < 16, r < 144), tw2[k1][n2] = W_144^(k1 n2)
// (k1 < 16, n2 < 9), bhat_mid[i][thread] = bhat[k0 + 16 k1 + 256 i] for the block (k0, k1) the thread owns in the middle stage
constexpr int kTw1 = 0, kTw2 = 16 * kBlk, kBhat = kTw2 + 16 * 9, kTabLen = kBhat + 9 * 256;

__device__ __forceinline__ double chirp_scale2(double v, int e) { return ldexp(v, e); }

// A thread's eight samples of both frames, n = tid + 144 n0: ALL sixteen loads in flight before the first is used -- the sample
// type is a launch constant (one straight-line copy per type), and no load sits under a condition: a frame with has[f] is live
// (start + N <= sig_len) with start + N > 0, so every position clamped to [max(start, 0), start + N - 1] is a sample of the
// stream; a frame without (dead, or wholly inside the left zero pad of sndenv.go:461-468) reads its partner's positions and
// discards them.  Slots outside [0, N) of the window or in the pad take 0 -- as integers / raw floats, before the conversion.
// int16 PCM / 0x7FFF (sound.go:138) by pcm16_to_double: the correctly rounded quotient for every int16 value (device_common.h)
template <typename S>
__device__ __forceinline__ void chirp_windows(const S* __restrict__ stream, int64_t stride, const int64_t (&start)[2], const bool (&has)[2],
                                              int N, int tid, double (&x)[2][8]) {
    S raw[2][8];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int64_t from = has[f] ? start[f] : start[1 - f];  // (uniform; the caller has checked has[0] || has[1])
        const int pad = from < 0 ? int(-from) : 0;               // samples of the window inside the left pad (< N)
        const S* __restrict__ first = stream + (from + pad) * stride;  // the window's first sample that exists
        const int hi = N - 1 - pad;
#pragma unroll
        for (int n0 = 0; n0 < 8; ++n0) {
            int rel = tid + kBlk * n0 - pad;
            rel = rel < 0 ? 0 : (rel > hi ? hi : rel);
            raw[f][n0] = first[int64_t(rel) * stride];
        }
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int lo = !has[f] ? N : start[f] < 0 ? int(-start[f]) : 0;  // first sample of the window that is not pad (has[f]: -start < N)
#pragma unroll
        for (int n0 = 0; n0 < 8; ++n0) {
            const int n = tid + kBlk * n0;
            const S r = (n >= lo && n < N) ? raw[f][n0] : S(0);
            if constexpr (sizeof(S) == 2) x[f][n0] = pcm16_to_double(int(r));
            else x[f][n0] = double(r);
        }
    }
}

// radix-16 butterfly whose inputs 8 .. 15 are zero: the first layer's 4-point DFTs see (u0, u1, 0, 0)
template <typename TT>
__device__ __forceinline__ void dft16_in8(C2<TT> (&v)[16]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const C2<TT> u0 = v[b], u1 = v[4 + b];
        v[b] = cadd(u0, u1);
        v[4 + b] = cadd(u0, mul_mi(u1));
        v[8 + b] = csub(u0, u1);
        v[12 + b] = cadd(u0, mul_pi(u1));
    }
    dft16_twiddle(v);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    dft16_transpose(v);
}
// radix-16 butterfly of which only the outputs 0 .. 7 are used (left in v[0 .. 8)): frequency k = k1 + 4 k2, k2 < 2
template <typename TT>
__device__ __forceinline__ void dft16_out8(C2<TT> (&v)[16]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4(v[b], v[4 + b], v[8 + b], v[12 + b]);
    dft16_twiddle(v);
    C2<TT> lo[4], hi[4];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        const C2<TT> s0 = cadd(v[4 * k1], v[4 * k1 + 2]), d0 = csub(v[4 * k1], v[4 * k1 + 2]);
        const C2<TT> s1 = cadd(v[4 * k1 + 1], v[4 * k1 + 3]), d1 = csub(v[4 * k1 + 1], v[4 * k1 + 3]);
        lo[k1] = cadd(s0, s1);
        hi[k1] = cadd(d0, mul_mi(d1));
    }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        v[k1] = lo[k1];
        v[4 + k1] = hi[k1];
    }
}

// float64 plans only (two frames per transform: F = 2).  The float32 opt-in stays on the any-N route: its N = 1103 results sit at
// the float32 criterion's edge on either route (the logarithm's own rounding), and nothing is gained by moving them
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
void k_melspec_chirp(const MelspecArgs a) {
    using TT = double;
    using Z = C2<TT>;
    Z* buf = reinterpret_cast<Z*>(dyn_lds());                              // 16 blocks of 144 elements, pitch 153
    int* wave_exp = reinterpret_cast<int*>(buf + 16 * kPitch);             // [4 waves][2 frames]
    // The radix-16 stages keep waves 0 .. 2 of the four busy (144 butterflies), and the hardware deals a workgroup's waves to
    // the CU's four SIMDs in order: the roles are rotated by the workgroup's number, so that the light wave of the workgroups a
    // CU holds does not sit on the same SIMD (measured: -2.7 %, profiles/round6_chirp_rotation_ab.txt)
    const int tid = (int(threadIdx.x) + 64 * int(blockIdx.x & 3)) & 255;
    constexpr int F = 2;  // (= a.F: the host launches this kernel for pair plans only)
    const int M = a.M, N = a.N, H = a.H, T = a.T;
    const Z* __restrict__ chirp = static_cast<const Z*>(a.bl_chirp);
    const Z* __restrict__ tab = static_cast<const Z*>(a.bl_fix);

    const int tiles = (T + F - 1) / F;
    const int wg = int(tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap));
    const int item = wg / tiles;
    const int t0 = (wg - item * tiles) * F;
    const aud_item it = a.items[item];
    // TWO real frames ride one complex transform: z[n] = x_0[n] + i x_1[n]

    // ---- the window(s), straight into the first stage's registers (sndenv.go:455-478): thread r < 144 takes the samples
    // n = r + 144 n0, n0 < 8, of both frames
    TT xs[2][8];
    int ex[2] = {kNoSignal, kNoSignal};
    {
        const int64_t stride = it.sig_stride > 1 ? it.sig_stride : 1;
        int64_t start[2];
        bool has[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int s = t0 + f;
            start[f] = int64_t(it.start0) + int64_t(a.S) * (s - a.border);
            has[f] = s < T && start[f] + N <= int64_t(it.sig_len) && start[f] + N > 0;  // (uniform) live, and not all in the left pad
#pragma unroll
            for (int n0 = 0; n0 < 8; ++n0) xs[f][n0] = TT(0);
        }
        if (tid < kBlk && (has[0] || has[1])) {
            if (a.sig_dtype == AUD_F32) chirp_windows(static_cast<const float*>(a.sig) + it.sig_off, stride, start, has, N, tid, xs);
            else if (a.sig_dtype == AUD_F64) chirp_windows(static_cast<const double*>(a.sig) + it.sig_off, stride, start, has, N, tid, xs);
            else chirp_windows(static_cast<const int16_t*>(a.sig) + it.sig_off, stride, start, has, N, tid, xs);
        }
        // each frame is divided by 2^(exponent of its largest sample) so that both components of z are O(1): what leaks from
        // one frame into the other through rounding is then 2^-53 of the frame's OWN peak; the powers are scaled back exactly
        // and a frame of exact zeros keeps an exactly zero spectrum (LogMin rule, mel.go:135-137).  An Inf / NaN sample takes
        // its frame OUT of the pair (sentinel): its bins are NaN, its partner runs alone (dft.go:42-50: independent frames)
        {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
#pragma unroll
                for (int n0 = 0; n0 < 8; ++n0) {
                    const TT v = xs[f][n0];
                    const int e = (v - v == TT(0)) ? amax_exponent<TT>(v < TT(0) ? -v : v) : kChirpNonFinite;
                    ex[f] = e > ex[f] ? e : ex[f];
                }
                ex[f] = wave_max_i32(ex[f]);
            }
            if ((tid & 63) == 0) {
                wave_exp[2 * (tid >> 6)] = ex[0];
                wave_exp[2 * (tid >> 6) + 1] = ex[1];
            }
            __syncthreads();
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                int e = wave_exp[f];
#pragma unroll
                for (int w = 1; w < 4; ++w) e = max(e, wave_exp[2 * w + f]);
                ex[f] = e;
            }
        }
    }
    const int x0 = ex[0], x1 = ex[1];
    const int e0 = (x0 == kNoSignal || x0 == kChirpNonFinite) ? 0 : x0, e1 = (x1 == kNoSignal || x1 == kChirpNonFinite) ? 0 : x1;

    // ---- forward stage 1: radix 16 over n0 (elements r + 144 n0), output k0 times W_2304^(r k0) to block k0, position r.
    // LDS: lanes = consecutive r: conflict-free
    if (tid < kBlk) {
        const int r = tid;
        Z v[16];
#pragma unroll
        for (int n0 = 0; n0 < 8; ++n0) {
            const int n = r + kBlk * n0;
            Z z = {xs[0][n0], xs[1][n0]};
            z = Z{x0 == kChirpNonFinite ? TT(0) : chirp_scale2(z.x, -e0), x1 == kChirpNonFinite ? TT(0) : chirp_scale2(z.y, -e1)};
            const Z c = chirp[n < M ? n : 0];
            v[n0] = n < M ? cmul<TT>(z, c) : Z{TT(0), TT(0)};
        }
        dft16_in8(v);
        buf[r] = v[0];
#pragma unroll
        for (int j0 = 1; j0 < 16; j0 += 8) {
            Z w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < 16) w8[u] = tab[kTw1 + (j0 + u) * kBlk + r];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < 16) buf[(j0 + u) * kPitch + r] = cmul<TT>(v[j0 + u], w8[u]);
        }
    }
    __syncthreads();

    // ---- forward stage 2: inside block k0, radix 16 over n1 (elements 9 n1 + n2), output k1 times W_144^(n2 k1) to 9 k1 + n2.
    // Thread t = 9 k0 + n2; position 153 k0 + 9 n1 + n2 = t + 9 n1 (mod 16): consecutive lanes, consecutive banks
    if (tid < kBlk) {
        const int k0 = tid / 9, n2 = tid - 9 * k0;
        Z* blk = buf + k0 * kPitch + n2;
        Z v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = blk[9 * n1];
        SmallDft<TT, 16>::run(v, nullptr, 0);
        blk[0] = v[0];
#pragma unroll
        for (int j0 = 1; j0 < 16; j0 += 8) {
            Z w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < 16) w8[u] = tab[kTw2 + (j0 + u) * 9 + n2];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < 16) blk[9 * (j0 + u)] = cmul<TT>(v[j0 + u], w8[u]);
        }
    }
    __syncthreads();

    // ---- middle: block (k0, k1) = nine consecutive elements: radix 9 (the forward transform's last stage: X[k0 + 16 k1 + 256 k2]
    // at k2), times bhat, conjugate, radix 9 (the second transform's first stage) -- all 256 threads, in registers.
    // Thread = 16 k0 + lane16 owns k1 = (lane16 - k0) mod 16: position 153 k0 + 9 k1 + i = 9 lane16 + i (mod 16) whatever k0
    {
        const int k0 = tid >> 4, k1 = (tid - k0) & 15;
        Z* blk = buf + k0 * kPitch + 9 * k1;
        Z v[9], bh[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) bh[i] = tab[kBhat + 256 * i + tid];
#pragma unroll
        for (int i = 0; i < 9; ++i) v[i] = blk[i];
        SmallDft<TT, 9>::run(v, nullptr, 0);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const Z c = cmul<TT>(v[i], bh[i]);
            v[i] = Z{c.x, -c.y};
        }
        SmallDft<TT, 9>::run(v, nullptr, 0);
#pragma unroll
        for (int i = 0; i < 9; ++i) blk[i] = v[i];
    }
    __syncthreads();

    // ---- second transform, stage 2: inside block m0, element 9 m1 + q times W_144^(m1 q), radix 16 over m1, output q1 to 9 q1 + q
    if (tid < kBlk) {
        const int m0 = tid / 9, q = tid - 9 * m0;
        Z* blk = buf + m0 * kPitch + q;
        Z v[16];
        v[0] = blk[0];
#pragma unroll
        for (int j0 = 1; j0 < 16; j0 += 8) {
            Z w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < 16) w8[u] = tab[kTw2 + (j0 + u) * 9 + q];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < 16) v[j0 + u] = cmul<TT>(blk[9 * (j0 + u)], w8[u]);
        }
        SmallDft<TT, 16>::run(v, nullptr, 0);
#pragma unroll
        for (int q1 = 0; q1 < 16; ++q1) blk[9 * q1] = v[q1];
    }
    __syncthreads();

    // ---- second transform, stage 3: element (block m0, position q) times W_2304^(m0 q), radix 16 over m0, output q0 = index
    // q + 144 q0 of the transform: only q0 < 8 (indices < 1152 >= M) is used.  The convolution's last step rides on the store:
    // Z[k] = chirp[k] . conj(.), at (block q0, position q) -- the thread's own column
    if (tid < kBlk) {
        const int q = tid;
        Z v[16];
        v[0] = buf[q];
#pragma unroll
        for (int j0 = 1; j0 < 16; j0 += 8) {
            Z w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < 16) w8[u] = tab[kTw1 + (j0 + u) * kBlk + q];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < 16) v[j0 + u] = cmul<TT>(buf[(j0 + u) * kPitch + q], w8[u]);
        }
        dft16_out8(v);
        Z c8[8];
#pragma unroll
        for (int q0 = 0; q0 < 8; ++q0) {
            const int k = q + kBlk * q0;
            c8[q0] = chirp[k < M ? k : 0];
        }
#pragma unroll
        for (int q0 = 0; q0 < 8; ++q0) buf[q0 * kPitch + q] = cmul<TT>(c8[q0], Z{v[q0].x, -v[q0].y});
    }
    __syncthreads();

    // ---- power spectrum (dft.go:64-66) into the buffer behind block 7: P[f][k], row pitch odd.  A pair's two frames are
    // separated here: X_0[k] = (Z[k] + conj Z[M - k]) / 2, X_1[k] = (Z[k] - conj Z[M - k]) / 2i, both from ONE read of the two
    TT* P = reinterpret_cast<TT*>(buf + 8 * kPitch);
    const int Hp = H | 1;
    auto zat = [&](int k) { return buf[k + 9 * (k / kBlk)]; };
    {
        for (int k = tid; k < H; k += 256) {
            const Z A = zat(k), B = zat(k == 0 ? 0 : M - k);
            {
                const TT re = (A.x + B.x) * TT(0.5), im = (A.y - B.y) * TT(0.5);
                P[k] = x0 == kNoSignal ? TT(0) : x0 == kChirpNonFinite ? TT(__builtin_nan("")) : chirp_scale2(re * re + im * im, 2 * x0);
            }
            {
                const TT re = (A.y + B.y) * TT(0.5), im = (B.x - A.x) * TT(0.5);
                P[size_t(Hp) + k] = x1 == kNoSignal ? TT(0) : x1 == kChirpNonFinite ? TT(__builtin_nan("")) : chirp_scale2(re * re + im * im, 2 * x1);
            }
        }
    }
    __syncthreads();

    frames_epilogue<TT, false>(a, it, item, tiles, t0, P, tid);
}

}  // namespace

// which plans the fixed-geometry kernel CAN serve: float64 (pairs of real frames), windows whose chirp convolution fits L = 2304
// and the first stage's eight blocks of 144 (N <= 1152).  Whether it SHOULD is the plan's business (capi.hip: every such window
// above 512 samples without a smooth in-place route; shorter ones where the any-N route would cost more than this kernel's fixed
// ~29 us per 256 segments of 14 frames)
bool melspec_chirp_serves(int N, int compute_dtype) {
    return compute_dtype == AUD_F64 && N >= 2 && 2 * N - 1 <= kL && N <= 8 * kBlk;
}

size_t melspec_chirp_lds_bytes() { return size_t(16 * kPitch) * 16 + 32; }
int melspec_chirp_table_len() { return kTabLen; }

// the three tables from the plan's long-double ones: twl[k] = exp(-2 pi i k / 2304), bhat[k] (natural order); out: kTabLen complex values
void melspec_chirp_tables(const double* twl, const double* bhat, double* out) {
    for (int j = 0; j < 16; ++j)
        for (int r = 0; r < kBlk; ++r) {
            const int e = (j * r) % kL;
            out[2 * (kTw1 + j * kBlk + r)] = twl[2 * e];
            out[2 * (kTw1 + j * kBlk + r) + 1] = twl[2 * e + 1];
        }
    for (int k1 = 0; k1 < 16; ++k1)
        for (int n2 = 0; n2 < 9; ++n2) {
            const int e = (16 * k1 * n2) % kL;  // W_144 = W_2304^16
            out[2 * (kTw2 + k1 * 9 + n2)] = twl[2 * e];
            out[2 * (kTw2 + k1 * 9 + n2) + 1] = twl[2 * e + 1];
        }
    for (int i = 0; i < 9; ++i)
        for (int t = 0; t < 256; ++t) {
            const int k0 = t >> 4, k1 = (t - k0) & 15;
            const int k = k0 + 16 * k1 + 256 * i;
            out[2 * (kBhat + 256 * i + t)] = bhat[2 * k];
            out[2 * (kBhat + 256 * i + t) + 1] = bhat[2 * k + 1];
        }
}

// the fused tail parks F x nf log-mel values behind the F power spectra, which start behind block 7 of the buffer
bool melspec_chirp_tail_fits(int H, int nf) {
    const size_t tsz = 8, F = 2;
    return size_t(8 * kPitch) * 2 * tsz + (size_t(F) * size_t(H | 1) + size_t(F) * size_t(nf)) * tsz <= size_t(16 * kPitch) * 2 * tsz;
}

hipError_t launch_melspec_chirp(const MelspecArgs& a, hipStream_t st) {
    if (a.F != 2 || !a.bl_fix) return hipErrorInvalidValue;
    const int tiles = (a.T + 1) / 2;
    const dim3 grid(unsigned(a.n_items) * unsigned(tiles));
    hipLaunchKernelGGL(k_melspec_chirp, grid, dim3(256), melspec_chirp_lds_bytes(), st, a);
    return hipGetLastError();
}

}  // namespace aud
