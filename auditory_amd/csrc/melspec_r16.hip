// Fast frame -> FFT -> power -> mel -> log kernel for win_samples = 512 (the "512-pt" BASELINE
// configuration): the packed-real trick turns each frame into one 256-point complex FFT,
// computed as 16 x 16 with both radix-16 passes held entirely in registers.
//
// Geometry (wave64, 256 threads = 4 waves per workgroup):
//   * one workgroup = one tile of 16 consecutive frames of one work item;
//   * 16 lanes cooperate on a frame, each lane owns 16 complex points, so a wave carries
//     4 frames and every butterfly is straight-line VALU code on registers;
//   * input, two variants (template flag DIRECT, chosen per plan, see aud_plan_set_option):
//       DIRECT  each lane pulls its 16 sample pairs straight from global memory into registers
//               (8-byte loads, 128 B contiguous per frame; the N/S = 3.2x overlap between frames
//               is served by L1/L2, HBM still sees every sample once).  No staging buffer, no
//               staging barrier, LDS = transpose buffer only => 4 workgroups per CU;
//       STAGED  the tile's contiguous sample span (15*S + N samples) is fetched once with 16-byte
//               loads into LDS and pass 1 reads it from there => 3 workgroups per CU;
//   * the single 16x16 transpose between the two passes goes through LDS in rows of
//     16 + pad complex so that both the 8-byte column writes and the 16-byte row reads are
//     bank-conflict free (row pitch = 4 * odd dwords; frame pitch = a multiple of 256 B);
//   * Z[k] / Z[256-k] pairs for the real-FFT split live in lanes j and 16-j of the same
//     16-lane group and are exchanged with wave shuffles (no LDS round trip);
//   * the power spectrum is parked in LDS (the sample span is dead by then) and the mel
//     triangles are reduced by 16 filter groups x 16 frames; the host balances the groups
//     so that all 256 threads carry about the same number of taps.
//
// Reference semantics: sound/sndenv.go:438-478, dft/dft.go:53-85, mel/mel.go:120-153.
#include "device_common.h"

namespace aud {
namespace {

// forward 16-point DFT of v[0..15], natural order in and out
template <typename TT>
__device__ __forceinline__ void dft16(C2<TT> (&v)[16]) { SmallDft<TT, 16>::run(v, nullptr, 0); }

constexpr int kF = 16;    // frames per workgroup
constexpr int kM = 256;   // complex FFT length
constexpr int kN = 512;   // window length
constexpr int kH = 257;   // power bins
constexpr int kHp = 260;  // P row pitch: 4 * 65 elements, so 4-bin (16-byte) chunks stay aligned and the
                          // 16 frames of a tile land on 16 distinct 4-bank groups

template <typename TT>
struct Layout {
    // transpose rows: 16 + pad complex; pitch in dwords must be 4 * odd
    static constexpr int kRowC = (sizeof(TT) == 4) ? 18 : 17;
    static constexpr int kFrameC = 16 * kRowC;
};

// pass-1 operands of frame (t0 + f) straight from global memory: z[16 n1 + j] = (x[32 n1 + 2j], x[.. + 1])
// (the two-tile kernel is built without the int16 route -- it would cost it a fourth wave per SIMD -- and the
// launcher sends int16 input to the one-tile kernel)
template <typename TT, bool PCM16_ROUTE>
__device__ __forceinline__ void r16_load_direct(const MelspecArgs& a, const aud_item& it, int t0, int f, int j,
                                                C2<TT> (&v)[16]) {
    load_frame_pairs<TT, 16, 16, kN, PCM16_ROUTE>(a, it, t0 + f, j, v);
}

// everything after the pass-1 operands are in registers: both DFT passes, the transpose, the split, the
// power spectrum and the tile epilogue, for the 16 frames t0 .. t0 + 15
// SCHED_LDS: the one-tile kernels, which have registers and LDS to spare (schedule in LDS, split twiddles early)
template <typename TT, bool DIRECT, bool MELMFMA, bool SCHED_LDS>
__device__ __forceinline__ void r16_tile(const MelspecArgs& a, const FastArgs& e, unsigned char* smem, C2<TT>* xch,
                                         TT* Pbase, const C2<TT>* __restrict__ tw, const aud_item& it, int item,
                                         int t0, int tid, int f, int j, C2<TT> (&v)[16]) {
    // per-lane twiddles W_256^(j*k1) = W_512^(2 j k1): 15 L1-resident loads, issued ahead of the first DFT
    C2<TT> tw1[16];
    tw1[0] = C2<TT>{TT(1), TT(0)};
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw1[k1] = tw[2 * j * k1];
    // ---- pass 1: 16-point DFT over n1 ----------------------------------------------------------
    dft16(v);
    {
        C2<TT>* col = xch + f * Layout<TT>::kFrameC + j;  // row k1, column n2 = j
        col[0] = v[0];
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) col[k1 * Layout<TT>::kRowC] = cmul(v[k1], tw1[k1]);
    }
    __syncthreads();

    // ---- pass 2: row k1 = j: 16-point DFT over n2 -> Z[j + 16 k2] in v[k2] --------------------
    {
        // row pitch and frame pitch are multiples of 16 B, so the row is read as 16-byte pieces
        const C2<TT>* row = xch + f * Layout<TT>::kFrameC + j * Layout<TT>::kRowC;
        if (sizeof(TT) == 4) {
            const C2x2<TT>* row2 = reinterpret_cast<const C2x2<TT>*>(row);
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) {
                const C2x2<TT> pr = row2[n2];
                v[2 * n2] = pr.a;
                v[2 * n2 + 1] = pr.b;
            }
        } else {
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) v[n2] = row[n2];
        }
    }
    // the split's twiddles W_512^(j + 16 q): requested here so that the second DFT covers their latency (the
    // compiler cannot lift them over the barrier by itself).  16 registers: one-tile kernels only.
    C2<TT> wsp[SCHED_LDS ? 8 : 1];
    if constexpr (SCHED_LDS) {
#pragma unroll
        for (int q = 0; q < 8; ++q) wsp[q] = tw[j + 16 * q];
    }
    if constexpr (DIRECT) __syncthreads();  // P reuses the transpose buffer: every row must have been read
    dft16(v);

    // ---- real-FFT split + power ------------------------------------------------------------
    // For k = j + 16 q (q = 0..7) the partner Z[256 - k] sits in lane (16 - j) & 15, register
    // 15 - q (lane 0 pairs with itself: register (16 - q) & 15).  X[k] = (E + T)/2,
    // X[256-k] = conj(E - T)/2 with E = Z[k] + conj Z[256-k], T = -i W_512^k (Z[k] - conj Z[256-k]).
    TT* P = Pbase + f * kHp;  // STAGED: over the dead sample span; DIRECT: over the transpose buffer
    {
        const int lane = tid & 63;
        const int partner = (lane & 48) | ((16 - j) & 15);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // what my partner needs from me (selected by value: a pointer select would pin v[] in scratch)
            const TT up0x = v[(16 - q) & 15].x, up0y = v[(16 - q) & 15].y;
            const TT up1x = v[15 - q].x, up1y = v[15 - q].y;
            const TT mine_x = (j == 0) ? up0x : up1x, mine_y = (j == 0) ? up0y : up1y;
            C2<TT> B;
            B.x = __shfl(mine_x, partner, 64);
            B.y = __shfl(mine_y, partner, 64);
            const C2<TT> A = v[q];
            const int k = j + 16 * q;
            C2<TT> w;                                   // W_512^k
            if constexpr (SCHED_LDS) w = wsp[q];
            else w = tw[k];
            const C2<TT> E = {A.x + B.x, A.y - B.y};    // A + conj(B)
            const C2<TT> D = {A.x - B.x, A.y + B.y};    // A - conj(B)
            const C2<TT> mD = {D.y, -D.x};              // -i D
            const C2<TT> Tm = cmul(mD, w);
            const TT xr = E.x + Tm.x, xi = E.y + Tm.y;  // 2 X[k]
            const TT yr = E.x - Tm.x, yi = E.y - Tm.y;  // 2 conj X[256-k]
            P[k] = TT(0.25) * (xr * xr + xi * xi);
            if (k != 0) P[kM - k] = TT(0.25) * (yr * yr + yi * yi);
            else P[kM] = TT(0.25) * (yr * yr + yi * yi);  // k = 0 also yields the Nyquist bin
        }
        // k = 128 (lane 0, register 8) pairs with itself: X[128] = conj(Z[128])
        if (j == 0) P[128] = v[8].x * v[8].x + v[8].y * v[8].y;
        // bins 257..259 only pad the last 4-bin chunk; their weights are zero but 0 * garbage must stay 0
        if (j >= 13) P[kH + (j - 13)] = TT(0);
    }
    __syncthreads();

    if constexpr (MELMFMA) {  // own instantiations: the accumulator AGPRs would cost the others a wave per SIMD
        static_assert(sizeof(TT) == 4, "float32 variant");
        tile_spectrum_outputs<TT, 256, kF>(a, Pbase, kHp, it, item, t0, tid);
        tile_mel_mfma<256, kF>(a, e, Pbase, kHp, it, item, t0, tid);
    } else {
        tile_epilogue<TT, 256, kF, SCHED_LDS>(a, e, Pbase, kHp, smem, it, item, t0, tid);
    }
}

// NTILE tiles of 16 frames per workgroup.  With NTILE = 2 (DIRECT only) the operands of the second tile are
// requested before the first tile is computed, so their memory latency hides behind a whole tile of work,
// and the grid shrinks to one residency round at the bench size.
template <typename TT, bool DIRECT, int NTILE, bool MELMFMA>
__global__ __launch_bounds__(256) void k_melspec_r16(const MelspecArgs a, const FastArgs e) {
    unsigned char* smem = dyn_lds();
    TT* sigbuf = reinterpret_cast<TT*>(smem);                   // STAGED only: [span]
    TT* Pbase = reinterpret_cast<TT*>(smem + e.p_off);          // power spectrum [16][kHp]
    C2<TT>* xch = reinterpret_cast<C2<TT>*>(smem + e.xch_off);  // [16][16][kRowC]
    const int tid = threadIdx.x;
    const int f = tid >> 4;   // frame within the tile
    const int j = tid & 15;   // lane within the frame's 16-lane group
    const int T = a.T, S = a.S;

    const int tiles = (T + kF * NTILE - 1) / (kF * NTILE);
    const int wg = int(tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap));
    const int item = wg / tiles;
    const int t0 = (wg - item * tiles) * kF * NTILE;
    const aud_item it = a.items[item];
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.tw);  // W_512^k
    const int64_t lim = it.sig_len;
    (void)sigbuf; (void)lim; (void)S;

    // the two-tile kernel keeps the schedule in global memory: with 64 operand registers in flight the LDS copy
    // costs it its fourth wave per SIMD
    constexpr bool kSchedLds = NTILE == 1;
    SchedRegs sched{0, 0};
    if constexpr (kSchedLds) sched = mel_schedule_fetch<256>(e, tid);  // issued ahead of the operand loads

    C2<TT> v[16];
    C2<TT> v2[NTILE > 1 ? 16 : 1];
    if constexpr (DIRECT) {
        r16_load_direct<TT, NTILE == 1>(a, it, t0, f, j, v);
        if constexpr (NTILE > 1) r16_load_direct<TT, NTILE == 1>(a, it, t0 + kF, f, j, v2);  // in flight during tile 1
    } else {
        // ---- stage the tile's sample span: positions g0 .. g0 + span of the item's stream ------
        const int64_t g0 = int64_t(it.start0) + int64_t(S) * (t0 - a.border);
        const int span = (kF - 1) * S + kN;
        if (a.sig_dtype == AUD_F32 && sizeof(TT) == 4 && it.sig_stride <= 1 && ((it.sig_off + g0) & 3) == 0 &&
            (reinterpret_cast<uintptr_t>(a.sig) & 15) == 0) {
            const float* __restrict__ src = static_cast<const float*>(a.sig) + it.sig_off;
            for (int c = tid; c * 4 < span; c += 256) {
                const int64_t p = g0 + 4 * c;
                float4 q4;
                if (p >= 0 && p + 3 < lim) {
                    q4 = *reinterpret_cast<const float4*>(src + p);
                } else {
                    q4.x = (p >= 0 && p < lim) ? src[p] : 0.f;
                    q4.y = (p + 1 >= 0 && p + 1 < lim) ? src[p + 1] : 0.f;
                    q4.z = (p + 2 >= 0 && p + 2 < lim) ? src[p + 2] : 0.f;
                    q4.w = (p + 3 >= 0 && p + 3 < lim) ? src[p + 3] : 0.f;
                }
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(sigbuf) + 4 * c) = q4;
            }
        } else {
            for (int c = tid; c < span; c += 256) {
                const int64_t p = g0 + c;
                sigbuf[c] = (p >= 0 && p < lim) ? load_sample<TT>(a.sig, a.sig_dtype, it.sig_off + p * (it.sig_stride > 1 ? it.sig_stride : 1)) : TT(0);
            }
        }
    }

    // the filter-group schedule and the chunked mel weights (a few KB) ride along into LDS; first used after the
    // last barrier
    if constexpr (kSchedLds) mel_schedule_store<256>(e, smem, tid, sched);
    stage_mel_weights<TT, 256>(e, smem, tid);

    if constexpr (!DIRECT) {
        __syncthreads();
        // ---- pass 1 operands from the staged span: z[16 n1 + j] ----------------------------------
        // S is even on this path (checked by the host), so every frame starts on an 8-byte
        // boundary and each point is one 8-byte LDS read
        const C2<TT>* fr = reinterpret_cast<const C2<TT>*>(sigbuf) + (f * (S >> 1) + j);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = fr[16 * n1];
    }
    r16_tile<TT, DIRECT, MELMFMA, kSchedLds>(a, e, smem, xch, Pbase, tw, it, item, t0, tid, f, j, v);
    if constexpr (NTILE > 1) {
        if (t0 + kF < T) {          // uniform: the item has a second tile for this workgroup
            __syncthreads();        // the power spectrum of tile 1 is consumed: the buffers are free again
            r16_tile<TT, DIRECT, MELMFMA, kSchedLds>(a, e, smem, xch, Pbase, tw, it, item, t0 + kF, tid, f, j, v2);
        }
    }
}


}  // namespace

bool melspec_r16_supported(int N, int S, int compute_dtype, int n_chunks, int nf, bool direct, FastArgs* out) {
    if (N != kN || S < 1 || nf < 1) return false;
    const int n_groups = 256 / kF, n_sched = n_groups + 1 + 4 * nf;
    const size_t sched = (size_t(n_sched) * 2 + 15) & ~size_t(15);  // kept as uint16 in LDS
    if (!direct && (S & 1)) return false;  // the staged variant reads 8-byte pairs from LDS
    const size_t tsz = compute_dtype == AUD_F64 ? 8 : 4;
    const size_t rowc = compute_dtype == AUD_F64 ? 17 : 18;
    const size_t xch = size_t(kF) * 16 * rowc * 2 * tsz;
    const size_t pbytes = (size_t(kF) * kHp * tsz + 31) & ~size_t(31);
    const size_t w4 = (size_t(n_chunks) * 4 * tsz + 31) & ~size_t(31);
    size_t first = 0, xch_off, p_off;
    if (direct) {
        xch_off = 0;
        p_off = 0;  // the power spectrum reuses the transpose buffer (one extra barrier)
        first = xch > pbytes ? xch : pbytes;
    } else {
        const size_t span = size_t(kF - 1) * S + kN;
        first = ((span * tsz + 31) & ~size_t(31));
        if (first < pbytes) first = pbytes;  // sample span, later the power spectrum
        p_off = 0;
        xch_off = first;
        first += xch;
    }
    const size_t total = first + w4 + sched;
    if (total > 160 * 1024) return false;
    if (out) {
        out->sched_off = int(first + w4);
        out->n_sched = n_sched;
        out->n_groups = n_groups;
        out->xch_off = int(xch_off);
        out->p_off = int(p_off);
        out->w4_off = int(first);
        out->lds_bytes = unsigned(total);
        out->n_chunks = n_chunks;
        out->direct = direct ? 1 : 0;
        if (!direct) out->ntile = 1;
    }
    return true;
}

hipError_t melspec_r16_prepare(unsigned lds_bytes) {
    // more than 64 KiB of dynamic LDS has to be requested explicitly
    const void* fns[] = {reinterpret_cast<const void*>(&k_melspec_r16<double, true, 1, false>),
                         reinterpret_cast<const void*>(&k_melspec_r16<double, true, 2, false>),
                         reinterpret_cast<const void*>(&k_melspec_r16<double, false, 1, false>),
                         reinterpret_cast<const void*>(&k_melspec_r16<float, true, 1, false>),
                         reinterpret_cast<const void*>(&k_melspec_r16<float, true, 2, false>),
                         reinterpret_cast<const void*>(&k_melspec_r16<float, false, 1, false>),
                         reinterpret_cast<const void*>(&k_melspec_r16<float, true, 1, true>),
                         reinterpret_cast<const void*>(&k_melspec_r16<float, true, 2, true>),
                         reinterpret_cast<const void*>(&k_melspec_r16<float, false, 1, true>)};
    for (const void* fn : fns) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_melspec_r16(const MelspecArgs& a, const FastArgs& e, int compute_dtype, hipStream_t st) {
    // int16 samples: the one-tile kernel has the 4-byte-per-pair route
    const int ntile = (e.direct && e.ntile == 2 && a.sig_dtype != AUD_I16) ? 2 : 1;
    const int tiles = (a.T + kF * ntile - 1) / (kF * ntile);
    const dim3 grid(unsigned(a.n_items) * unsigned(tiles)), blk(256);
    const unsigned lds = e.lds_bytes;
#define AUD_R16_LAUNCH(TT, MF)                                                                                  \
    do {                                                                                                        \
        if (e.direct && ntile == 2)                                                                             \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_r16<TT, true, 2, MF>), grid, blk, lds, st, a, e);      \
        else if (e.direct)                                                                                      \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_r16<TT, true, 1, MF>), grid, blk, lds, st, a, e);      \
        else                                                                                                    \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_r16<TT, false, 1, MF>), grid, blk, lds, st, a, e);     \
    } while (0)
    if (compute_dtype == AUD_F64)
        AUD_R16_LAUNCH(double, false);
    else if (e.mel_mfma)
        AUD_R16_LAUNCH(float, true);
    else
        AUD_R16_LAUNCH(float, false);
#undef AUD_R16_LAUNCH
    return hipGetLastError();
}

}  // namespace aud
