// Wave-autonomous frame -> FFT -> power -> mel -> log kernels: N = 512 ("w16x16") and N = 400 ("w25x8").
//
// What the round-2 counters said about the workgroup-tile kernels (melspec_r16.hip, melspec_r25.hip): the float64
// instantiations keep a whole tile's complex transpose buffer in LDS (58-72 KB per workgroup), so a CU holds one
// or two waves per SIMD, every wave spends more than half its life parked at one of three workgroup barriers
// (SQ_WAIT_ANY 55 % of SQ_WAVE_CYCLES) and the vector ALU is busy a fifth of the time.  These kernels remove the
// cause instead of tuning around it:
//
//   * the unit of work is a WAVE, not a workgroup: 64 lanes carry 4 frames x 16 lanes (N = 512) or 8 frames x
//     8 lanes (N = 400) from the samples to the mel values.  Lanes of one wave exchange data through a
//     wave-private LDS region ordered by wave_lds_fence() -- the hardware runs a wave's LDS instructions in
//     order -- and there is NO workgroup barrier at all: the stamps of the first version (profiles/r02c_stamps_*)
//     showed the one barrier behind the staging of the mel weights costing 11 % of a wave's life (the waves of a
//     workgroup leave their load phase thousands of cycles apart).  Each wave now writes its own, identical copy
//     of the small read-only tables (mel weights, filter slots) to the shared LDS locations just before its
//     epilogue -- a benign same-value race with the other waves' reads -- so no wave ever waits for another;
//   * the wave index is made scalar (readfirstlane), so the work-item record is one scalar load and the
//     address arithmetic runs on the scalar unit; operand and twiddle loads are issued back to back before the
//     first wait;
//   * the mel reduction walks per-group filter slots ({filter, first chunk, chunks, weight offset} records, one
//     LDS read each) two filters at a time with four partial sums per filter: eight independent FMA chains
//     instead of one (the first version's epilogue was a single dependent chain and took 27 % of the wave);
//   * the transposes go through LDS one component at a time (all real parts, then all imaginary parts), which
//     halves the footprint: 9 KB (N = 512) / 16 KB (N = 400) per wave in float64, half of that in float32;
//     the power spectrum then reuses the same region.  That is 3 waves per SIMD in float64 for N = 512, 2 for
//     N = 400, 4-6 in float32;
//   * row pitches are an odd number of 16-byte slots and frame pitches a multiple of 16 slots, so the column
//     stores (one frame per 16-lane store group) and the 16-byte row loads are conflict-free;
//   * float64 plans take the final logarithm in float32 (feature_log, device_common.h): the stored value is a
//     float32 anyway.
//
// Arithmetic (DFT factorisation, twiddles, real-FFT split, chunked mel reduction and its summation order) is
// that of the workgroup-tile kernels, which stay in the library as plan option "kernel" = 2.
//
// Reference semantics: sound/sndenv.go:438-478, dft/dft.go:53-85, mel/mel.go:120-153.
#include "device_common.h"

namespace aud {
namespace {

constexpr int kWaves = 4;  // waves per workgroup (they only share the LDS copy of the mel weights)

// ================================================================================================
// N = 512: 256-point complex FFT as 16 x 16, 16 lanes per frame, 4 frames per wave
// ================================================================================================
namespace w16 {
constexpr int kFW = 4;    // frames per wave
constexpr int kM = 256;   // complex FFT length
constexpr int kN = 512;   // window length
constexpr int kH = 257;   // power bins
constexpr int kHp = 260;  // P row pitch: 4 * 65 elements (4-bin chunks stay 16/32-byte aligned)
template <typename TT>
struct Layout {
    // scalar transpose rows of 16 + pad: 20 floats = 5 slots, 18 doubles = 9 slots (odd)
    static constexpr int kRow = (sizeof(TT) == 4) ? 20 : 18;
    static constexpr int kFrame = 16 * kRow;                   // 80 / 144 slots: a multiple of 16
    static constexpr int kXch = kFW * kFrame;                  // elements
    static constexpr int kP = kFW * kHp;                       // elements
    static constexpr int kRegion = (kXch > kP ? kXch : kP) * int(sizeof(TT));  // bytes per wave
};
}  // namespace w16

template <typename TT, bool PCM16>
__global__ __launch_bounds__(64 * kWaves) void k_melspec_w16(const MelspecArgs a, const FastArgs e) {
    using L = w16::Layout<TT>;
    unsigned char* smem = dyn_lds();
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);  // scalar: item record and addresses on the SALU
    const int lane = int(threadIdx.x) & 63;
    const int f = lane >> 4;   // frame within the wave
    const int j = lane & 15;   // lane within the frame's 16-lane group
    const int T = a.T;

    const int tiles = (T + w16::kFW - 1) / w16::kFW;  // wave tiles per item
    const unsigned wg = tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap);
    const int64_t wt = int64_t(wg) * kWaves + wave;
    if (wt >= int64_t(a.n_items) * tiles) return;  // wave-uniform; no barrier anywhere below
    const int item = int(wt / tiles);
    const int t0 = int(wt - int64_t(item) * tiles) * w16::kFW;
    const aud_item it = a.items[item];
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.tw);  // W_512^k
    AUD_STAMP_DECL;
    AUD_STAMP(0);

    // ---- pass 1 operands straight from global memory: z[16 n1 + j] = (x[32 n1 + 2j], x[.. + 1]) --------------
    C2<TT> v[16];
    load_frame_pairs<TT, 16, 16, w16::kN, PCM16>(a, it, t0 + f, j, v);
    // per-lane twiddles W_256^(j k1) = W_512^(2 j k1): 15 L1/L2-resident loads behind the operands, one wait for both
    C2<TT> tw1[16];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw1[k1] = tw[2 * j * k1];
    AUD_STAMP(1);
    AUD_STAMP(2);

    TT* xw = reinterpret_cast<TT*>(smem + e.xch_off + wave * L::kRegion);  // this wave's region

    // ---- pass 1: 16-point DFT over n1, twiddle -----------------------------------------------------------
#ifdef AUD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // operands and twiddles have landed
#endif
    AUD_STAMP(3);
    SmallDft<TT, 16>::run(v, nullptr, 0);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmul(v[k1], tw1[k1]);
    AUD_STAMP(4);

    // ---- transpose through the wave's LDS region, real parts then imaginary parts ---------------------------
    // element (row k1, column n2 = j) of frame f; afterwards lane j holds row k1 = j
    TT* col = xw + f * L::kFrame + j;
    const TT* row = xw + f * L::kFrame + j * L::kRow;
    TT re[16], im[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) col[k1 * L::kRow] = v[k1].x;
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (sizeof(TT) == 4) {
            const Q4<TT> r4 = reinterpret_cast<const Q4<TT>*>(row)[q];
            re[4 * q] = r4.x; re[4 * q + 1] = r4.y; re[4 * q + 2] = r4.z; re[4 * q + 3] = r4.w;
        } else {
            const C2<TT> a2 = reinterpret_cast<const C2<TT>*>(row)[2 * q], b2 = reinterpret_cast<const C2<TT>*>(row)[2 * q + 1];
            re[4 * q] = a2.x; re[4 * q + 1] = a2.y; re[4 * q + 2] = b2.x; re[4 * q + 3] = b2.y;
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) col[k1 * L::kRow] = v[k1].y;
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (sizeof(TT) == 4) {
            const Q4<TT> r4 = reinterpret_cast<const Q4<TT>*>(row)[q];
            im[4 * q] = r4.x; im[4 * q + 1] = r4.y; im[4 * q + 2] = r4.z; im[4 * q + 3] = r4.w;
        } else {
            const C2<TT> a2 = reinterpret_cast<const C2<TT>*>(row)[2 * q], b2 = reinterpret_cast<const C2<TT>*>(row)[2 * q + 1];
            im[4 * q] = a2.x; im[4 * q + 1] = a2.y; im[4 * q + 2] = b2.x; im[4 * q + 3] = b2.y;
        }
    }
    AUD_STAMP(5);
    // the split's twiddles W_512^(j + 16 q): requested here so that the second DFT covers their latency
    C2<TT> wsp[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) wsp[q] = tw[j + 16 * q];
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2] = C2<TT>{re[n2], im[n2]};

    // ---- pass 2: row k1 = j: 16-point DFT over n2 -> Z[j + 16 k2] in v[k2] ------------------------------
    SmallDft<TT, 16>::run(v, nullptr, 0);
    wave_lds_fence();  // every row has been read: the region may take the power spectrum
    AUD_STAMP(6);
    // this wave's copy of the mel weights and filter slots: requested now, stored behind the split
    WaveTables<TT, 3> tabs;
    wave_tables_fetch<TT, 3>(e, lane, tabs);

    // ---- real-FFT split + power (as melspec_r16.hip) -----------------------------------------------------
    // For k = j + 16 q (q = 0..7) the partner Z[256 - k] sits in lane (16 - j) & 15, register 15 - q (lane 0
    // pairs with itself: register (16 - q) & 15).  X[k] = (E + T)/2, X[256-k] = conj(E - T)/2 with
    // E = Z[k] + conj Z[256-k], T = -i W_512^k (Z[k] - conj Z[256-k]).
    TT* Pw = xw;                       // [4][kHp]
    TT* P = Pw + f * w16::kHp;
    {
        const int partner = (lane & 48) | ((16 - j) & 15);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const TT up0x = v[(16 - q) & 15].x, up0y = v[(16 - q) & 15].y;
            const TT up1x = v[15 - q].x, up1y = v[15 - q].y;
            const TT mine_x = (j == 0) ? up0x : up1x, mine_y = (j == 0) ? up0y : up1y;
            C2<TT> B;
            B.x = __shfl(mine_x, partner, 64);
            B.y = __shfl(mine_y, partner, 64);
            const C2<TT> A = v[q];
            const int k = j + 16 * q;
            const C2<TT> w = wsp[q];                    // W_512^k
            const C2<TT> E = {A.x + B.x, A.y - B.y};    // A + conj(B)
            const C2<TT> D = {A.x - B.x, A.y + B.y};    // A - conj(B)
            const C2<TT> mD = {D.y, -D.x};              // -i D
            const C2<TT> Tm = cmul(mD, w);
            const TT xr = E.x + Tm.x, xi = E.y + Tm.y;  // 2 X[k]
            const TT yr = E.x - Tm.x, yi = E.y - Tm.y;  // 2 conj X[256-k]
            P[k] = TT(0.25) * (xr * xr + xi * xi);
            if (k != 0) P[w16::kM - k] = TT(0.25) * (yr * yr + yi * yi);
            else P[w16::kM] = TT(0.25) * (yr * yr + yi * yi);  // k = 0 also yields the Nyquist bin
        }
        // k = 128 (lane 0, register 8) pairs with itself: X[128] = conj(Z[128])
        if (j == 0) P[128] = v[8].x * v[8].x + v[8].y * v[8].y;
        // bins 257..259 only pad the last 4-bin chunk; their weights are zero but 0 * garbage must stay 0
        if (j >= 13) P[w16::kH + (j - 13)] = TT(0);
    }
    wave_tables_store<TT, 3>(e, smem, lane, tabs);
    wave_lds_fence();
    AUD_STAMP(7);

    // ---- optional spectrum outputs and the mel reduction: 4 frames x 16 filter groups on this wave ----------
    wave_mel_epilogue<TT, w16::kFW>(a, e, Pw, w16::kHp, smem, it, item, t0, lane);
    AUD_STAMP(8);
    AUD_STAMP_FLUSH(a, wt, lane);
}


// ================================================================================================
// N = 400: 200-point complex FFT as 25 x 8, 8 lanes per frame, 8 frames per wave
// ================================================================================================
namespace w25 {
constexpr int kFW = 8;    // frames per wave
constexpr int kM = 200;   // complex FFT length
constexpr int kN = 400;   // window length
constexpr int kH = 201;   // power bins
constexpr int kHp = 204;  // P row pitch: 4 * 51 elements
template <typename TT>
struct Layout {
    // scalar transpose rows of 8 + pad: 12 floats = 3 slots, 10 doubles = 5 slots (odd); the frame pitch is
    // = 8 (mod 16) slots, which puts the four frames a 16-lane read group touches on disjoint slots
    static constexpr int kRow = (sizeof(TT) == 4) ? 12 : 10;
    static constexpr int kFrame = (sizeof(TT) == 4) ? 352 : 272;  // 88 / 136 slots; >= 25 rows
    static constexpr int kXch = kFW * kFrame;
    static constexpr int kP = kFW * kHp;
    static constexpr int kRegion = (kXch > kP ? kXch : kP) * int(sizeof(TT));  // bytes per wave
};

// the 8 scalars of one transposed row as 16-byte LDS reads
template <typename TT>
__device__ __forceinline__ void read_row8(const TT* row, TT (&d)[8]) {
    if constexpr (sizeof(TT) == 4) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const Q4<TT> r4 = reinterpret_cast<const Q4<TT>*>(row)[q];
            d[4 * q] = r4.x; d[4 * q + 1] = r4.y; d[4 * q + 2] = r4.z; d[4 * q + 3] = r4.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const C2<TT> r2 = reinterpret_cast<const C2<TT>*>(row)[q];
            d[2 * q] = r2.x; d[2 * q + 1] = r2.y;
        }
    }
}

// One (Z[k], Z[200-k]) pair of the real-FFT split -> power bins k and 200 - k, k <= 100 (melspec_r25.hip's
// arithmetic): X[k] = (E + T)/2, X[200-k] = conj(E - T)/2, E = A + conj B, T = -i W_400^k (A - conj B).
template <typename TT>
__device__ __forceinline__ void split_pair(TT* P, const C2<TT>* __restrict__ tw, int k, C2<TT> A, C2<TT> B) {
    const C2<TT> w = tw[k];
    const C2<TT> E = {A.x + B.x, A.y - B.y};
    const C2<TT> D = {A.x - B.x, A.y + B.y};
    const C2<TT> mD = {D.y, -D.x};
    const C2<TT> Tm = cmul(mD, w);
    const TT xr = E.x + Tm.x, xi = E.y + Tm.y;
    const TT yr = E.x - Tm.x, yi = E.y - Tm.y;
    P[k] = TT(0.25) * (xr * xr + xi * xi);
    P[kM - k] = TT(0.25) * (yr * yr + yi * yi);  // k = 0 -> Nyquist bin 200; k = 100 -> the same bin, same value
}
}  // namespace w25

// second launch-bounds argument = waves per SIMD the register allocator must leave room for: without it the
// float64 instantiation is scheduled into 256 VGPRs + 30 AGPRs (one wave per SIMD); LDS admits two
template <typename TT, bool PCM16>
__global__ __launch_bounds__(64 * kWaves, 2) void k_melspec_w25(const MelspecArgs a, const FastArgs e) {
    using L = w25::Layout<TT>;
    unsigned char* smem = dyn_lds();
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);  // scalar: item record and addresses on the SALU
    const int lane = int(threadIdx.x) & 63;
    const int f = lane >> 3;  // frame within the wave
    const int j = lane & 7;   // lane within the frame's 8-lane group
    const int T = a.T;

    const int tiles = (T + w25::kFW - 1) / w25::kFW;  // wave tiles per item
    const unsigned wg = tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap);
    const int64_t wt = int64_t(wg) * kWaves + wave;
    if (wt >= int64_t(a.n_items) * tiles) return;  // wave-uniform; no barrier anywhere below
    const int item = int(wt / tiles);
    const int t0 = int(wt - int64_t(item) * tiles) * w25::kFW;
    const aud_item it = a.items[item];
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.tw);  // W_400^k
    AUD_STAMP_DECL;
    AUD_STAMP(0);

    // ---- pass A operands: z[8 n1 + j] = (x[16 n1 + 2j], x[16 n1 + 2j + 1]), n1 = 0..24 ---------------------
    C2<TT> v[25];
    load_frame_pairs<TT, 25, 8, w25::kN, PCM16>(a, it, t0 + f, j, v);
    AUD_STAMP(1);
    AUD_STAMP(2);

    TT* xw = reinterpret_cast<TT*>(smem + e.xch_off + wave * L::kRegion);  // this wave's region
#ifdef AUD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // operands have landed
#endif
    AUD_STAMP(3);

    // ---- pass A: 25-point DFT over n1, twiddle W_200^(j k1) = W_400^(2 j k1) -------------------------------
    SmallDft<TT, 25>::run(v, tw, w25::kN);
    // six twiddles at a time: left alone the scheduler requests all 24 first (96 registers in float64)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int k1 = 1 + 6 * g; k1 < 7 + 6 * g; ++k1) v[k1] = cmul(v[k1], tw[2 * j * k1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    AUD_STAMP(4);

    // ---- transpose through the wave's LDS region, real parts then imaginary parts ----------------------------
    // element (row k1, column n2 = j) of frame f.  Pass B and the real-FFT split then work on ROW PAIRS: the partner
    // of Z[k], k = k1 + 25 k2, is Z[200 - k] = element (row 25 - k1, column 7 - k2), so a lane that holds rows r and
    // 25 - r has both halves of all eight pairs in its own registers (no cross-lane traffic, no selects):
    //   slot 0: rows j + 1 and 24 - j           (lanes 0..7: row pairs 1..8)
    //   slot 1: rows 9 + j and 16 - j           (lanes 0..3: row pairs 9..12);  row 0, which pairs with itself (lane 4)
    TT* col = xw + f * L::kFrame + j;
    const TT* rows = xw + f * L::kFrame;
    const int r0 = j + 1, r0p = 24 - j;                       // slot 0
    const bool pair1 = j <= 3, self1 = j == 4;                // slot 1: a row pair, or row 0 alone
    const int r1 = pair1 ? 9 + j : 0, r1p = 16 - j;
    TT ur[4][8], ui[4][8];  // [slot 0 row, its partner, slot 1 row, its partner][column]
#pragma unroll
    for (int k1 = 0; k1 < 25; ++k1) col[k1 * L::kRow] = v[k1].x;
    wave_lds_fence();
    w25::read_row8<TT>(rows + r0 * L::kRow, ur[0]);
    w25::read_row8<TT>(rows + r0p * L::kRow, ur[1]);
    if (pair1 || self1) w25::read_row8<TT>(rows + r1 * L::kRow, ur[2]);
    if (pair1) w25::read_row8<TT>(rows + r1p * L::kRow, ur[3]);
    wave_lds_fence();
#pragma unroll
    for (int k1 = 0; k1 < 25; ++k1) col[k1 * L::kRow] = v[k1].y;
    wave_lds_fence();
    w25::read_row8<TT>(rows + r0 * L::kRow, ui[0]);
    w25::read_row8<TT>(rows + r0p * L::kRow, ui[1]);
    if (pair1 || self1) w25::read_row8<TT>(rows + r1 * L::kRow, ui[2]);
    if (pair1) w25::read_row8<TT>(rows + r1p * L::kRow, ui[3]);
    wave_lds_fence();  // every row has been read: the region may take the power spectrum
    AUD_STAMP(5);
    // this wave's copy of the mel weights and filter slots: requested now, stored behind the split
    WaveTables<TT, 3> tabs;
    wave_tables_fetch<TT, 3>(e, lane, tabs);

    // ---- pass B (8-point DFT over n2 of each row, in registers: Z[k1 + 25 k2]) + real-FFT split + power ---------
    // pairs are always evaluated from their k <= 100 side, A = Z[k], B = Z[200 - k], as melspec_r25.hip does:
    // rows (r, r' = 25 - r), r <= 12: k = r + 25 c and k = r' + 25 c for c = 0..3, partners in column 7 - c
    TT* Pw = xw;  // [8][kHp]
    TT* P = Pw + f * w25::kHp;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const bool is_pair = sl == 0 || pair1;
        const bool is_self = sl == 1 && self1;
        if (is_pair || is_self) {
            const int ra = sl == 0 ? r0 : r1, rb = sl == 0 ? r0p : r1p;
            C2<TT> za[8], zb[8];
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) za[n2] = C2<TT>{ur[2 * sl][n2], ui[2 * sl][n2]};
            SmallDft<TT, 8>::run(za, nullptr, 0);
            if (is_pair) {
#pragma unroll
                for (int n2 = 0; n2 < 8; ++n2) zb[n2] = C2<TT>{ur[2 * sl + 1][n2], ui[2 * sl + 1][n2]};
                SmallDft<TT, 8>::run(zb, nullptr, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    w25::split_pair<TT>(P, tw, ra + 25 * c, za[c], zb[7 - c]);
                    w25::split_pair<TT>(P, tw, rb + 25 * c, zb[c], za[7 - c]);
                }
            } else {  // row 0: k = 25 c pairs with column 8 - c of the same row (c = 0: DC + Nyquist; c = 4: itself)
#pragma unroll
                for (int c = 0; c <= 4; ++c) w25::split_pair<TT>(P, tw, 25 * c, za[c], za[(8 - c) & 7]);
            }
        }
    }
    AUD_STAMP(6);
    if (j < 3) P[w25::kH + j] = TT(0);  // pad bins of the last 4-bin chunk
    wave_tables_store<TT, 3>(e, smem, lane, tabs);
    wave_lds_fence();
    AUD_STAMP(7);

    // ---- optional spectrum outputs and the mel reduction: 8 frames x 8 filter groups on this wave -------------
    wave_mel_epilogue<TT, w25::kFW>(a, e, Pw, w25::kHp, smem, it, item, t0, lane);
    AUD_STAMP(8);
    AUD_STAMP_FLUSH(a, wt, lane);
}

}  // namespace

bool melspec_w16_supported(int N, int S, int compute_dtype, int n_chunks, int nf, FastArgs* out) {
    if (N != w16::kN || S < 1 || nf < 1) return false;
    const int n_groups = 64 / w16::kFW, n_sched = n_groups + 1 + 4 * nf;
    const size_t tsz = compute_dtype == AUD_F64 ? 8 : 4;
    const size_t w4 = (size_t(n_chunks) * 4 * tsz + 31) & ~size_t(31);
    const int n_slots = wave_slot_bound(nf, n_groups);               // LDS carve; the plan sets the actual n_slots
    const size_t sched = (size_t(n_groups) * n_slots * 8 + 31) & ~size_t(31);  // filter slots, 8 bytes each
    const size_t region = compute_dtype == AUD_F64 ? size_t(w16::Layout<double>::kRegion) : size_t(w16::Layout<float>::kRegion);
    const size_t total = w4 + sched + kWaves * region;
    if (total > 160 * 1024) return false;
    if (out) {
        *out = FastArgs{};
        out->w4_off = 0;
        out->sched_off = int(w4);
        out->slots_off = int(w4);
        out->xch_off = int(w4 + sched);
        out->p_off = out->xch_off;
        out->n_sched = n_sched;
        out->n_groups = n_groups;
        out->lds_bytes = unsigned(total);
        out->n_chunks = n_chunks;
        out->direct = 1;
        out->ntile = 1;
    }
    return true;
}

hipError_t melspec_w16_prepare(unsigned lds_bytes) {
    const void* fns[] = {reinterpret_cast<const void*>(&k_melspec_w16<double, true>),
                         reinterpret_cast<const void*>(&k_melspec_w16<float, true>)};
    for (const void* fn : fns) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_melspec_w16(const MelspecArgs& a, const FastArgs& e, int compute_dtype, hipStream_t st) {
    const int64_t tiles = (a.T + w16::kFW - 1) / w16::kFW;
    const int64_t waves = int64_t(a.n_items) * tiles;
    const dim3 grid(unsigned((waves + kWaves - 1) / kWaves)), blk(64 * kWaves);
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_w16<double, true>), grid, blk, e.lds_bytes, st, a, e);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_w16<float, true>), grid, blk, e.lds_bytes, st, a, e);
    return hipGetLastError();
}

bool melspec_w25_supported(int N, int S, int compute_dtype, int n_chunks, int nf, FastArgs* out) {
    if (N != w25::kN || S < 1 || nf < 1) return false;
    const int n_groups = 64 / w25::kFW, n_sched = n_groups + 1 + 4 * nf;
    const size_t tsz = compute_dtype == AUD_F64 ? 8 : 4;
    const size_t w4 = (size_t(n_chunks) * 4 * tsz + 31) & ~size_t(31);
    const int n_slots = wave_slot_bound(nf, n_groups);               // LDS carve; the plan sets the actual n_slots
    const size_t sched = (size_t(n_groups) * n_slots * 8 + 31) & ~size_t(31);  // filter slots, 8 bytes each
    const size_t region = compute_dtype == AUD_F64 ? size_t(w25::Layout<double>::kRegion) : size_t(w25::Layout<float>::kRegion);
    const size_t total = w4 + sched + kWaves * region;
    if (total > 160 * 1024) return false;
    if (out) {
        *out = FastArgs{};
        out->w4_off = 0;
        out->sched_off = int(w4);
        out->slots_off = int(w4);
        out->xch_off = int(w4 + sched);
        out->p_off = out->xch_off;
        out->n_sched = n_sched;
        out->n_groups = n_groups;
        out->lds_bytes = unsigned(total);
        out->n_chunks = n_chunks;
        out->direct = 1;
        out->ntile = 1;
    }
    return true;
}

hipError_t melspec_w25_prepare(unsigned lds_bytes) {
    const void* fns[] = {reinterpret_cast<const void*>(&k_melspec_w25<double, true>),
                         reinterpret_cast<const void*>(&k_melspec_w25<float, true>)};
    for (const void* fn : fns) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_melspec_w25(const MelspecArgs& a, const FastArgs& e, int compute_dtype, hipStream_t st) {
    const int64_t tiles = (a.T + w25::kFW - 1) / w25::kFW;
    const int64_t waves = int64_t(a.n_items) * tiles;
    const dim3 grid(unsigned((waves + kWaves - 1) / kWaves)), blk(64 * kWaves);
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_w25<double, true>), grid, blk, e.lds_bytes, st, a, e);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_melspec_w25<float, true>), grid, blk, e.lds_bytes, st, a, e);
    return hipGetLastError();
}

}  // namespace aud
