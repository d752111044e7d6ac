// Wave-autonomous frame -> FFT -> power -> mel -> log kernels: N = 512 ("w16x16"), N = 400 ("w20x10", the default, and
// "w25x8") and N = 2048 ("w64x16", one frame per wave).
//
// What the round-2 counters said about the workgroup-tile kernels (melspec_r16.hip, melspec_r25.hip): the float64
// instantiations keep a whole tile's complex transpose buffer in LDS (58-72 KB per workgroup), so a CU holds one
// or two waves per SIMD, every wave spends more than half its life parked at one of three workgroup barriers
// (SQ_WAIT_ANY 55 % of SQ_WAVE_CYCLES) and the vector ALU is busy a fifth of the time.  These kernels remove the
// causes instead of tuning around them (s_memtime stamps of each step of the way: profiles/r02c..e_stamps_*):
//
//   * the unit of work is a WAVE, not a workgroup: 64 lanes carry 4 frames x 16 lanes (N = 512), 6 frames x 10 lanes or
//     8 frames x 8 lanes (N = 400), or one frame (N = 2048) from the samples to the mel values.  Lanes of one wave exchange data through a
//     wave-private LDS region ordered by wave_lds_fence() -- the hardware runs a wave's LDS instructions in
//     order -- so the data path has no workgroup barrier;
//   * every read-only table (mel weight rows, the epilogue's slot records, pass and split twiddles) comes as ONE
//     blob that the workgroup copies into LDS at its very start: the blob loads are issued first, the operand
//     loads behind them, and a counted wait (the loads return in order) lets the blob be stored and the single
//     barrier be passed while the operands are still in flight.  Twiddles read from global memory per wave were
//     two thirds of the first version's L1 traffic (15 + 2 KB per 4 frames in float64, against 8 KB of samples);
//   * the transposes go through LDS one component at a time (all real parts, then all imaginary parts), which
//     halves the footprint: 9 KB (N = 512) / 17 KB (N = 400) per wave in float64, half of that in float32; the
//     power spectrum then reuses the same region.  Row pitches are an odd number of 16-byte slots and frame
//     pitches 0 / 8 (mod 16) slots, so the 16-byte row loads are conflict-free;
//   * N = 400: the partner of Z[k1 + 25 k2] in the real-FFT split is element (25 - k1, 7 - k2) (20 x 10: (20 - k1,
//     9 - k2)), so a lane takes the row PAIR (r, 25 - r) and has both halves of all its pairs in its own registers: no
//     cross-lane traffic; N = 2048 does the same with pairs of columns;
//   * no LDS access (and no second flavour of global loads) sits under a lane condition: idle lanes read rows they do not
//     use or shadow a working lane -- a lane-conditional access block costs hipcc 60-100 registers in these kernels;
//   * the wave index is scalar (readfirstlane): the work-item record is one scalar load and the address
//     arithmetic runs on the scalar unit;
//   * the mel reduction runs slot-uniform chunk steps: scalar loop bounds, two LDS reads and four FMAs per step
//     (wave_mel_steps, device_common.h);
//   * float64 plans take the final logarithm in float32 (feature_log): the stored value is a float32 anyway.
//
// Arithmetic (DFT factorisation, twiddle values, real-FFT split) is that of the workgroup-tile kernels, which stay
// in the library as plan option "kernel" = 2.
//
// Reference semantics: sound/sndenv.go:438-478, dft/dft.go:53-85, mel/mel.go:120-153.
#include "device_common.h"

namespace aud {
namespace {

// The workgroup's table blob: loads first (kept in registers), stores after the caller has issued its operand loads.
// NT threads, up to 4 x 16 bytes per thread in flight; larger blobs finish with a plain copy loop.
template <int NT>
struct BlobRegs {
    uint4 v[4];
};
template <int NT>
__device__ __forceinline__ void blob_fetch(const FastArgs& e, int tid, BlobRegs<NT>& b) {
    const uint4* __restrict__ g = static_cast<const uint4*>(e.blob);
    const int n16 = e.blob_bytes >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + NT * q;
        b.v[q] = g[i < n16 ? i : 0];
    }
}
template <int NT>
__device__ __forceinline__ void blob_store(const FastArgs& e, unsigned char* smem, int tid, const BlobRegs<NT>& b) {
    uint4* l = reinterpret_cast<uint4*>(smem);
    const uint4* __restrict__ g = static_cast<const uint4*>(e.blob);
    const int n16 = e.blob_bytes >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + NT * q;
        if (i < n16) l[i] = b.v[q];
    }
#pragma unroll 1
    for (int i = tid + 4 * NT; i < n16; i += NT) l[i] = g[i];
}

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
    // P row pitch: float64 rows are 2 elements longer, an ODD number of 16-byte pieces, so that the epilogue's 16-byte
    // reads of different frames fall on both halves of the bank set (modelled 12.3 -> 8.6 LDS cycles per read)
    static constexpr int kHp = (sizeof(TT) == 8) ? w16::kHp + 2 : w16::kHp;
    static constexpr int kP = kFW * kHp;                       // elements
    static constexpr int kRegion = (kXch > kP ? kXch : kP) * int(sizeof(TT));  // bytes per wave
};
}  // namespace w16

// One wave tile (4 consecutive frames of one item) from its operands to its mel values.  MODE says where the operands
// come from: 0 = `raw` (requested earlier); the next tile's are requested into `raw` before the arithmetic starts;
// 1 = `raw`, nothing requested; 2 = requested and awaited here.
template <typename TT, bool PCM16, int MAXS, int MODE>
__device__ __forceinline__ void w16_tile(const MelspecArgs& a, const FastArgs& e, unsigned char* smem, TT* xw,
                                         const C2<TT>* twa, const C2<TT>* tws, int lane_in, int tiles, int64_t wt,
                                         int64_t wt_next, int64_t total, FrameRaw<16>& raw, aud_item& it, int& item,
                                         int& t0, unsigned* queue_fetch = nullptr) {
    using L = w16::Layout<TT>;
    // the lane id is made opaque per tile: otherwise the compiler hoists what only depends on it out of the tile loop
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int f = lane >> 4;   // frame within the wave
    const int j = lane & 15;   // lane within the frame's 16-lane group
    AUD_STAMP_DECL;
    AUD_STAMP(0);
    AUD_STAMP_REAL(9);
    C2<TT> v[16];
    if constexpr (MODE == 2) {
        item = int(wt / tiles);
        t0 = int(wt - int64_t(item) * tiles) * w16::kFW;
        it = a.items[item];
        load_frame_pairs<TT, 16, 16, w16::kN, PCM16>(a, it, t0 + f, j, v);
    } else {
        frame_pairs_take<TT, 16, 16, w16::kN, PCM16>(a, it, t0 + f, j, raw, v);
    }
    const int item_cur = item, t0_cur = t0;
    const aud_item it_cur = it;
    if (queue_fetch) {  // dynamic grid: ask for the next tile now, the answer is read after this tile's arithmetic
        unsigned got = 0;
        if (lane == 0) got = atomicAdd(a.queue, 1u);
        *queue_fetch = got;
    }
    if constexpr (MODE == 0) {
        if (wt_next < total) {  // the next tile's operands land while this tile is computed
            item = int(wt_next / tiles);
            t0 = int(wt_next - int64_t(item) * tiles) * w16::kFW;
            it = a.items[item];
            frame_pairs_issue<16, 16, w16::kN, PCM16>(a, it, t0 + f, j, raw);
        }
    }

    // ---- pass 1: 16-point DFT over n1, twiddle -----------------------------------------------------------
    AUD_STAMP(3);
    SmallDft<TT, 16>::run(v, nullptr, 0);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[k1] = cmul(v[k1], twa[(k1 - 1) * 16 + j]);
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
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2] = C2<TT>{re[n2], im[n2]};

    // ---- pass 2: row k1 = j: 16-point DFT over n2 -> Z[j + 16 k2] in v[k2] ------------------------------
    SmallDft<TT, 16>::run(v, nullptr, 0);
    wave_lds_fence();  // every row has been read: the region may take the power spectrum
    AUD_STAMP(6);

    // ---- real-FFT split + power (as melspec_r16.hip) -----------------------------------------------------
    // For k = j + 16 q (q = 0..7) the partner Z[256 - k] sits in lane (16 - j) & 15, register 15 - q (lane 0
    // pairs with itself: register (16 - q) & 15).  X[k] = (E + T)/2, X[256-k] = conj(E - T)/2 with
    // E = Z[k] + conj Z[256-k], T = -i W_512^k (Z[k] - conj Z[256-k]).
    TT* Pw = xw;                       // [4][kHp]
    TT* P = Pw + f * L::kHp;
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
            const C2<TT> w = tws[k];                    // W_512^k
            const C2<TT> E = {A.x + B.x, A.y - B.y};    // A + conj(B)
            const C2<TT> D = {A.x - B.x, A.y + B.y};    // A - conj(B)
            const C2<TT> mD = {D.y, -D.x};              // -i D
            const C2<TT> Tm = cmul(mD, w);
            const TT xr = E.x + Tm.x, xi = E.y + Tm.y;  // 2 X[k]
            const TT yr = E.x - Tm.x, yi = E.y - Tm.y;  // 2 conj X[256-k]
            P[k] = xr * xr + xi * xi;  // FOUR times the power: the 1/4 of the split lives in the blob's mel weights
            if (k != 0) P[w16::kM - k] = yr * yr + yi * yi;
            else P[w16::kM] = yr * yr + yi * yi;  // k = 0 also yields the Nyquist bin
        }
        // k = 128 (lane 0, register 8) pairs with itself: X[128] = conj(Z[128])
        if (j == 0) P[128] = TT(4) * (v[8].x * v[8].x + v[8].y * v[8].y);  // (x 4 like every bin of P)
        // bins 257..259 only pad the last 4-bin chunk; their weights are zero but 0 * garbage must stay 0
        if (j >= 13) P[w16::kH + (j - 13)] = TT(0);
    }
    wave_lds_fence();
    AUD_STAMP(7);

    // ---- optional spectrum outputs and the mel reduction: 4 frames x 16 filter groups on this wave ----------
    wave_mel_steps<TT, w16::kFW, MAXS>(a, e, Pw, L::kHp, smem, it_cur, item_cur, t0_cur, lane);
    AUD_STAMP(8);
    AUD_STAMP_REAL(10);
    AUD_STAMP_FLUSH(a, wt, lane);
    wave_lds_fence();  // the region is free for the next tile
}

// Persistent kernel: the workgroup stages the table blob once, then each of its waves walks wave tiles wt, wt + stride, ...
// VAR (A/B, plan option "wave_variant"): 0 = every tile requests the next tile's operands before it computes
// (32 more live registers); 1 = the same with the register budget capped for 3 (float64) / 5 (float32) waves per SIMD;
// 2 = only the first tile's operands are requested early (under the blob staging), later tiles load at their top.
template <typename TT, bool PCM16, int NW, int MAXS, int VAR>
__global__ __launch_bounds__(64 * NW, VAR == 1 ? (sizeof(TT) == 8 ? 3 : 5) : 1) void k_melspec_w16(const MelspecArgs a,
                                                                                                    const FastArgs e) {
    using L = w16::Layout<TT>;
    unsigned char* smem = dyn_lds();
    const int tid = int(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: item record and addresses on the SALU
    const int lane = tid & 63;

    // the workgroup's tables: requested before anything else so that a counted wait can pick them out
    BlobRegs<64 * NW> blob;
    blob_fetch<64 * NW>(e, tid, blob);

    const int tiles = (a.T + w16::kFW - 1) / w16::kFW;  // wave tiles per item
    const int64_t total = int64_t(a.n_items) * tiles;
    const int64_t stride = int64_t(gridDim.x) * NW;
    const unsigned wg = tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap);
    int64_t wt = int64_t(wg) * NW + wave;
    int item = wt < total ? int(wt / tiles) : 0;
    int t0 = wt < total ? int(wt - int64_t(item) * tiles) * w16::kFW : 0;
    aud_item it = a.items[item];

    // pass 1 operands of the first tile: z[16 n1 + j] = (x[32 n1 + 2j], x[.. + 1])
    FrameRaw<16> raw;
    raw.route = 0;
    if (wt < total) frame_pairs_issue<16, 16, w16::kN, PCM16>(a, it, t0 + (lane >> 4), lane & 15, raw);

    blob_store<64 * NW>(e, smem, tid, blob);  // waits for the blob loads only: the operands stay in flight
    __syncthreads();                          // the one barrier: tables visible to the workgroup's waves

    TT* xw = reinterpret_cast<TT*>(smem + e.xch_off + wave * L::kRegion);  // this wave's region
    const C2<TT>* twa = reinterpret_cast<const C2<TT>*>(smem + e.twa_off);  // W_512^(2 j k1) at [(k1 - 1) 16 + j]
    const C2<TT>* tws = reinterpret_cast<const C2<TT>*>(smem + e.tws_off);  // W_512^k, k <= 128

    if constexpr (VAR == 3) {
        // dynamic tile queue: the first tile of every wave is static (its operands are already in flight), further
        // tiles are handed out by one returning atomic each, requested a tile ahead.  The last wave to leave resets
        // the slot for the next launch.
        const unsigned grid_waves = gridDim.x * NW;
        bool first = true;
        while (wt < total) {
            unsigned nxt = 0;
            if (first) w16_tile<TT, PCM16, MAXS, 1>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt, total, raw, it, item, t0, &nxt);
            else w16_tile<TT, PCM16, MAXS, 2>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt, total, raw, it, item, t0, &nxt);
            first = false;
            wt = int64_t(grid_waves) + unsigned(__builtin_amdgcn_readfirstlane(int(nxt)));
        }
        if (lane == 0) {
            const unsigned left = atomicAdd(a.queue + 16, 1u);
            if (left == grid_waves - 1) {  // every other wave has left: nobody touches the slot any more
                atomicExch(a.queue, 0u);
                atomicExch(a.queue + 16, 0u);
            }
        }
    } else if constexpr (VAR == 2) {
        if (wt < total) {
            w16_tile<TT, PCM16, MAXS, 1>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt + stride, total, raw, it, item, t0);
            wt += stride;
        }
        while (wt < total) {
            w16_tile<TT, PCM16, MAXS, 2>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt + stride, total, raw, it, item, t0);
            wt += stride;
        }
    } else {
        while (wt < total) {
            w16_tile<TT, PCM16, MAXS, 0>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt + stride, total, raw, it, item, t0);
            wt += stride;
        }
    }
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
    static constexpr int kHp = (sizeof(TT) == 8) ? w25::kHp + 2 : w25::kHp;  // float64: odd number of 16-byte pieces (see w16)
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
    P[k] = xr * xr + xi * xi;  // FOUR times the power: the 1/4 of the split lives in the blob's mel weights
    P[kM - k] = yr * yr + yi * yi;  // k = 0 -> Nyquist bin 200; k = 100 -> the same bin, same value
}
}  // namespace w25

// One wave tile (8 consecutive frames of one item); MODE as w16_tile.
template <typename TT, bool PCM16, int MAXS, int MODE>
__device__ __forceinline__ void w25_tile(const MelspecArgs& a, const FastArgs& e, unsigned char* smem, TT* xw,
                                         const C2<TT>* twa, const C2<TT>* tws, int lane_in, int tiles, int64_t wt,
                                         int64_t wt_next, int64_t total, FrameRaw<25>& raw, aud_item& it, int& item,
                                         int& t0, unsigned* queue_fetch = nullptr) {
    using L = w25::Layout<TT>;
    int lane = lane_in;  // opaque per tile (see w16_tile)
    asm volatile("" : "+v"(lane));
    const int f = lane >> 3;  // frame within the wave
    const int j = lane & 7;   // lane within the frame's 8-lane group
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.tw);  // W_400^k (the 25-point DFT's wave-uniform inner twiddles)
    AUD_STAMP_DECL;
    AUD_STAMP(0);
    AUD_STAMP_REAL(9);
    C2<TT> v[25];
    if constexpr (MODE == 2) {
        item = int(wt / tiles);
        t0 = int(wt - int64_t(item) * tiles) * w25::kFW;
        it = a.items[item];
        load_frame_pairs<TT, 25, 8, w25::kN, PCM16>(a, it, t0 + f, j, v);
    } else {
        frame_pairs_take<TT, 25, 8, w25::kN, PCM16>(a, it, t0 + f, j, raw, v);
    }
    const int item_cur = item, t0_cur = t0;
    const aud_item it_cur = it;
    if (queue_fetch) {  // dynamic grid: ask for the next tile now, the answer is read after this tile's arithmetic
        unsigned got = 0;
        if (lane == 0) got = atomicAdd(a.queue, 1u);
        *queue_fetch = got;
    }
    if constexpr (MODE == 0) {
        if (wt_next < total) {  // the next tile's operands land while this tile is computed
            item = int(wt_next / tiles);
            t0 = int(wt_next - int64_t(item) * tiles) * w25::kFW;
            it = a.items[item];
            frame_pairs_issue<25, 8, w25::kN, PCM16>(a, it, t0 + f, j, raw);
        }
    }
    AUD_STAMP(3);

    // ---- pass A: 25-point DFT over n1, twiddle W_200^(j k1) = W_400^(2 j k1) -------------------------------
    SmallDft<TT, 25>::run(v, tw, w25::kN);
#pragma unroll
    for (int k1 = 1; k1 < 25; ++k1) v[k1] = cmul(v[k1], twa[(k1 - 1) * 8 + j]);
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
    w25::read_row8<TT>(rows + r1 * L::kRow, ur[2]);   // unconditional (lanes without a slot 1 read valid rows they do not
    w25::read_row8<TT>(rows + r1p * L::kRow, ur[3]);  // use): reads under a lane condition cost 100 registers in float64
    wave_lds_fence();
#pragma unroll
    for (int k1 = 0; k1 < 25; ++k1) col[k1 * L::kRow] = v[k1].y;
    wave_lds_fence();
    w25::read_row8<TT>(rows + r0 * L::kRow, ui[0]);
    w25::read_row8<TT>(rows + r0p * L::kRow, ui[1]);
    w25::read_row8<TT>(rows + r1 * L::kRow, ui[2]);
    w25::read_row8<TT>(rows + r1p * L::kRow, ui[3]);
    wave_lds_fence();  // every row has been read: the region may take the power spectrum
    AUD_STAMP(5);

    // ---- pass B (8-point DFT over n2 of each row, in registers: Z[k1 + 25 k2]) + real-FFT split + power ---------
    // pairs are always evaluated from their k <= 100 side, A = Z[k], B = Z[200 - k], as melspec_r25.hip does:
    // rows (r, r' = 25 - r), r <= 12: k = r + 25 c and k = r' + 25 c for c = 0..3, partners in column 7 - c
    TT* Pw = xw;  // [8][kHp]
    TT* P = Pw + f * L::kHp;
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
                    w25::split_pair<TT>(P, tws, ra + 25 * c, za[c], zb[7 - c]);
                    w25::split_pair<TT>(P, tws, rb + 25 * c, zb[c], za[7 - c]);
                }
            } else {  // row 0: k = 25 c pairs with column 8 - c of the same row (c = 0: DC + Nyquist; c = 4: itself)
#pragma unroll
                for (int c = 0; c <= 4; ++c) w25::split_pair<TT>(P, tws, 25 * c, za[c], za[(8 - c) & 7]);
            }
        }
    }
    AUD_STAMP(6);
    if (j < 3) P[w25::kH + j] = TT(0);  // pad bins of the last 4-bin chunk
    wave_lds_fence();
    AUD_STAMP(7);

    // ---- optional spectrum outputs and the mel reduction: 8 frames x 8 filter groups on this wave -------------
    wave_mel_steps<TT, w25::kFW, MAXS>(a, e, Pw, L::kHp, smem, it_cur, item_cur, t0_cur, lane);
    AUD_STAMP(8);
    AUD_STAMP_REAL(10);
    AUD_STAMP_FLUSH(a, wt, lane);
    wave_lds_fence();  // the region is free for the next tile
}

// Persistent kernel, as k_melspec_w16.  VAR 0 / 1 / 2 likewise (the float64 instantiation only exists as VAR 2: the
// 50 registers of a prefetched tile would all spill).  Second launch-bounds argument = waves per SIMD the register
// allocator must leave room for: LDS admits two (float64) / three (float32).
template <typename TT, bool PCM16, int NW, int MAXS, int VAR>
__global__ __launch_bounds__(64 * NW, sizeof(TT) == 8 ? 2 : (VAR == 1 ? 3 : 2)) void k_melspec_w25(const MelspecArgs a,
                                                                                                   const FastArgs e) {
    using L = w25::Layout<TT>;
    unsigned char* smem = dyn_lds();
    const int tid = int(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: item record and addresses on the SALU
    const int lane = tid & 63;

    BlobRegs<64 * NW> blob;
    blob_fetch<64 * NW>(e, tid, blob);

    const int tiles = (a.T + w25::kFW - 1) / w25::kFW;  // wave tiles per item
    const int64_t total = int64_t(a.n_items) * tiles;
    const int64_t stride = int64_t(gridDim.x) * NW;
    const unsigned wg = tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap);
    int64_t wt = int64_t(wg) * NW + wave;
    int item = wt < total ? int(wt / tiles) : 0;
    int t0 = wt < total ? int(wt - int64_t(item) * tiles) * w25::kFW : 0;
    aud_item it = a.items[item];

    // pass A operands of the first tile: z[8 n1 + j] = (x[16 n1 + 2j], x[16 n1 + 2j + 1]), n1 = 0..24
    FrameRaw<25> raw;
    raw.route = 0;
    if (wt < total) frame_pairs_issue<25, 8, w25::kN, PCM16>(a, it, t0 + (lane >> 3), lane & 7, raw);

    blob_store<64 * NW>(e, smem, tid, blob);  // waits for the blob loads only: the operands stay in flight
    __syncthreads();                          // the one barrier: tables visible to the workgroup's waves

    TT* xw = reinterpret_cast<TT*>(smem + e.xch_off + wave * L::kRegion);  // this wave's region
    const C2<TT>* twa = reinterpret_cast<const C2<TT>*>(smem + e.twa_off);  // W_400^(2 j k1) at [(k1 - 1) 8 + j]
    const C2<TT>* tws = reinterpret_cast<const C2<TT>*>(smem + e.tws_off);  // W_400^k, k <= 100

    if constexpr (VAR == 3) {  // dynamic tile queue, as k_melspec_w16
        const unsigned grid_waves = gridDim.x * NW;
        bool first = true;
        while (wt < total) {
            unsigned nxt = 0;
            if (first) w25_tile<TT, PCM16, MAXS, 1>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt, total, raw, it, item, t0, &nxt);
            else w25_tile<TT, PCM16, MAXS, 2>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt, total, raw, it, item, t0, &nxt);
            first = false;
            wt = int64_t(grid_waves) + unsigned(__builtin_amdgcn_readfirstlane(int(nxt)));
        }
        if (lane == 0) {
            const unsigned left = atomicAdd(a.queue + 16, 1u);
            if (left == grid_waves - 1) {  // every other wave has left: nobody touches the slot any more
                atomicExch(a.queue, 0u);
                atomicExch(a.queue + 16, 0u);
            }
        }
    } else if constexpr (VAR == 2) {
        if (wt < total) {
            w25_tile<TT, PCM16, MAXS, 1>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt + stride, total, raw, it, item, t0);
            wt += stride;
        }
        while (wt < total) {
            w25_tile<TT, PCM16, MAXS, 2>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt + stride, total, raw, it, item, t0);
            wt += stride;
        }
    } else {
        while (wt < total) {
            w25_tile<TT, PCM16, MAXS, 0>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt + stride, total, raw, it, item, t0);
            wt += stride;
        }
    }
}


// ================================================================================================
// N = 400, second geometry: 200-point complex FFT as 20 x 10, 10 lanes per frame, 6 frames per wave (60 of 64 lanes)
// ================================================================================================
// The 25 x 8 kernel keeps 25 points and then four 8-column rows per lane: 174 registers and 17 KB of LDS per wave in
// float64 -- two waves per SIMD, and the stamps show its vector ALU waiting.  20 x 10 gives a lane 20 points and then
// exactly one row PAIR (r, 20 - r) of 10 columns: every lane has both halves of its ten split pairs locally (lane 0
// takes the two self-paired rows 0 and 10), two thirds of the registers, 10 KB of LDS per wave: three waves per SIMD.
namespace w20 {
constexpr int kFW = 6;    // frames per wave
constexpr int kLPF = 10;  // lanes per frame
constexpr int kM = 200, kN = 400, kH = 201, kHp = 204;
template <typename TT>
struct Layout {
    // scalar transpose rows of 10 (+2 pad in float32: 16-byte rows); pitches from a search over the hardware's lane
    // groups (10-lane frames cannot be conflict-free against 16-lane groups: 2-way at best, which these reach)
    static constexpr int kRow = (sizeof(TT) == 4) ? 12 : 10;
    static constexpr int kFrame = (sizeof(TT) == 4) ? 248 : 206;
    static constexpr int kXch = kFW * kFrame;
    static constexpr int kHp = (sizeof(TT) == 8) ? w20::kHp + 2 : w20::kHp;  // float64: odd number of 16-byte pieces (see w16)
    static constexpr int kP = kFW * kHp;
    static constexpr int kRegion = ((kXch > kP ? kXch : kP) * int(sizeof(TT)) + 31) & ~31;  // bytes per wave
};
template <typename TT>
__device__ __forceinline__ void read_row10(const TT* row, C2<TT> (&z)[10], bool imag) {
    TT d[12];
    if constexpr (sizeof(TT) == 4) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const Q4<TT> r4 = reinterpret_cast<const Q4<TT>*>(row)[q];
            d[4 * q] = r4.x; d[4 * q + 1] = r4.y; d[4 * q + 2] = r4.z; d[4 * q + 3] = r4.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const C2<TT> r2 = reinterpret_cast<const C2<TT>*>(row)[q];
            d[2 * q] = r2.x; d[2 * q + 1] = r2.y;
        }
    }
#pragma unroll
    for (int n2 = 0; n2 < 10; ++n2) {
        if (imag) z[n2].y = d[n2];
        else z[n2].x = d[n2];
    }
}
}  // namespace w20

template <typename TT, bool PCM16, int MAXS, int MODE>
__device__ __forceinline__ void w20_tile(const MelspecArgs& a, const FastArgs& e, unsigned char* smem, TT* xw,
                                         const C2<TT>* twa, const C2<TT>* tws, int lane_in, int tiles, int64_t wt,
                                         int64_t wt_next, int64_t total, FrameRaw<20>& raw, aud_item& it, int& item,
                                         int& t0, unsigned* queue_fetch = nullptr) {
    using L = w20::Layout<TT>;
    int lane = lane_in;  // opaque per tile (see w16_tile)
    asm volatile("" : "+v"(lane));
    // lanes 60..63 have no frame of their own: they SHADOW lanes 50..53 (same frame, same column) through the whole FFT --
    // same loads, same arithmetic, same values stored to the same LDS addresses -- so that no LDS access sits under a
    // lane condition (a lane-conditional block of LDS stores or loads costs hipcc 60-100 registers here)
    const bool own = lane < w20::kFW * w20::kLPF;
    const int f = own ? lane / w20::kLPF : w20::kFW - 1;
    const int j = own ? lane - f * w20::kLPF : lane - w20::kFW * w20::kLPF;
    AUD_STAMP_DECL;
    AUD_STAMP(0);
    AUD_STAMP_REAL(9);
    C2<TT> v[20];
    if constexpr (MODE == 2) {
        item = int(wt / tiles);
        t0 = int(wt - int64_t(item) * tiles) * w20::kFW;
        it = a.items[item];
        load_frame_pairs<TT, 20, 10, w20::kN, PCM16>(a, it, t0 + f, j, v);
    } else {
        frame_pairs_take<TT, 20, 10, w20::kN, PCM16>(a, it, t0 + f, j, raw, v);
    }
    const int item_cur = item, t0_cur = t0;
    const aud_item it_cur = it;
    if (queue_fetch) {
        unsigned got = 0;
        if (lane == 0) got = atomicAdd(a.queue, 1u);
        *queue_fetch = got;
    }
    AUD_STAMP(3);

    // ---- pass A: 20-point DFT over n1, twiddle W_200^(j k1) = W_400^(2 j k1) -------------------------------
    SmallDft<TT, 20>::run(v, nullptr, 0);
#pragma unroll
    for (int k1 = 1; k1 < 20; ++k1) v[k1] = cmul(v[k1], twa[(k1 - 1) * w20::kLPF + j]);
    AUD_STAMP(4);

    // ---- transpose (real parts, then imaginary parts): element (row k1, column n2 = j) of frame f; afterwards the lane
    // holds the row pair (j, 20 - j) -- lane 0 the self-paired rows 0 and 10 -- with all ten columns of each
    TT* col = xw + f * L::kFrame + j;
    constexpr int cstep = L::kRow;
    const TT* rows = xw + f * L::kFrame;
    const int ra = j, rb = j == 0 ? 10 : 20 - j;
    C2<TT> za[10], zb[10];
    AUD_BENIGN_RACE_BEGIN();
#pragma unroll
    for (int k1 = 0; k1 < 20; ++k1) col[k1 * cstep] = v[k1].x;
    AUD_BENIGN_RACE_END();
    wave_lds_fence();
    w20::read_row10<TT>(rows + ra * L::kRow, za, false);
    w20::read_row10<TT>(rows + rb * L::kRow, zb, false);
    wave_lds_fence();
    AUD_BENIGN_RACE_BEGIN();
#pragma unroll
    for (int k1 = 0; k1 < 20; ++k1) col[k1 * cstep] = v[k1].y;
    AUD_BENIGN_RACE_END();
    wave_lds_fence();
    w20::read_row10<TT>(rows + ra * L::kRow, za, true);
    w20::read_row10<TT>(rows + rb * L::kRow, zb, true);
    wave_lds_fence();  // every row has been read: the region may take the power spectrum
    AUD_STAMP(5);
    if constexpr (MODE == 0) {
        // the next tile's operands are requested HERE, not at the top: the 40 registers of v[] are dead from this point
        // on, so the raw words in flight do not add to the kernel's register peak (pass A + transposes)
        if (wt_next < total) {
            item = int(wt_next / tiles);
            t0 = int(wt_next - int64_t(item) * tiles) * w20::kFW;
            it = a.items[item];
            frame_pairs_issue<20, 10, w20::kN, PCM16>(a, it, t0 + f, j, raw);
        }
    }

    // ---- pass B: 10-point DFT over n2 of both rows: Z[k1 + 20 k2] -----------------------------------------------
    SmallDft<TT, 10>::run(za, nullptr, 0);
    SmallDft<TT, 10>::run(zb, nullptr, 0);
    AUD_STAMP(6);

    // ---- real-FFT split + power: the partner of Z[k1 + 20 k2] is element (20 - k1, 9 - k2); pairs are evaluated from
    // their k <= 100 side (A = Z[k], B = Z[200 - k]) as everywhere else
    TT* Pw = xw;  // [6][kHp]
    TT* P = Pw + f * L::kHp;
    AUD_BENIGN_RACE_BEGIN();
    {
        // lanes 1..9: rows (j, 20 - j): k = j + 20 c pairs with (row 20 - j, column 9 - c) and vice versa, c = 0..4;
        // lane 0: row 0: k = 20 c pairs with column 10 - c of the same row (c = 0: DC + Nyquist; c = 5: itself), row 10:
        // k = 10 + 20 c pairs with column 9 - c of the same row.  One code path, partners selected by value.
        const bool self = j == 0;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const C2<TT> pa = za[(10 - c) % 10], pb = zb[9 - c], pc = za[9 - c];
            const C2<TT> b_first = {self ? pa.x : pb.x, self ? pa.y : pb.y};
            const C2<TT> b_second = {self ? pb.x : pc.x, self ? pb.y : pc.y};
            w25::split_pair<TT>(P, tws, ra + 20 * c, za[c], b_first);
            w25::split_pair<TT>(P, tws, rb + 20 * c, zb[c], b_second);
        }
        // lane 0's eleventh pair, k = 100 (row 0, column 5, paired with itself); the other lanes repeat their k = j pair
        w25::split_pair<TT>(P, tws, self ? 100 : ra, self ? za[5] : za[0], self ? za[5] : zb[9]);
        P[w20::kH + (j < 3 ? j : 0)] = TT(0);  // pad bins 201..203 of the last 4-bin chunk
    }
    AUD_BENIGN_RACE_END();
    wave_lds_fence();
    AUD_STAMP(7);

    // ---- optional spectrum outputs and the mel reduction: 6 frames x 10 filter groups on this wave ---------------
    wave_mel_steps<TT, w20::kFW, MAXS>(a, e, Pw, L::kHp, smem, it_cur, item_cur, t0_cur, lane);
    AUD_STAMP(8);
    AUD_STAMP_REAL(10);
    AUD_STAMP_FLUSH(a, wt, lane);
    wave_lds_fence();  // the region is free for the next tile
}

template <typename TT, bool PCM16, int NW, int MAXS, int VAR>
__global__ __launch_bounds__(64 * NW, sizeof(TT) == 8 ? 3 : 4) void k_melspec_w20(const MelspecArgs a, const FastArgs e) {
    using L = w20::Layout<TT>;
    unsigned char* smem = dyn_lds();
    const int tid = int(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    BlobRegs<64 * NW> blob;
    blob_fetch<64 * NW>(e, tid, blob);

    const int tiles = (a.T + w20::kFW - 1) / w20::kFW;  // wave tiles per item
    const int64_t total = int64_t(a.n_items) * tiles;
    const int64_t stride = int64_t(gridDim.x) * NW;
    const unsigned wg = tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap);
    int64_t wt = int64_t(wg) * NW + wave;
    int item = wt < total ? int(wt / tiles) : 0;
    int t0 = wt < total ? int(wt - int64_t(item) * tiles) * w20::kFW : 0;
    aud_item it = a.items[item];

    FrameRaw<20> raw;
    raw.route = 0;
    {
        const bool own = lane < w20::kFW * w20::kLPF;  // lanes 60..63 shadow lanes 50..53 (w20_tile)
        const int f0 = own ? lane / w20::kLPF : w20::kFW - 1;
        const int j0 = own ? lane - f0 * w20::kLPF : lane - w20::kFW * w20::kLPF;
        if (wt < total) frame_pairs_issue<20, 10, w20::kN, PCM16>(a, it, t0 + f0, j0, raw);
    }

    blob_store<64 * NW>(e, smem, tid, blob);
    __syncthreads();  // the one barrier: tables visible to the workgroup's waves

    TT* xw = reinterpret_cast<TT*>(smem + e.xch_off + wave * L::kRegion);
    const C2<TT>* twa = reinterpret_cast<const C2<TT>*>(smem + e.twa_off);  // W_400^(2 j k1) at [(k1 - 1) 10 + j]
    const C2<TT>* tws = reinterpret_cast<const C2<TT>*>(smem + e.tws_off);  // W_400^k, k <= 100

    if constexpr (VAR == 3) {
        const unsigned grid_waves = gridDim.x * NW;
        bool first = true;
        while (wt < total) {
            unsigned nxt = 0;
            if (first) w20_tile<TT, PCM16, MAXS, 1>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt, total, raw, it, item, t0, &nxt);
            else w20_tile<TT, PCM16, MAXS, 2>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt, total, raw, it, item, t0, &nxt);
            first = false;
            wt = int64_t(grid_waves) + unsigned(__builtin_amdgcn_readfirstlane(int(nxt)));
        }
        if (lane == 0) {
            const unsigned left = atomicAdd(a.queue + 16, 1u);
            if (left == grid_waves - 1) {
                atomicExch(a.queue, 0u);
                atomicExch(a.queue + 16, 0u);
            }
        }
    } else if constexpr (VAR == 2) {
        if (wt < total) {
            w20_tile<TT, PCM16, MAXS, 1>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt + stride, total, raw, it, item, t0);
            wt += stride;
        }
        while (wt < total) {
            w20_tile<TT, PCM16, MAXS, 2>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt + stride, total, raw, it, item, t0);
            wt += stride;
        }
    } else {
        while (wt < total) {
            w20_tile<TT, PCM16, MAXS, 0>(a, e, smem, xw, twa, tws, lane, tiles, wt, wt + stride, total, raw, it, item, t0);
            wt += stride;
        }
    }
}

// ================================================================================================
// N = 2048 (BASELINE config 5): 1024-point complex FFT as 16 x 16 x 4, ONE frame per wave
// ================================================================================================
//   n = 64 n1 + l,  l = 4 n2 + n3 = lane;   k = k1 + 16 k2 + 256 k3
//   pass 1  lane l: 16-point DFT over n1 of z[64 n1 + l] (512 contiguous bytes per load), twiddle W_1024^(l k1) multiplied
//           together from four lane-ordered table values in global memory (the whole table in LDS would cost a wave per SIMD)
//   -- transpose through the wave's LDS plane (real parts, then imaginary parts): rows k1 of 64 + 4 --
//   pass 2  lane (k1, n3): 16-point DFT over n2, twiddle W_64^(n3 k2) (LDS, 4 distinct rows)
//   -- second transpose, plane [k1][k2][n3]: a reader takes the four n3 of a column (k1, k2) as one 16/32-byte read --
//   pass 3 + split: every lane owns two COLUMN PAIRS {(k1, k2), partner column}: the partner of Z[k1 + 16 k2 + 256 k3] in
//           the real-FFT split is Z[1024 - k] = element (16 - k1, 15 - k2, 3 - k3) (with carries for k1 = 0), so after the
//           lane's four 4-point DFTs over n3 both halves of all its pairs are in its own registers: no shuffles, no
//           spectrum scatter/gather.  127 regular pairs of columns + one slot (lane 63's second) holding the two
//           self-paired columns (0, 8) and (0, 0); that slot runs the same code with partners selected by value.
//   power (x 4) to the wave's plane, then the shared slot-uniform mel epilogue with 64 filter groups (one per lane).
// A wave walks kFPW consecutive frames so that a workgroup's one table staging serves NW x kFPW frames.
namespace w64 {
constexpr int kM = 1024, kN = 2048, kH = 1025, kHp = 1028;
constexpr int kRow = 68;           // transpose rows: 64 (= 16 x 4) elements + 4
constexpr int kPlane = 16 * kRow;  // 1088 elements, also holds the power row
constexpr int kFPW = 4;            // consecutive frames per wave
template <typename TT>
struct Layout {
    static constexpr int kHp = (sizeof(TT) == 8) ? w64::kHp + 2 : w64::kHp;  // float64: odd number of 16-byte pieces (see w16)
    static constexpr int kRegion = kPlane * int(sizeof(TT));                  // bytes per wave
};
template <typename TT>
__device__ __forceinline__ void split_pair(TT* P, C2<TT> w, int k, C2<TT> A, C2<TT> B) {
    const C2<TT> E = {A.x + B.x, A.y - B.y};
    const C2<TT> D = {A.x - B.x, A.y + B.y};
    const C2<TT> mD = {D.y, -D.x};
    const C2<TT> Tm = cmul(mD, w);
    const TT xr = E.x + Tm.x, xi = E.y + Tm.y;
    const TT yr = E.x - Tm.x, yi = E.y - Tm.y;
    P[k] = xr * xr + xi * xi;  // FOUR times the power (the 1/4 lives in the mel weights)
    P[kM - k] = yr * yr + yi * yi;
}
}  // namespace w64

template <typename TT, bool PCM16, int MAXS>
__device__ __forceinline__ void w64_tile(const MelspecArgs& a, const FastArgs& e, unsigned char* smem, TT* xw, int lane_in,
                                         int64_t wt) {
    using L = w64::Layout<TT>;
    int lane = lane_in;  // opaque per tile (see w16_tile)
    asm volatile("" : "+v"(lane));
    const int item = int(wt / a.T);
    const int sstep = int(wt - int64_t(item) * a.T);
    const aud_item it = a.items[item];
    AUD_STAMP_DECL;
    AUD_STAMP(0);
    AUD_STAMP_REAL(9);
    C2<TT> v[16];
    load_frame_pairs<TT, 16, 64, w64::kN, PCM16, true>(a, it, sstep, lane, v);
    AUD_STAMP(3);

    // ---- pass 1 ------------------------------------------------------------------------------------------------
    SmallDft<TT, 16>::run(v, nullptr, 0);
    {
        // W_1024^(l k1), k1 = 1..15, as products of at most three of the four table values k1 = 1, 2, 4, 8 (lane-ordered,
        // global memory: 4 KB in float64 that stay in L1; the full [15][64] table read per frame was twice the sample bytes)
        const C2<TT>* __restrict__ g1 = static_cast<const C2<TT>*>(e.gtab) + lane;
        const C2<TT> b1 = g1[0], b2 = g1[64], b4 = g1[128], b8 = g1[192];
        const C2<TT> w3 = cmul(b1, b2), w5 = cmul(b1, b4), w6 = cmul(b2, b4), w9 = cmul(b1, b8), w10 = cmul(b2, b8),
                     w12 = cmul(b4, b8);
        const C2<TT> w7 = cmul(w3, b4), w11 = cmul(w3, b8), w13 = cmul(w5, b8), w14 = cmul(w6, b8);
        v[1] = cmul(v[1], b1);
        v[2] = cmul(v[2], b2);
        v[3] = cmul(v[3], w3);
        v[4] = cmul(v[4], b4);
        v[5] = cmul(v[5], w5);
        v[6] = cmul(v[6], w6);
        v[7] = cmul(v[7], w7);
        v[8] = cmul(v[8], b8);
        v[9] = cmul(v[9], w9);
        v[10] = cmul(v[10], w10);
        v[11] = cmul(v[11], w11);
        v[12] = cmul(v[12], w12);
        v[13] = cmul(v[13], w13);
        v[14] = cmul(v[14], w14);
        v[15] = cmul(v[15], cmul(w7, b8));
    }
    AUD_STAMP(4);
    // ---- transpose 1: row k1, column l; lane (k1r, n3) then holds column 4 n2 + n3 of row k1r ------------------------
    const int k1r = lane >> 2, n3 = lane & 3;
    {
        TT re[16];
        TT* wcol = xw + lane;
        const TT* rcol = xw + k1r * w64::kRow + n3;
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) wcol[k1 * w64::kRow] = v[k1].x;
        wave_lds_fence();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) re[n2] = rcol[4 * n2];
        wave_lds_fence();
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) wcol[k1 * w64::kRow] = v[k1].y;
        wave_lds_fence();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) v[n2] = C2<TT>{re[n2], rcol[4 * n2]};
        wave_lds_fence();
    }
    AUD_STAMP(5);
    // ---- pass 2: DFT over n2, twiddle W_64^(n3 k2) ---------------------------------------------------------------
    SmallDft<TT, 16>::run(v, nullptr, 0);
    {
        const C2<TT>* tw2 = reinterpret_cast<const C2<TT>*>(smem + e.twa_off) + n3 * 16;
#pragma unroll
        for (int k2 = 1; k2 < 16; ++k2) v[k2] = cmul(v[k2], tw2[k2]);
    }
    // ---- transpose 2: plane [k1][k2][n3] (row k1 of 64 + 4); the lane's two column pairs come back as 4-element reads --
    const unsigned short* pr = reinterpret_cast<const unsigned short*>(smem + e.pairs_off) + 4 * lane;  // ka0 kb0 ka1 kb1
    const int ka0 = pr[0], kb0 = pr[1], ka1 = pr[2], kb1 = pr[3];  // column base bins k1 + 16 k2 (k3 = 0)
    C2<TT> za[2][4], zb[2][4];
    {
        TT* wrow = xw + k1r * w64::kRow + n3;
        // column (k1, k2) = base bin k: k1 = k & 15, k2 = k >> 4
        const Q4<TT>* ca0 = reinterpret_cast<const Q4<TT>*>(xw + (ka0 & 15) * w64::kRow + 4 * (ka0 >> 4));
        const Q4<TT>* cb0 = reinterpret_cast<const Q4<TT>*>(xw + (kb0 & 15) * w64::kRow + 4 * (kb0 >> 4));
        const Q4<TT>* ca1 = reinterpret_cast<const Q4<TT>*>(xw + (ka1 & 15) * w64::kRow + 4 * (ka1 >> 4));
        const Q4<TT>* cb1 = reinterpret_cast<const Q4<TT>*>(xw + (kb1 & 15) * w64::kRow + 4 * (kb1 >> 4));
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) wrow[4 * k2] = v[k2].x;
        wave_lds_fence();
        {
            const Q4<TT> a0 = *ca0, b0 = *cb0, a1 = *ca1, b1 = *cb1;
            za[0][0].x = a0.x; za[0][1].x = a0.y; za[0][2].x = a0.z; za[0][3].x = a0.w;
            zb[0][0].x = b0.x; zb[0][1].x = b0.y; zb[0][2].x = b0.z; zb[0][3].x = b0.w;
            za[1][0].x = a1.x; za[1][1].x = a1.y; za[1][2].x = a1.z; za[1][3].x = a1.w;
            zb[1][0].x = b1.x; zb[1][1].x = b1.y; zb[1][2].x = b1.z; zb[1][3].x = b1.w;
        }
        wave_lds_fence();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) wrow[4 * k2] = v[k2].y;
        wave_lds_fence();
        {
            const Q4<TT> a0 = *ca0, b0 = *cb0, a1 = *ca1, b1 = *cb1;
            za[0][0].y = a0.x; za[0][1].y = a0.y; za[0][2].y = a0.z; za[0][3].y = a0.w;
            zb[0][0].y = b0.x; zb[0][1].y = b0.y; zb[0][2].y = b0.z; zb[0][3].y = b0.w;
            za[1][0].y = a1.x; za[1][1].y = a1.y; za[1][2].y = a1.z; za[1][3].y = a1.w;
            zb[1][0].y = b1.x; zb[1][1].y = b1.y; zb[1][2].y = b1.z; zb[1][3].y = b1.w;
        }
        wave_lds_fence();  // every column has been read: the plane may take the power spectrum
    }
    AUD_STAMP(6);
    // ---- pass 3 (4-point DFTs over n3) + split + power ------------------------------------------------------------
    TT* P = xw;  // [kHp]
    {
        const C2<TT>* __restrict__ gs = static_cast<const C2<TT>*>(e.gtab) + 4 * 64 + lane;  // [2][64]: W_2048^ka of the lane's slots
        const TT r8 = TT(0.70710678118654752440L);
        const C2<TT> c8 = {r8, -r8};  // W_2048^256
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            dft4(za[s][0], za[s][1], za[s][2], za[s][3]);
            dft4(zb[s][0], zb[s][1], zb[s][2], zb[s][3]);
            const int ka = s ? ka1 : ka0, kb = s ? kb1 : kb0;
            // regular slot: Za[k3] pairs with Zb[3 - k3], each pair evaluated from its k <= 512 side.  Lane 63's second slot
            // holds the self-paired columns a = (0, 8) (k3 <-> 3 - k3) and b = (0, 0) (DC + Nyquist, 1 <-> 3, 512 with itself)
            const bool sp = s == 1 && lane == 63;
            // twiddles of the slot's pairs from W^ka: W^(ka + 256) = W^ka W^256; kb = 256 - ka: W^kb = W^256 conj(W^ka),
            // W^(kb + 256) = W^512 conj(W^ka) = -i conj(W^ka); the special slot's b column has k = 0, 256, 512
            const C2<TT> w0 = gs[s * 64];
            const C2<TT> w1 = cmul(w0, c8);
            const C2<TT> cw = {w0.x, -w0.y};
            const C2<TT> t2 = cmul(c8, cw);
            const C2<TT> w2 = {sp ? TT(1) : t2.x, sp ? TT(0) : t2.y};
            const C2<TT> w3 = {sp ? c8.x : -w0.y, sp ? c8.y : -w0.x};
            const C2<TT> p0 = {sp ? za[s][3].x : zb[s][3].x, sp ? za[s][3].y : zb[s][3].y};
            const C2<TT> p1 = {sp ? za[s][2].x : zb[s][2].x, sp ? za[s][2].y : zb[s][2].y};
            const C2<TT> p2 = {sp ? zb[s][0].x : za[s][3].x, sp ? zb[s][0].y : za[s][3].y};
            const C2<TT> p3 = {sp ? zb[s][3].x : za[s][2].x, sp ? zb[s][3].y : za[s][2].y};
            w64::split_pair<TT>(P, w0, ka, za[s][0], p0);
            w64::split_pair<TT>(P, w1, ka + 256, za[s][1], p1);
            w64::split_pair<TT>(P, w2, kb, zb[s][0], p2);
            w64::split_pair<TT>(P, w3, kb + 256, zb[s][1], p3);
            if (s == 1) {  // the special slot's fifth pair, k = 512 with itself; every other lane repeats its first pair
                const C2<TT> q = {sp ? zb[s][2].x : za[s][0].x, sp ? zb[s][2].y : za[s][0].y};
                const C2<TT> r = {sp ? zb[s][2].x : p0.x, sp ? zb[s][2].y : p0.y};
                const C2<TT> w4 = {sp ? TT(0) : w0.x, sp ? TT(-1) : w0.y};
                w64::split_pair<TT>(P, w4, sp ? 512 : ka, q, r);
            }
        }
        AUD_BENIGN_RACE_BEGIN();  // lanes 3..63 repeat lane 0's store (no LDS access under a lane condition)
        P[w64::kH + (lane < 3 ? lane : 0)] = TT(0);  // pad bins 1025..1027 of the last 4-bin chunk
        AUD_BENIGN_RACE_END();
    }
    wave_lds_fence();
    AUD_STAMP(7);
    wave_mel_steps<TT, 1, MAXS, true>(a, e, P, L::kHp, smem, it, item, sstep, lane);
    AUD_STAMP(8);
    AUD_STAMP_REAL(10);
    AUD_STAMP_FLUSH(a, wt, lane);
    wave_lds_fence();  // the plane is free for the next frame
}

template <typename TT, bool PCM16, int NW, int MAXS>
__global__ __launch_bounds__(64 * NW) void k_melspec_w64(const MelspecArgs a, const FastArgs e) {
    using L = w64::Layout<TT>;
    unsigned char* smem = dyn_lds();
    const int tid = int(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    BlobRegs<64 * NW> blob;
    blob_fetch<64 * NW>(e, tid, blob);
    blob_store<64 * NW>(e, smem, tid, blob);
    __syncthreads();  // the one barrier: tables visible to the workgroup's waves
    TT* xw = reinterpret_cast<TT*>(smem + e.xch_off + wave * L::kRegion);
    const int64_t total = int64_t(a.n_items) * a.T;  // one frame per wave tile
    const unsigned wg = tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap);
    const int64_t wt0 = (int64_t(wg) * NW + wave) * w64::kFPW;
#pragma unroll 1
    for (int i = 0; i < w64::kFPW; ++i)
        if (wt0 + i < total) w64_tile<TT, PCM16, MAXS>(a, e, smem, xw, lane, wt0 + i);
}

}  // namespace

// waves per workgroup: the waves of a workgroup share one LDS copy of the table blob (eight-wave workgroups were
// tried for the 17 KB float64 N = 400 regions: their one barrier cost a quarter of the wave's life)
// (the N = 2048 kernel's float64 tables + planes fill a CU with ONE workgroup of twelve waves = three per SIMD; two
// workgroups of six were measured to leave one of them waiting: six waves land 2-2-1-1 on the SIMDs and the second
// workgroup's pair does not fit beside the first's at 160 VGPRs)
static int wave_kernel_waves(int kind, int compute_dtype) { return kind == 4 && compute_dtype == AUD_F64 ? 12 : 4; }

bool melspec_wave_geometry(int kind, int N, WaveGeometry* g) {
    if (kind == 1 && N == w16::kN) {
        *g = WaveGeometry{64 / w16::kFW, 16, 16, w16::kM / 2 + 1};
        return true;
    }
    if (kind == 2 && N == w25::kN) {
        *g = WaveGeometry{64 / w25::kFW, 8, 25, w25::kM / 2 + 1};
        return true;
    }
    if (kind == 3 && N == w20::kN) {
        *g = WaveGeometry{10, w20::kLPF, 20, w20::kM / 2 + 1};
        return true;
    }
    if (kind == 4 && N == w64::kN) {
        *g = WaveGeometry{64, 64, 16, 0};  // one frame per wave: every lane is a filter group
        return true;
    }
    return false;
}

bool melspec_wave_finish(int kind, int compute_dtype, FastArgs* e) {
    const bool f64 = compute_dtype == AUD_F64;
    const size_t region = kind == 1   ? (f64 ? size_t(w16::Layout<double>::kRegion) : size_t(w16::Layout<float>::kRegion))
                          : kind == 2 ? (f64 ? size_t(w25::Layout<double>::kRegion) : size_t(w25::Layout<float>::kRegion))
                          : kind == 3 ? (f64 ? size_t(w20::Layout<double>::kRegion) : size_t(w20::Layout<float>::kRegion))
                                      : (f64 ? size_t(w64::Layout<double>::kRegion) : size_t(w64::Layout<float>::kRegion));
    const int nw = wave_kernel_waves(kind, compute_dtype);
    const size_t first = (size_t(e->blob_bytes) + 255) & ~size_t(255);
    const size_t total = first + size_t(nw) * region;
    if (total > 160 * 1024) return false;
    e->xch_off = int(first);
    e->p_off = int(first);
    e->lds_bytes = unsigned(total);
    e->waves = nw;
    e->max_wgs = 0;  // set by melspec_wave_prepare
    e->variant = 2;  // measured (profiles/r02f_ab_*): one tile per wave, first operands under the blob staging
    e->persistent = -1;  // automatic (launch_melspec_wave)
    e->direct = 1;
    e->ntile = 1;
    return true;
}

// the instantiation a plan runs: kind, compute type and the slot capacity of its epilogue (4 or 8)
typedef void (*wave_kernel_t)(const MelspecArgs, const FastArgs);
static wave_kernel_t wave_kernel(int kind, bool f64, int n_slots, int var) {
    const bool s8 = n_slots > 4;
    if (kind == 1) {
#define AUD_W16(TT, S) (var == 1 ? k_melspec_w16<TT, true, 4, S, 1> : var == 2 ? k_melspec_w16<TT, true, 4, S, 2> : var == 3 ? k_melspec_w16<TT, true, 4, S, 3> : k_melspec_w16<TT, true, 4, S, 0>)
        if (f64) return s8 ? AUD_W16(double, 8) : AUD_W16(double, 4);
        return s8 ? AUD_W16(float, 8) : AUD_W16(float, 4);
#undef AUD_W16
    }
    if (kind == 4) {
        if (f64) return s8 ? k_melspec_w64<double, true, 12, 8> : k_melspec_w64<double, true, 12, 4>;
        return s8 ? k_melspec_w64<float, true, 4, 8> : k_melspec_w64<float, true, 4, 4>;
    }
    if (kind == 3) {
#define AUD_W20(TT, S) (var == 3 ? k_melspec_w20<TT, true, 4, S, 3> : var == 0 ? k_melspec_w20<TT, true, 4, S, 0> : k_melspec_w20<TT, true, 4, S, 2>)
        if (f64) return s8 ? AUD_W20(double, 8) : AUD_W20(double, 4);
        return s8 ? AUD_W20(float, 8) : AUD_W20(float, 4);
#undef AUD_W20
    }
    if (f64) return var == 3 ? (s8 ? k_melspec_w25<double, true, 4, 8, 3> : k_melspec_w25<double, true, 4, 4, 3>)
                             : (s8 ? k_melspec_w25<double, true, 4, 8, 2> : k_melspec_w25<double, true, 4, 4, 2>);
#define AUD_W25(S) (var == 1 ? k_melspec_w25<float, true, 4, S, 1> : var == 2 ? k_melspec_w25<float, true, 4, S, 2> : var == 3 ? k_melspec_w25<float, true, 4, S, 3> : k_melspec_w25<float, true, 4, S, 0>)
    return s8 ? AUD_W25(8) : AUD_W25(4);
#undef AUD_W25
}

hipError_t melspec_wave_prepare(int kind, int compute_dtype, FastArgs* e) {
    if (e->n_slots > 8) return hipErrorInvalidValue;  // more filters per group than the epilogue has slots for
    const void* fn = reinterpret_cast<const void*>(wave_kernel(kind, compute_dtype == AUD_F64, e->n_slots, e->variant));
    if (e->lds_bytes > 64u * 1024u) {
        hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(e->lds_bytes));
        if (rc != hipSuccess) return rc;
    }
    // the persistent grid: as many workgroups as are resident at once (no grid-wide wait anywhere, so an
    // over-estimate only queues the surplus workgroups behind the first ones)
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc == hipSuccess) rc = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (rc == hipSuccess) rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * e->waves, e->lds_bytes);
    if (rc != hipSuccess) return rc;
    e->max_wgs = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
    e->wgs_per_cu = per_cu;
    return hipSuccess;
}

hipError_t launch_melspec_wave(int kind, const MelspecArgs& a, const FastArgs& e, int compute_dtype, hipStream_t st) {
    if (kind == 4) {  // one frame per wave tile, kFPW consecutive frames per wave, never persistent
        const int64_t frames = int64_t(a.n_items) * a.T;
        const int64_t per_wg = int64_t(e.waves) * w64::kFPW;
        const int64_t wgs64 = (frames + per_wg - 1) / per_wg;
        if (wgs64 > 0x7FFFFFFF) return hipErrorInvalidValue;
        hipLaunchKernelGGL(wave_kernel(kind, compute_dtype == AUD_F64, e.n_slots, e.variant), dim3(unsigned(wgs64)),
                           dim3(64 * e.waves), e.lds_bytes, st, a, e);
        return hipGetLastError();
    }
    const int fw = kind == 1 ? w16::kFW : kind == 2 ? w25::kFW : w20::kFW;
    const int64_t tiles = (a.T + fw - 1) / fw;
    const int64_t waves = int64_t(a.n_items) * tiles;
    const int nw = e.waves;
    int64_t wgs = (waves + nw - 1) / nw;
    // persistent grid (every wave walks tiles wt, wt + stride, ...): for the 16 x 16 and 25 x 8 kernels it pays once a
    // launch holds several rounds of resident waves (profiles/r02i_ab_*, r02z_ab_*: B = 4096 12-20 % faster, B = 256 up
    // to 20 % slower: static tile assignment); the 20 x 10 kernel measures the same or better with one tile per wave at
    // every size (r02z_ab_*), so its automatic choice is never persistent
    const bool dynamic = e.persistent == 2 && a.queue != nullptr;  // persistent grid + dynamic tile queue
    const bool persistent =
        dynamic || e.persistent == 1 || (e.persistent < 0 && kind != 3 && wgs >= 4 * int64_t(e.max_wgs));
    if (persistent && e.max_wgs > 0 && wgs > e.max_wgs) wgs = e.max_wgs;
    const dim3 grid{unsigned(wgs)}, blk(64 * nw);
    hipLaunchKernelGGL(wave_kernel(kind, compute_dtype == AUD_F64, e.n_slots, dynamic ? 3 : e.variant), grid, blk, e.lds_bytes, st,
                       a, e);
    return hipGetLastError();
}

}  // namespace aud
