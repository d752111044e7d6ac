// N = 400 wave-autonomous kernel "w20x10" (see melspec_wave.hip for what the wave kernels have in common).
// Reference semantics: sound/sndenv.go:438-478, dft/dft.go:53-85, mel/mel.go:120-153.
#include "wave_common.h"
#include "gabor_tile.h"

namespace aud {

// ================================================================================================
// N = 400: 200-point complex FFT as 20 x 10, 10 lanes per frame, 6 frames per wave (60 of 64 lanes)
// ================================================================================================
// A lane takes 20 points in pass A and then exactly one row PAIR (r, 20 - r) of 10 columns: every lane has both halves of
// its ten split pairs locally (lane 0 takes the two self-paired rows 0 and 10).  20 = 4 x 5 and 10 = 2 x 5 have coprime
// factors, so both small DFTs are prime-factor transforms without inner twiddles.  128 registers and 5.9 KB of LDS per
// wave in float64: four waves per SIMD.
namespace w20 {
constexpr int kH = 201;  // power bins (6 frames per wave, 10 lanes per frame: wave_common.h)
constexpr int kHp = 204;  // P row pitch in floats: 51 16-byte pieces (odd)
template <typename TT>
struct Layout {
    // The transposes move HALF the rows at a time (rows 0..9, then rows 10..19: a lane's row pair (j, 20 - j) has one row
    // in each half, lane 0's (0, 10) too), one component at a time: a quarter of a frame's complex data is in LDS at any
    // moment, 5.9 KB per wave in float64 -- with the tables 31 KB per workgroup, so that LDS admits the four workgroups
    // per CU the register budget (128) allows.  Scalar rows of 10 (+2 pad in float32: 16-byte rows); pitches from a
    // search over the hardware's lane groups (tools/lds_bank_model.py): float64 column stores conflict-free (frame pitch
    // = 10 mod 16 elements: the ten-lane frames tile the 16-lane store groups), row reads 2-way -- ten-lane frames
    // against ds_read_b128's 16-lane groups cannot be conflict-free
    static constexpr int kRow = (sizeof(TT) == 4) ? 12 : 10;
    static constexpr int kFrame = (sizeof(TT) == 4) ? 120 : 122;
    static constexpr int kXchBytes = kFW * kFrame * int(sizeof(TT));
    static constexpr int kPBytes = kFW * kHp * 4;
    static constexpr int kExpOff = ((kXchBytes > kPBytes ? kXchBytes : kPBytes) + 15) & ~15;  // the frames' scale words
    static constexpr int kRegion = kExpOff + 32;  // bytes per wave
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
// Real-FFT split + power of a lane's row pair.  Lanes 1..9: rows (j, 20 - j): k = j + 20 c pairs with (row 20 - j, column
// 9 - c) and vice versa, c = 0..4; lane 0: row 0: k = 20 c pairs with column 10 - c of the same row (c = 0: DC + Nyquist;
// c = 5: itself), row 10: k = 10 + 20 c pairs with column 9 - c of the same row.  One code path, partners selected by value.
template <typename TT, bool SCALED>
__device__ __forceinline__ void split_rows(float* P, const C2<TT>* tws, const C2<TT> (&za)[10], const C2<TT> (&zb)[10], int j,
                                           int ra, int rb, int sc) {
    const bool self = j == 0;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const C2<TT> pa = za[(10 - c) % 10], pb = zb[9 - c], pc = za[9 - c];
        const C2<TT> b_first = {self ? pa.x : pb.x, self ? pa.y : pb.y};
        const C2<TT> b_second = {self ? pb.x : pc.x, self ? pb.y : pc.y};
        split_pair<TT, SCALED>(P, tws[ra + 20 * c], kM, ra + 20 * c, za[c], b_first, sc);
        split_pair<TT, SCALED>(P, tws[rb + 20 * c], kM, rb + 20 * c, zb[c], b_second, sc);
    }
    // bin 100 (row 0, column 5: lane 0) pairs with itself: X[100] = conj Z[100], so FOUR times its power is 4 |Z[100]|^2
    // directly -- the same bits the split gives (its operands are 2 Re Z and -2 Im Z: an exact factor).  Every other lane
    // parks the value of ITS column 5 in pad bin 203, which the next store zeroes (a wave's LDS stores execute in order)
    P[self ? 100 : kH + 2] = scaled_power<SCALED>(TT(4) * mad(za[5].x, za[5].x, za[5].y * za[5].y), sc);
    P[kH + (j < 3 ? j : 0)] = 0.f;  // pad bins 201..203 of the last 4-bin chunk
}
}  // namespace w20

namespace {

// A wave tile of six frames from its converted operands to the float32 power spectrum in the wave's LDS region (passes A and
// B, the transposes between them, the real-FFT split): what both kernels below run per tile.  Returns the scale of the frame
// this lane TRANSFORMED (lane / 10); the epilogue's lanes own other frames and read theirs from the region's scale words.
template <typename TT, int SRC>
__device__ __forceinline__ void w20_tile_front(const MelspecArgs& a, const WaveArgs& e, unsigned char* smem, unsigned char* region,
                                               const aud_item& it, int t0, int f, int j, int64_t pos0,
                                               const PairRaw<20>& raw AUD_STAMP_PARAM) {
    using L = w20::Layout<TT>;
    C2<TT> v[20];
    TT amax;
    pairs_take<TT, SRC, 20, 10>(a, it, pos0, t0 + f < a.T, raw, v, amax);

    TT* xw = reinterpret_cast<TT*>(region);
    int* exps = reinterpret_cast<int*>(region + L::kExpOff);
    const C2<TT>* twa = reinterpret_cast<const C2<TT>*>(smem + e.twa_off);  // W_400^(2 j k1) at [(k1 - 1) 10 + j]
    const C2<TT>* tws = reinterpret_cast<const C2<TT>*>(smem + e.tws_off);  // W_400^k, k <= 100
    const int sc = frame_scale<TT>(exps + f, amax);
    AUD_STAMP(3);

    // ---- pass A: 20-point DFT over n1, twiddle W_200^(j k1) = W_400^(2 j k1) -------------------------------
    SmallDft<TT, 20>::run(v, nullptr, 0);
#pragma unroll
    for (int k1 = 1; k1 < 20; ++k1) v[k1] = cmul(v[k1], twa[(k1 - 1) * w20::kLPF + j]);
    AUD_STAMP(4);

    // ---- transpose, half the rows at a time (real parts, then imaginary parts): element (row k1, column n2 = j) of frame f;
    // afterwards the lane holds the row pair (j, 20 - j) -- lane 0 the self-paired rows 0 and 10 -- with all ten columns
    // of each.  Half A = rows 0..9 (every lane's first row), half B = rows 10..19 stored at row k1 - 10 (its second row).
    TT* col = xw + f * L::kFrame + j;
    constexpr int cstep = L::kRow;
    const TT* row_a = xw + f * L::kFrame + j * L::kRow;
    const TT* row_b = xw + f * L::kFrame + (j == 0 ? 0 : 10 - j) * L::kRow;
    const int ra = j, rb = j == 0 ? 10 : 20 - j;
    C2<TT> za[10], zb[10];
#pragma unroll
    for (int part = 0; part < 2; ++part) {  // 0: real parts, 1: imaginary parts
        AUD_BENIGN_RACE_BEGIN();
#pragma unroll
        for (int k1 = 0; k1 < 10; ++k1) col[k1 * cstep] = part ? v[k1].y : v[k1].x;
        AUD_BENIGN_RACE_END();
        wave_lds_fence();
        w20::read_row10<TT>(row_a, za, part != 0);
        wave_lds_fence();
        AUD_BENIGN_RACE_BEGIN();
#pragma unroll
        for (int k1 = 0; k1 < 10; ++k1) col[k1 * cstep] = part ? v[10 + k1].y : v[10 + k1].x;
        AUD_BENIGN_RACE_END();
        wave_lds_fence();
        w20::read_row10<TT>(row_b, zb, part != 0);
        wave_lds_fence();  // (after the last one: every row has been read, the region may take the power spectrum)
    }
    AUD_STAMP(5);

    // ---- pass B: 10-point DFT over n2 of both rows: Z[k1 + 20 k2] -----------------------------------------------
    SmallDft<TT, 10>::run(za, nullptr, 0);
    SmallDft<TT, 10>::run(zb, nullptr, 0);
    AUD_STAMP(6);

    // ---- real-FFT split + power: the partner of Z[k1 + 20 k2] is element (20 - k1, 9 - k2); pairs are evaluated from
    // their k <= 100 side (A = Z[k], B = Z[200 - k])
    float* P = reinterpret_cast<float*>(region) + f * w20::kHp;  // [6][kHp]
    AUD_BENIGN_RACE_BEGIN();
    // (frames at ordinary levels carry scale 0, device_common.h scale_of_exponent: a wave of them skips every bin's ldexp)
    if (sizeof(TT) == 4 || __builtin_amdgcn_ballot_w64(sc != 0) == 0) w20::split_rows<TT, false>(P, tws, za, zb, j, ra, rb, sc);
    else w20::split_rows<TT, true>(P, tws, za, zb, j, ra, rb, sc);
    AUD_BENIGN_RACE_END();
    wave_lds_fence();
    AUD_STAMP(7);
}

// float64: four waves per SIMD (122 registers are the kernel's working set: DESIGN.md 4.1); float32: five
template <typename TT, int SRC, int NW, int MAXS>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(sizeof(TT) == 8 ? 4 : 5, sizeof(TT) == 8 ? 4 : 5)))
void k_melspec_w20(const aud_item* items, unsigned total, unsigned tiles, unsigned tile_mul, int tile_shift, const void* blob_ptr,
                   int blob_bytes, unsigned n_wgs, int xcd_remap, const MelspecArgs a, const WaveArgs e) {
    using L = w20::Layout<TT>;
    unsigned char* smem = dyn_lds();
    const int tid = int(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    BlobRegs<64 * NW> blob;
    blob_fetch<64 * NW>(blob_ptr, blob_bytes, tid, blob);

    const unsigned wg = tile_of_workgroup(blockIdx.x, n_wgs, xcd_remap);
    const unsigned wt = wg * NW + wave;
    const bool active = wt < total;
    const int item = active ? int(tile_div(tile_mul, tile_shift, wt)) : 0;  // (total = n_items x tiles < 2^31: launch_melspec_wave checks)
    const int t0 = active ? int(wt - unsigned(item) * tiles) * w20::kFW : 0;
    const aud_item it = items[item];
    // lanes 60..63 have no frame of their own: they SHADOW lanes 50..53 (same frame, same column) through the whole FFT --
    // same loads, same arithmetic, same values stored to the same LDS addresses -- so that no LDS access sits under a
    // lane condition
    const bool own = lane < w20::kFW * w20::kLPF;
    const int f = own ? lane / w20::kLPF : w20::kFW - 1;
    const int j = own ? lane - f * w20::kLPF : lane - w20::kFW * w20::kLPF;
    AUD_STAMP_DECL;
    AUD_STAMP(0);
    AUD_STAMP_REAL(9);

    // pass A operands: z[10 n1 + j] = (x[20 n1 + 2j], x[.. + 1]), n1 = 0..19
    const int64_t first_start = int64_t(it.start0) + int64_t(a.S) * (t0 - a.border);
    const SampleWindow<SRC> win = sample_window<SRC>(a, it, first_start, a.S * (w20::kFW - 1) + w20::kN);
    const int64_t pos0 = first_start + int64_t(a.S) * f + 2 * j;
    PairRaw<20> raw;
    if (active) pairs_issue<SRC, 20, 10>(win, pos0, raw);

    blob_store<64 * NW>(e, smem, tid, blob);
    __syncthreads();  // the one barrier: tables visible to the workgroup's waves
    if (!active) return;

    unsigned char* region = smem + e.xch_off + wave * L::kRegion;
    w20_tile_front<TT, SRC>(a, e, smem, region, it, t0, f, j, pos0, raw AUD_STAMP_ARG);

    // ---- optional spectrum outputs and the mel reduction: 6 frames x 10 filter groups on this wave ---------------
    const int* exps = reinterpret_cast<const int*>(region + L::kExpOff);
    wave_mel_epilogue_pick<TT, w20::kFW, MAXS>(a, e, reinterpret_cast<const float*>(region), w20::kHp, smem,
                                          sizeof(TT) == 8 ? frame_scale_of(exps + lane % w20::kFW) : 0, it, item, t0, lane, exps);
    AUD_STAMP(8);
    AUD_STAMP_REAL(10);
    AUD_STAMP_FLUSH(a, wt, lane);
}

// ================================================================================================
// The workgroup-per-ITEM variant: one workgroup takes every frame of one work item, its NW waves walk the item's tiles
// (wave w: tiles w, w + NW, ...), and every mel value is also kept in the item's [nf][T] float32 matrix in LDS.  Behind one workgroup barrier the same waves run
// agabor.Convolve on that matrix (gabor_tile.h): SndEnv.ProcessSegment + ApplyGabor's convolution as ONE launch whose
// memory traffic is the samples in and the two feature tensors out (sound/sndenv.go:342-359, :481-497).
// The tables are staged once per 18 tiles instead of once per 4, at the price of whole-item granularity: 104 frames are
// 18 tiles on five waves (four rounds, 90 % of the wave slots busy).
// ================================================================================================
template <typename TT, int SRC, int NW, int MAXS>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(sizeof(TT) == 8 ? 4 : 5, sizeof(TT) == 8 ? 4 : 5)))
void k_melspec_w20_item(const aud_item* items, unsigned n_items, unsigned tiles, const void* blob_ptr, int blob_bytes,
                        const float* __restrict__ k32, const MelspecArgs a, const WaveArgs e, const ItemArgs g) {
    using L = w20::Layout<TT>;
    unsigned char* smem = dyn_lds();
    const int tid = int(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    BlobRegs<64 * NW> blob;
    blob_fetch<64 * NW>(blob_ptr, blob_bytes, tid, blob);

    const int item = int(blockIdx.x);  // (grid = n_items)
    const aud_item it = items[item];
    AUD_STAMP_DECL;
    AUD_STAMP(0);
    AUD_STAMP_REAL(9);

    // a lane's frame and column inside a tile (lanes 60..63 shadow lanes 50..53, as in k_melspec_w20).  Derived from a
    // LAUNDERED lane id once per tile: everything per-lane the tile body computes from them (LDS addresses of the transposes,
    // twiddle and weight rows, output columns) is then per-tile work, as in the one-tile kernel -- left loop-invariant the
    // compiler hoists ~45 registers of such addresses across the loop and spills the butterflies instead
    auto frame_of = [](int ln) { return ln < w20::kFW * w20::kLPF ? ln / w20::kLPF : w20::kFW - 1; };
    auto column_of = [](int ln, int fr) { return ln < w20::kFW * w20::kLPF ? ln - fr * w20::kLPF : ln - w20::kFW * w20::kLPF; };
    // the tile's operands: frame f of tile `tile` starts at start0 + S (6 tile + f - border)
    auto tile_pos0 = [&](int tile, int fr, int cl) {
        return int64_t(it.start0) + int64_t(a.S) * (tile * w20::kFW - a.border + fr) + 2 * cl;
    };
    auto tile_window = [&](int tile) {
        return sample_window<SRC>(a, it, int64_t(it.start0) + int64_t(a.S) * (tile * w20::kFW - a.border),
                                  a.S * (w20::kFW - 1) + w20::kN);
    };
    int tile = wave;
    PairRaw<20> raw;
    {
        const int f0 = frame_of(lane), j0 = column_of(lane, f0);
        if (tile < int(tiles)) pairs_issue<SRC, 20, 10>(tile_window(tile), tile_pos0(tile, f0, j0), raw);
    }

    blob_store<64 * NW>(e, smem, tid, blob);
    __syncthreads();  // tables visible to the workgroup's waves

    float* melL = reinterpret_cast<float*>(smem + g.mel_off);
    while (tile < int(tiles)) {  // wave-uniform
        int ln = lane;
        asm volatile("" : "+v"(ln));  // (see above)
        const int f = frame_of(ln), j = column_of(ln, f);
        unsigned char* region = smem + e.xch_off + wave * L::kRegion;
        const int* exps = reinterpret_cast<const int*>(region + L::kExpOff);
        const int t0 = tile * w20::kFW;
        w20_tile_front<TT, SRC>(a, e, smem, region, it, t0, f, j, tile_pos0(tile, f, j), raw AUD_STAMP_ARG);
        const int next = tile + NW;
        if (g.nG > 0)
            wave_mel_epilogue_pick<TT, w20::kFW, MAXS, true>(a, e, reinterpret_cast<const float*>(region), w20::kHp, smem,
                                                        sizeof(TT) == 8 ? frame_scale_of(exps + ln % w20::kFW) : 0, it, item, t0,
                                                        ln, exps, melL);
        else
            wave_mel_epilogue_pick<TT, w20::kFW, MAXS, false>(a, e, reinterpret_cast<const float*>(region), w20::kHp, smem,
                                                         sizeof(TT) == 8 ? frame_scale_of(exps + ln % w20::kFW) : 0, it, item, t0,
                                                         ln, exps);
        wave_lds_fence();  // the region is free for the next tile's transposes
        if (next < int(tiles)) pairs_issue<SRC, 20, 10>(tile_window(next), tile_pos0(next, f, j), raw);
        tile = next;
    }
    AUD_STAMP(8);
    if (g.nG <= 0) return;  // uniform: the mel-only item kernel
    __syncthreads();  // the item's mel matrix is complete
    if (g.SX == 9 && g.SY == 9) gabor_from_lds<TT, 9, 9>(g, melL, k32, a.T, item, wave, NW, lane);  // processspeech.go:226-253
    else gabor_from_lds<TT, 0, 0>(g, melL, k32, a.T, item, wave, NW, lane);
    AUD_STAMP_REAL(10);
    AUD_STAMP_FLUSH(a, unsigned(item) * NW + wave, lane);
}

}  // namespace

size_t w20_region_bytes(bool f64) { return f64 ? size_t(w20::Layout<double>::kRegion) : size_t(w20::Layout<float>::kRegion); }

#ifdef AUD_W20_QUICK  // resource experiments: one instantiation of each kernel (tools, never the product build)
wave_kernel_t w20_kernel(bool, int, int) { return k_melspec_w20<double, AUD_F32, 4, 4>; }
item_kernel_t w20_item_kernel(bool, int, int, int) { return k_melspec_w20_item<double, AUD_F32, 5, 4>; }
#else
#define AUD_W20_NW 4
#define AUD_W20_PICK(TT)                                                                                  \
    (sig_dtype == AUD_F64   ? (s8 ? k_melspec_w20<TT, AUD_F64, AUD_W20_NW, 8> : k_melspec_w20<TT, AUD_F64, AUD_W20_NW, 4>)   \
     : sig_dtype == AUD_I16 ? (s8 ? k_melspec_w20<TT, AUD_I16, AUD_W20_NW, 8> : k_melspec_w20<TT, AUD_I16, AUD_W20_NW, 4>)   \
                            : (s8 ? k_melspec_w20<TT, AUD_F32, AUD_W20_NW, 8> : k_melspec_w20<TT, AUD_F32, AUD_W20_NW, 4>))
wave_kernel_t w20_kernel(bool f64, int sig_dtype, int n_slots) {
    const bool s8 = n_slots > 4;
    return f64 ? AUD_W20_PICK(double) : AUD_W20_PICK(float);
}
#undef AUD_W20_PICK

#define AUD_W20_ITEM_PICK(TT, NW)                                                                                    \
    (sig_dtype == AUD_F64   ? (s8 ? k_melspec_w20_item<TT, AUD_F64, NW, 8> : k_melspec_w20_item<TT, AUD_F64, NW, 4>)   \
     : sig_dtype == AUD_I16 ? (s8 ? k_melspec_w20_item<TT, AUD_I16, NW, 8> : k_melspec_w20_item<TT, AUD_I16, NW, 4>)   \
                            : (s8 ? k_melspec_w20_item<TT, AUD_F32, NW, 8> : k_melspec_w20_item<TT, AUD_F32, NW, 4>))
item_kernel_t w20_item_kernel(bool f64, int sig_dtype, int n_slots, int waves) {
    const bool s8 = n_slots > 4;
    if (waves == 5) return f64 ? AUD_W20_ITEM_PICK(double, 5) : AUD_W20_ITEM_PICK(float, 5);
    return nullptr;
}
#undef AUD_W20_ITEM_PICK
#endif

}  // namespace aud
