// N = 2048 wave-autonomous kernel "w64x16", one frame per wave (see melspec_wave.hip for what the wave kernels have in common).
// Reference semantics: sound/sndenv.go:438-478, dft/dft.go:53-85, mel/mel.go:120-153.
#include "wave_common.h"

namespace aud {

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
//   power (x 4, scaled, float32) to the wave's plane, then the shared slot-uniform mel epilogue with 64 filter groups (one per lane).
// A wave walks kFPW consecutive frames so that a workgroup's one table staging serves NW x kFPW frames.
namespace w64 {
constexpr int kH = 1025, kHp = 1028;
constexpr int kRow = 68;           // transpose rows: 64 (= 16 x 4) elements + 4
constexpr int kPlane = 16 * kRow;  // 1088 elements, also holds the power row
template <typename TT>
struct Layout {
    static constexpr int kRegion = kPlane * int(sizeof(TT));  // bytes per wave
};
}  // namespace w64

namespace {

// the operands of frame (item, sstep) (one frame per wave tile): requested into `raw`, nothing awaited.  item and sstep
// are SCALAR values (the caller keeps them in scalar registers): the buffer descriptor must be wave-uniform to the
// compiler, or every load is wrapped in a loop over the descriptor's values
template <int SRC>
__device__ __forceinline__ void w64_issue(const MelspecArgs& a, int lane, int item, int sstep, PairRaw<16>& raw) {
    const aud_item it = a.items[item];
    const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
    const SampleWindow<SRC> win = sample_window<SRC>(a, it, start, w64::kN);
    pairs_issue<SRC, 16, 64>(win, start + 2 * lane, raw);
}

// One frame.  `raw` holds its operands (w64_issue); the NEXT frame's are requested into it before this frame's epilogue
// (the load phase was 38 % of a wave's life without it: profiles/r05e_stamps_n46.44_f64_b64.txt).
template <typename TT, int SRC, int MAXS>
__device__ __forceinline__ void w64_tile(const MelspecArgs& a, const WaveArgs& e, unsigned char* smem, unsigned char* region,
                                         int lane_in, int item, int sstep, bool more, PairRaw<16>& raw, float* stash, int stash_i) {
    TT* xw = reinterpret_cast<TT*>(region);
    int lane = lane_in;  // opaque per frame: otherwise the compiler hoists what only depends on it out of the frame loop
    asm volatile("" : "+v"(lane));
    const aud_item it = a.items[item];
    const int64_t wt = int64_t(item) * a.T + sstep;  // (the stamps' index)
    (void)wt;
    AUD_STAMP_DECL;
    AUD_STAMP(0);
    AUD_STAMP_REAL(9);
    C2<TT> v[16];
    TT amax;
    if constexpr (sizeof(TT) == 4) w64_issue<SRC>(a, lane, item, sstep, raw);  // (float32 plans: no prefetch, see below)
    pairs_take<TT, SRC, 16, 64>(a, it, int64_t(it.start0) + int64_t(a.S) * (sstep - a.border) + 2 * lane, true, raw, v, amax);
    // the frame is the whole wave: its scale is a wave-wide maximum (six data-parallel moves, no LDS)
    int sc = 0;
    if constexpr (sizeof(TT) == 8) {
        const int ex = wave_max_i32(amax_exponent<TT>(amax));
        sc = scale_of_exponent(ex);
    }
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
    float* P = reinterpret_cast<float*>(region);  // [kHp]
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
            split_pair<TT>(P, w0, w64::kM, ka, za[s][0], p0, sc);
            split_pair<TT>(P, w1, w64::kM, ka + 256, za[s][1], p1, sc);
            split_pair<TT>(P, w2, w64::kM, kb, zb[s][0], p2, sc);
            split_pair<TT>(P, w3, w64::kM, kb + 256, zb[s][1], p3, sc);
            if (s == 1) {
                // bin 512 pairs with itself: X[512] = conj Z[512], so 4 |X|^2 = 4 |Z|^2 -- Z[512] is element k3 = 2 of the
                // special slot's b column.  Every other lane stores a zero into the last pad bin (no LDS access under a
                // lane condition; the pad bins are zeroed below anyway)
                const C2<TT> z512 = zb[s][2];
                const float p512 = scaled_power(TT(4) * mad(z512.x, z512.x, z512.y * z512.y), sc);
                AUD_BENIGN_RACE_BEGIN();
                P[sp ? 512 : w64::kH + 2] = sp ? p512 : 0.f;
                AUD_BENIGN_RACE_END();
            }
        }
        AUD_BENIGN_RACE_BEGIN();  // lanes 3..63 repeat lane 0's store (no LDS access under a lane condition)
        P[w64::kH + (lane < 3 ? lane : 0)] = 0.f;  // pad bins 1025..1027 of the last 4-bin chunk
        AUD_BENIGN_RACE_END();
    }
    wave_lds_fence();
    AUD_STAMP(7);
    // the next frame's operands land during the epilogue (a fifth of the frame's time); requested here, not at the top,
    // because the FFT passes are where the registers run out.  float64 plans only: their three waves per SIMD leave the
    // 32 registers free, in float32 plans they would cost a wave per SIMD
    if constexpr (sizeof(TT) == 8)
        if (more) {  // the next frame of the wave: the next step, or step 0 of the next item
            const bool wrap = sstep + 1 == a.T;
            w64_issue<SRC>(a, lane, wrap ? item + 1 : item, wrap ? 0 : sstep + 1, raw);
        }
    wave_mel_epilogue<TT, 1, MAXS, true>(a, e, P, w64::kHp, smem, sc, it, item, sstep, lane, nullptr, nullptr, stash, stash_i);
    AUD_STAMP(8);
    AUD_STAMP_REAL(10);
    AUD_STAMP_FLUSH(a, wt, lane);
    wave_lds_fence();  // the plane is free for the next frame
}

template <typename TT, int SRC, int NW, int MAXS>
__global__ __launch_bounds__(64 * NW) void k_melspec_w64(const aud_item*, unsigned, unsigned, unsigned, int, const void* blob_ptr, int blob_bytes, unsigned n_wgs,
                   int xcd_remap, const MelspecArgs a, const WaveArgs e) {
    using L = w64::Layout<TT>;
    unsigned char* smem = dyn_lds();
    const int tid = int(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    BlobRegs<64 * NW> blob;
    blob_fetch<64 * NW>(blob_ptr, blob_bytes, tid, blob);
    blob_store<64 * NW>(e, smem, tid, blob);
    __syncthreads();  // the one barrier: tables visible to the workgroup's waves
    unsigned char* region = smem + e.xch_off + wave * L::kRegion;
    const int64_t total = int64_t(a.n_items) * a.T;  // one frame per wave tile
    const unsigned wg = tile_of_workgroup(blockIdx.x, n_wgs, xcd_remap);
    const int64_t wt0 = (int64_t(wg) * NW + wave) * w64::kFPW;
    // (item, step) of the wave's frames, kept on the scalar unit
    int item = __builtin_amdgcn_readfirstlane(int(tile_div(a, unsigned(wt0))));  // (a.tiles == a.T here; wt0 < 2^31)
    int sstep = __builtin_amdgcn_readfirstlane(int(wt0 - int64_t(item) * a.T));
    PairRaw<16> raw;
    if constexpr (sizeof(TT) == 8)
        if (wt0 < total) w64_issue<SRC>(a, lane, item, sstep, raw);
    // the wave's four frames are steps t0 .. t0 + 3 of ONE item whenever T is a multiple of 4 (wt0 is): their mel values then
    // leave as 16-byte [filter][4 steps] pieces from an LDS stash instead of 4 x n_filters scattered 4-byte stores
    static_assert(w64::kFPW == 4, "wave_mel_flush4 writes four steps");
    const bool quad = e.stash_off >= 0 && (a.T & 3) == 0 && wt0 + w64::kFPW <= total && (reinterpret_cast<uintptr_t>(a.mel) & 15) == 0;
    float* stash = quad ? reinterpret_cast<float*>(smem + e.stash_off) + size_t(wave) * e.n_slots * 64 * 4 : nullptr;
    const int item0 = item, t0 = sstep;
#pragma unroll 1
    for (int i = 0; i < w64::kFPW; ++i) {
        if (wt0 + i >= total) break;
        w64_tile<TT, SRC, MAXS>(a, e, smem, region, lane, item, sstep, i + 1 < w64::kFPW && wt0 + i + 1 < total, raw, stash, i);
        if (++sstep == a.T) {
            sstep = 0;
            ++item;
        }
    }
    if (quad) {
        wave_lds_fence();
        wave_mel_flush4<MAXS>(a, e, smem, stash, item0, t0, lane);
    }
}

}  // namespace

size_t w64_region_bytes(bool f64) { return f64 ? size_t(w64::Layout<double>::kRegion) : size_t(w64::Layout<float>::kRegion); }

#define AUD_W64_PICK(TT, NW)                                                                                \
    (sig_dtype == AUD_F64   ? (s8 ? k_melspec_w64<TT, AUD_F64, NW, 8> : k_melspec_w64<TT, AUD_F64, NW, 4>)   \
     : sig_dtype == AUD_I16 ? (s8 ? k_melspec_w64<TT, AUD_I16, NW, 8> : k_melspec_w64<TT, AUD_I16, NW, 4>)   \
                            : (s8 ? k_melspec_w64<TT, AUD_F32, NW, 8> : k_melspec_w64<TT, AUD_F32, NW, 4>))
wave_kernel_t w64_kernel(bool f64, int sig_dtype, int n_slots) {
    const bool s8 = n_slots > 4;
    return f64 ? AUD_W64_PICK(double, 12) : AUD_W64_PICK(float, 4);
}
#undef AUD_W64_PICK

}  // namespace aud
