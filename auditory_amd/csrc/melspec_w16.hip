// N = 512 wave-autonomous kernel "w16x16" (see melspec_wave.hip for what the wave kernels have in common).
// Reference semantics: sound/sndenv.go:438-478, dft/dft.go:53-85, mel/mel.go:120-153.
#include "wave_common.h"

namespace aud {

// ================================================================================================
// N = 512: 256-point complex FFT as 16 x 16, 16 lanes per frame, 4 frames per wave
// ================================================================================================
namespace w16 {
constexpr int kH = 257;   // power bins
constexpr int kHp = 260;  // P row pitch in floats: 65 16-byte pieces (odd: the frames of one chunk read spread over the banks)
template <typename TT>
struct Layout {
    // scalar transpose rows of 16 + pad: 20 floats = 5 slots, 18 doubles = 9 slots (odd)
    static constexpr int kRow = (sizeof(TT) == 4) ? 20 : 18;
    static constexpr int kFrame = 16 * kRow;                   // 80 / 144 slots: a multiple of 16
    static constexpr int kXchBytes = kFW * kFrame * int(sizeof(TT));
    static constexpr int kPBytes = kFW * kHp * 4;
    static constexpr int kExpOff = ((kXchBytes > kPBytes ? kXchBytes : kPBytes) + 15) & ~15;  // the frames' scale words
    static constexpr int kRegion = kExpOff + 32;               // bytes per wave
};
}  // namespace w16

namespace {

template <typename TT, int SRC, int NW, int MAXS>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(sizeof(TT) == 8 ? 3 : 5, sizeof(TT) == 8 ? 3 : 5)))
void k_melspec_w16(const aud_item* items, unsigned total, unsigned tiles, unsigned tile_mul, int tile_shift, const void* blob_ptr,
                   int blob_bytes, unsigned n_wgs, int xcd_remap, const MelspecArgs a, const WaveArgs e) {
    using L = w16::Layout<TT>;
    unsigned char* smem = dyn_lds();
    const int tid = int(threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: item record and addresses on the SALU
    const int lane = tid & 63;

    // the workgroup's tables: requested before anything else so that a counted wait can pick them out
    BlobRegs<64 * NW> blob;
    blob_fetch<64 * NW>(blob_ptr, blob_bytes, tid, blob);

    const unsigned wg = tile_of_workgroup(blockIdx.x, n_wgs, xcd_remap);
    const unsigned wt = wg * NW + wave;
    const bool active = wt < total;
    const int item = active ? int(tile_div(tile_mul, tile_shift, wt)) : 0;  // (total = n_items x tiles < 2^31: launch_melspec_wave checks)
    const int t0 = active ? int(wt - unsigned(item) * tiles) * w16::kFW : 0;
    const aud_item it = items[item];
    const int f = lane >> 4;   // frame within the wave
    const int j = lane & 15;   // lane within the frame's 16-lane group
    AUD_STAMP_DECL;
    AUD_STAMP(0);
    AUD_STAMP_REAL(9);

    // pass 1 operands: z[16 n1 + j] = (x[32 n1 + 2j], x[.. + 1])
    const int64_t first_start = int64_t(it.start0) + int64_t(a.S) * (t0 - a.border);
    const SampleWindow<SRC> win = sample_window<SRC>(a, it, first_start, a.S * (w16::kFW - 1) + w16::kN);
    const int64_t pos0 = first_start + int64_t(a.S) * f + 2 * j;
    PairRaw<16> raw;
    if (active) pairs_issue<SRC, 16, 16>(win, pos0, raw);

    blob_store<64 * NW>(e, smem, tid, blob);  // waits for the blob loads only: the operands stay in flight
    __syncthreads();                          // the one barrier: tables visible to the workgroup's waves
    if (!active) return;

    unsigned char* region = smem + e.xch_off + wave * L::kRegion;  // this wave's region
    TT* xw = reinterpret_cast<TT*>(region);
    int* exps = reinterpret_cast<int*>(region + L::kExpOff);
    const C2<TT>* twa = reinterpret_cast<const C2<TT>*>(smem + e.twa_off);  // W_512^(2 j k1) at [(k1 - 1) 16 + j]
    const C2<TT>* tws = reinterpret_cast<const C2<TT>*>(smem + e.tws_off);  // W_512^k, k <= 128

    C2<TT> v[16];
    TT amax;
    pairs_take<TT, SRC, 16, 16>(a, it, pos0, t0 + f < a.T, raw, v, amax);
    const int sc = frame_scale<TT>(exps + f, amax);
    AUD_STAMP(3);

    // ---- pass 1: 16-point DFT over n1, twiddle -----------------------------------------------------------
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

    // ---- real-FFT split + power --------------------------------------------------------------------------
    // For k = j + 16 q (q = 0..7) the partner Z[256 - k] sits in lane (16 - j) & 15, register 15 - q (lane 0
    // pairs with itself: register (16 - q) & 15).  X[k] = (E + T)/2, X[256-k] = conj(E - T)/2 with
    // E = Z[k] + conj Z[256-k], T = -i W_512^k (Z[k] - conj Z[256-k]).
    float* Pw = reinterpret_cast<float*>(region);  // [4][kHp]
    float* P = Pw + f * w16::kHp;
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
            P[k] = scaled_power(mad(xr, xr, xi * xi), sc);  // FOUR times the power (the 1/4 lives in the mel weights)
            if (k != 0) P[w16::kM - k] = scaled_power(mad(yr, yr, yi * yi), sc);
            else P[w16::kM] = scaled_power(mad(yr, yr, yi * yi), sc);  // k = 0 also yields the Nyquist bin
        }
        // k = 128 (lane 0, register 8) pairs with itself: X[128] = conj(Z[128])
        if (j == 0) P[128] = scaled_power(TT(4) * mad(v[8].x, v[8].x, v[8].y * v[8].y), sc);  // (x 4 like every bin of P)
        // bins 257..259 only pad the last 4-bin chunk; their weights are zero but 0 * garbage must stay 0
        if (j >= 13) P[w16::kH + (j - 13)] = 0.f;
    }
    wave_lds_fence();
    AUD_STAMP(7);

    // ---- optional spectrum outputs and the mel reduction: 4 frames x 16 filter groups on this wave ----------
    wave_mel_epilogue_pick<TT, w16::kFW, MAXS>(a, e, Pw, w16::kHp, smem,
                                          sizeof(TT) == 8 ? frame_scale_of(exps + lane % w16::kFW) : 0, it, item, t0, lane, exps);
    AUD_STAMP(8);
    AUD_STAMP_REAL(10);
    AUD_STAMP_FLUSH(a, wt, lane);
}

}  // namespace

size_t w16_region_bytes(bool f64) { return f64 ? size_t(w16::Layout<double>::kRegion) : size_t(w16::Layout<float>::kRegion); }

#define AUD_W16_PICK(TT)                                                                                  \
    (sig_dtype == AUD_F64   ? (s8 ? k_melspec_w16<TT, AUD_F64, 4, 8> : k_melspec_w16<TT, AUD_F64, 4, 4>)   \
     : sig_dtype == AUD_I16 ? (s8 ? k_melspec_w16<TT, AUD_I16, 4, 8> : k_melspec_w16<TT, AUD_I16, 4, 4>)   \
                            : (s8 ? k_melspec_w16<TT, AUD_F32, 4, 8> : k_melspec_w16<TT, AUD_F32, 4, 4>))
wave_kernel_t w16_kernel(bool f64, int sig_dtype, int n_slots) {
    const bool s8 = n_slots > 4;
    return f64 ? AUD_W16_PICK(double) : AUD_W16_PICK(float);
}
#undef AUD_W16_PICK

}  // namespace aud
