// Wave-autonomous frame -> FFT -> power -> mel -> log kernels: N = 512 ("w16x16"), N = 400 ("w20x10") and N = 2048
// ("w64x16", one frame per wave).
//
//   * the unit of work is a WAVE, not a workgroup: 64 lanes carry 4 frames x 16 lanes (N = 512), 6 frames x 10 lanes
//     (N = 400) or one frame (N = 2048) from the samples to the mel values.  Lanes of one wave exchange data through a
//     wave-private LDS region ordered by wave_lds_fence() -- the hardware runs a wave's LDS instructions in order -- so
//     the data path has no workgroup barrier;
//   * every read-only table (mel weight rows, the epilogue's slot records, pass and split twiddles) comes as ONE blob
//     that the workgroup copies into LDS at its very start: the blob loads are issued first, the operand loads behind
//     them, and a counted wait (the loads return in order) lets the blob be stored and the single barrier be passed
//     while the operands are still in flight;
//   * operands: where an item's samples come from is wave-uniform, so the load route is chosen on the scalar unit and
//     the common one is one 8-byte buffer load per sample pair through a descriptor over the item's samples -- the
//     hardware's range check is SndToWindow's left zero pad (device_common.h, "first-pass operands");
//   * the transposes go through LDS one component at a time (all real parts, then all imaginary parts), which halves
//     the footprint; the power spectrum then reuses the same region -- as float32, each frame scaled by a power of two
//     (device_common.h frame_scale): half the epilogue's LDS traffic and float32 multiply-adds for float64 plans;
//   * N = 400: the partner of Z[k1 + 20 k2] in the real-FFT split is element (20 - k1, 9 - k2), so a lane takes the row
//     PAIR (r, 20 - r) and has both halves of all its pairs in its own registers: no cross-lane traffic; N = 2048 does
//     the same with pairs of columns;
//   * no LDS access sits under a lane condition: idle lanes read rows they do not use or shadow a working lane -- a
//     lane-conditional access block costs hipcc 60-100 registers in these kernels;
//   * the wave index is scalar (readfirstlane): the work-item record is one scalar load and the address arithmetic runs
//     on the scalar unit;
//   * the mel reduction runs slot-uniform chunk steps: scalar loop bounds, two LDS reads and four multiply-adds per step
//     (wave_mel_epilogue, device_common.h); the final logarithm is taken in float32 (the stored value is a float32).
//
// Reference semantics: sound/sndenv.go:438-478, dft/dft.go:53-85, mel/mel.go:120-153.
#include <algorithm>

#include "wave_common.h"

namespace aud {

// waves per workgroup: the waves of a workgroup share one LDS copy of the table blob (eight-wave workgroups were
// tried for the float64 N = 400 regions: their one barrier cost a quarter of the wave's life)
// (the N = 2048 kernel's float64 tables + planes fill a CU with ONE workgroup of twelve waves = three per SIMD; two
// workgroups of six were measured to leave one of them waiting: six waves land 2-2-1-1 on the SIMDs and the second
// workgroup's pair does not fit beside the first's at 160 VGPRs)
static int wave_kernel_waves(int kind, int compute_dtype) {
    return kind == 4 && compute_dtype == AUD_F64 ? 12 : 4;
}

int melspec_wave_kind(int N) { return N == w16::kN ? 1 : N == w20::kN ? 3 : N == w64::kN ? 4 : 0; }

int melspec_wave_frames_per_wave(int kind) { return kind == 1 ? w16::kFW : kind == 3 ? w20::kFW : kind == 4 ? 1 : 0; }

bool melspec_wave_geometry(int kind, int N, WaveGeometry* g) {
    if (kind == 1 && N == w16::kN) {
        *g = WaveGeometry{64 / w16::kFW, 16, 16, w16::kM / 2 + 1};
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

static size_t wave_region_bytes(int kind, bool f64) {
    return kind == 1 ? w16_region_bytes(f64) : kind == 3 ? w20_region_bytes(f64) : w64_region_bytes(f64);
}
// the instantiation a launch runs: kind, compute type, sample type and the slot capacity of its epilogue (4 or 8)
static wave_kernel_t wave_kernel(int kind, bool f64, int sig_dtype, int n_slots) {
    return kind == 1 ? w16_kernel(f64, sig_dtype, n_slots) : kind == 3 ? w20_kernel(f64, sig_dtype, n_slots)
                                                                        : w64_kernel(f64, sig_dtype, n_slots);
}

bool melspec_wave_finish(int kind, int compute_dtype, WaveArgs* e) {
    const int nw = wave_kernel_waves(kind, compute_dtype);
    const size_t first = (size_t(e->blob_bytes) + 255) & ~size_t(255);
    size_t total = first + size_t(nw) * wave_region_bytes(kind, compute_dtype == AUD_F64);
    total = std::max(total, size_t(64) * 64 * nw);  // blob_store writes 4 x 16 bytes per thread whatever the blob's size
    if (total > 160 * 1024) return false;
    e->stash_off = -1;
    // the output stash of the one-frame-per-wave kernel (wave_mel_flush4), where LDS has room for it -- FLOAT32 plans only: in
    // an in-process A/B on 1280 streams of 5 s they ran 1224 us with it against 1380 without (the 128 scattered 4-byte stores
    // per frame were their bound), float64 plans 2090 us with it against 2066 without (their bound is the LDS pipe, which
    // the stash loads further)
    if (kind == 4 && compute_dtype == AUD_F32) {
        const size_t stash = size_t(nw) * size_t(e->n_slots) * 64 * 4 * sizeof(float);
        const size_t at = (total + 15) & ~size_t(15);
        if (at + stash <= 160 * 1024) {
            e->stash_off = int(at);
            total = at + stash;
        }
    }
    e->xch_off = int(first);
    e->lds_bytes = unsigned(total);
    e->waves = nw;
    e->wgs_per_cu = 0;  // set by melspec_wave_prepare
    return true;
}

hipError_t melspec_wave_prepare(int kind, int compute_dtype, WaveArgs* e) {
    if (e->n_slots > 8) return hipErrorInvalidValue;  // more filters per group than the epilogue has slots for
    int per_cu = 0;
    for (int sig_dtype : {AUD_F64, AUD_I16, AUD_F32}) {
        const void* fn = reinterpret_cast<const void*>(wave_kernel(kind, compute_dtype == AUD_F64, sig_dtype, e->n_slots));
        // the attribute belongs to the kernel instantiation, not to a plan: two plans may share one, so it is only ever
        // raised to the device's limit (lowering it would break the launches of a plan created earlier)
        if (e->lds_bytes > 64u * 1024u) {
            hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (rc != hipSuccess) return rc;
        }
        hipError_t rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * e->waves, e->lds_bytes);
        if (rc != hipSuccess) return rc;
    }
    e->wgs_per_cu = per_cu;  // (of the float32-sample instantiation, the one the device entry points mostly run)
    return hipSuccess;
}

// ---- the workgroup-per-item variant (N = 400: k_melspec_w20_item) ---------------------------------------------------
// LDS = the plan's blob + one region per wave + the item's [nf][T] float32 mel matrix.  Five waves per workgroup: three such
// workgroups fit a CU's 160 KB for the metric's table (15 waves per CU; the 18 tiles of an item are four rounds of five).
bool melspec_item_finish(int kind, int compute_dtype, const WaveArgs& e, int nf, int T, ItemArgs* it) {
    if (kind != 3) return false;
    constexpr int nw = 5;
    const size_t first = size_t(e.xch_off), region = wave_region_bytes(kind, compute_dtype == AUD_F64);
    const size_t mel = (size_t(nf) * size_t(T) * 4 + 15) & ~size_t(15);
    size_t total = first + size_t(nw) * region + mel;
    total = std::max(total, size_t(64) * 64 * nw);
    if (total > 160 * 1024) return false;
    it->waves = nw;
    it->mel_off = int(first + size_t(nw) * region);
    it->lds_bytes = unsigned(total);
    it->wgs_per_cu = 0;
    return true;
}

hipError_t melspec_item_prepare(int kind, int compute_dtype, const WaveArgs& e, ItemArgs* it) {
    if (kind != 3 || e.n_slots > 8) return hipErrorInvalidValue;
    int per_cu = 0;
    for (int sig_dtype : {AUD_F64, AUD_I16, AUD_F32}) {
        const void* fn = reinterpret_cast<const void*>(w20_item_kernel(compute_dtype == AUD_F64, sig_dtype, e.n_slots, it->waves));
        if (!fn) return hipErrorInvalidValue;
        if (it->lds_bytes > 64u * 1024u) {
            hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (rc != hipSuccess) return rc;
        }
        hipError_t rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * it->waves, it->lds_bytes);
        if (rc != hipSuccess) return rc;
    }
    it->wgs_per_cu = per_cu;
    return hipSuccess;
}

hipError_t launch_melspec_item(int kind, const MelspecArgs& a, const WaveArgs& e, const ItemArgs& it, int compute_dtype,
                               hipStream_t st) {
    if (kind != 3 || a.n_items <= 0) return kind != 3 ? hipErrorInvalidValue : hipSuccess;
    const int tiles = (a.T + w20::kFW - 1) / w20::kFW;
    item_kernel_t fn = w20_item_kernel(compute_dtype == AUD_F64, a.sig_dtype, e.n_slots, it.waves);
    if (!fn) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fn, dim3(unsigned(a.n_items)), dim3(64 * it.waves), it.lds_bytes, st, a.items, unsigned(a.n_items),
                       unsigned(tiles), e.blob, e.blob_bytes, it.k32, a, e, it);
    return hipGetLastError();
}

hipError_t launch_melspec_wave(int kind, const MelspecArgs& a, const WaveArgs& e, int compute_dtype, hipStream_t st) {
    const int fw = kind == 4 ? 1 : melspec_wave_frames_per_wave(kind);
    const int64_t tiles = (int64_t(a.T) + fw - 1) / fw;
    const int64_t per_wg = int64_t(e.waves) * (kind == 4 ? w64::kFPW : 1);  // wave tiles per workgroup
    const int64_t wgs64 = (int64_t(a.n_items) * tiles + per_wg - 1) / per_wg;
    if (wgs64 > 0x7FFFFFFF || int64_t(a.n_items) * tiles >= (int64_t(1) << 31) - 64 * per_wg) return hipErrorInvalidValue;
    MelspecArgs b = a;
    b.tiles = int(tiles);
    int l = 0;
    while ((int64_t(1) << l) < tiles) ++l;
    b.tile_shift = l - 1;
    b.tile_mul = l == 0 ? 0u : unsigned(((uint64_t(1) << (31 + l)) + uint64_t(tiles) - 1) / uint64_t(tiles));
    hipLaunchKernelGGL(wave_kernel(kind, compute_dtype == AUD_F64, a.sig_dtype, e.n_slots), dim3(unsigned(wgs64)),
                       dim3(64 * e.waves), e.lds_bytes, st, b.items, unsigned(a.n_items) * unsigned(tiles), unsigned(tiles),
                       b.tile_mul, b.tile_shift, e.blob, e.blob_bytes, unsigned(wgs64), a.xcd_remap, b, e);
    return hipGetLastError();
}

}  // namespace aud
