// Internal launch descriptors shared between the C API (capi.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "auditory_hip.h"

namespace aud {

constexpr int kMaxFactors = 24;

// All LDS scratch lives in the dynamic region (keeps its base 16-byte aligned; every carve
// offset below is a multiple of 16).
__device__ __forceinline__ unsigned char* dyn_lds() {
    extern __shared__ __attribute__((aligned(16))) unsigned char aud_dyn_lds[];
    return aud_dyn_lds;
}

// Workgroup -> tile order.  The hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its
// own L2), while consecutive tiles of a stream share N - S samples of input.  With the remap the workgroups that
// share an XCD (equal id mod 8) walk one contiguous run of tiles, so the overlap is an L2 hit instead of a second
// fetch through the fabric.  A bijection for any grid size (the first n % 8 runs are one tile longer); results do
// not depend on it.  It pays once the batch no longer fits the 256 MB Infinity Cache (BASELINE config 5: 1.1 GB).
__device__ __forceinline__ unsigned tile_of_workgroup(unsigned b, unsigned n, int remap) {
    if (!remap || n < 16) return b;
    const unsigned x = b & 7u, i = b >> 3, per = n >> 3, rem = n & 7u;
    return x * per + (x < rem ? x : rem) + i;
}

// Arguments of every frame->mel kernel family.  Doubles are converted to the compute
// type inside the kernel; `tw` and `filt` are already stored in the compute type.
struct MelspecArgs {
    const void* sig;
    int sig_dtype;
    const aud_item* items;
    int n_items;
    int N, S, T, border, H;
    int M;      // complex FFT length: N/2 for even N (packed real trick), N for odd N
    int ratio;  // N / M
    int nfac;
    int fac[kMaxFactors];
    const void* tw;  // [N] complex<TT>: exp(-2 pi i k / N)
    int nf;
    const int32_t* bin_pts;  // [nf+2]
    const void* filt;        // [nf*(nf+2)] TT, the reference's table layout
    double mel_log_off, mel_log_min;
    int renorm;
    double renorm_min, renorm_scale;
    int comp_log_pow;
    double dft_log_min, dft_log_off;
    float* mel;        // [n_items, nf, T]
    float* power;      // [n_items, H, T] or null
    float* log_power;  // [n_items, H, T] or null
    int F;             // frames per workgroup
    // generic kernel, Bluestein route for complex FFT lengths M with a prime factor > 25 (bl_L = 0: not used):
    // Z = chirp . IFFT_L(FFT_L(z . chirp) . bhat), L a power of two >= 2 M - 1, one frame per workgroup
    int bl_L, bl_nfac;
    int bl_fac[kMaxFactors];
    const void* bl_chirp;  // [M] complex<TT>: exp(-i pi n^2 / M)
    const void* bl_bhat;   // [L] complex<TT>: FFT_L of the wrapped conjugate chirp, / L
    const void* bl_tw;     // [L] complex<TT>: exp(-2 pi i k / L)
    int bl_inplace;        // 1: ONE padded LDS buffer, stages through registers (melspec_generic.hip stage_inplace)
    // generic kernel, smooth window lengths IN PLACE (ip_nfac > 0; bl_L = 0): the F frames of a workgroup as one batched transform
    // of F M points in one padded buffer (melspec_generic.hip plain_fft_inplace); its radices, powers of two first
    int ip_nfac;
    int ip_fac[kMaxFactors];
    const void* bl_fix;    // the fixed-geometry chirp kernel's tables (melspec_chirp.hip: twiddles of both outer stages, bhat in its digit-reversed order); null: the any-N route
    const void* tw64;  // the direct kernel (melspec_direct.hip): [N] complex<double> exp(-2 pi i k / N) whatever the plan computes in; null otherwise
    int xcd_remap;     // 1: tile_of_workgroup() order (plan option "xcd_remap", default on)
    // wave kernels: wave tiles per item (N = 2048: frames per item) and its reciprocal, set by launch_melspec_wave -- a
    // wave finds its item with one scalar multiply (tile_div) instead of the 64-bit division's twenty vector instructions
    int tiles;
    unsigned tile_mul;
    int tile_shift;    // < 0: tiles == 1
    // fused segment tail (aud_segment_batch_dev): the wave kernels w16x16 / w20x10 of a plan with WaveArgs::dct_off >= 0
    // also leave, per item, the CepstrumDct rows 1.. of the UNROUNDED log-mel values and per-tile Energy sums for
    // launch_segment_finish; both null otherwise
    void* mfcc_acc;      // [n_items][n_coefs][T] compute type (row 0 is not written: ProcessSegment overwrites it with Energy)
    void* energy_part;   // [n_items][tiles][T] compute type: sum over the tile's frames of LogPower[bin s < T][frame] (Q8 axis quirk)
    int n_coefs;
    const void* dct_rows;  // the any-N kernel's fused tail (round 6): [n_coefs][nf] DCT-I rows, compute type (its tile = the F frames of a workgroup)
    // diagnostic builds only (-DAUD_STAMPS, tools/stamp_profile.py): [waves][16] s_memtime stamps of the wave
    // kernels' phases.  Never read by anything that computes an output.
    unsigned long long* stamps;
};

#ifdef AUD_STAMPS
// one stamp = s_memtime behind a drained LDS queue, fenced against the scheduler (cdna_hip_programming.md 7)
#define AUD_STAMP_DECL unsigned long long aud_stamp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
// a kernel's phases that live in a device function of their own take / pass the stamp array
#define AUD_STAMP_PARAM , unsigned long long (&aud_stamp_)[12]
#define AUD_STAMP_ARG , aud_stamp_
#define AUD_STAMP(i)                                                                          \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(aud_stamp_[i])::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
// the chip-wide constant-rate counter (100 MHz): comparable between waves on different CUs and XCDs, which s_memtime is not
#define AUD_STAMP_REAL(i)                                                                          \
    do {                                                                                           \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(aud_stamp_[i])::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define AUD_STAMP_FLUSH(a, wave_tile, lane)                                                   \
    do {                                                                                      \
        if ((a).stamps && (lane) == 0) {                                                      \
            for (int s_ = 0; s_ < 12; ++s_) (a).stamps[size_t(wave_tile) * 16 + s_] = aud_stamp_[s_]; \
            (a).stamps[size_t(wave_tile) * 16 + 12] = __builtin_amdgcn_s_getreg((4 << 11) | 20); /* XCC_ID */ \
            (a).stamps[size_t(wave_tile) * 16 + 13] = __builtin_amdgcn_s_getreg((31 << 11) | 4); /* HW_ID */ \
        }                                                                                     \
    } while (0)
#elif defined(AUD_PHASE_MARKERS)
// static diagnostic (no GPU): "; AUD_PHASE n" comments in the assembly at the phase boundaries, fenced against the
// scheduler, for tools/phase_count.py (instructions per phase of a wave kernel)
#define AUD_STAMP_DECL
#define AUD_STAMP_PARAM
#define AUD_STAMP_ARG
#define AUD_STAMP(i)                                             \
    do {                                                         \
        __builtin_amdgcn_sched_barrier(0);                       \
        asm volatile("; AUD_PHASE " #i ::: "memory");            \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
#define AUD_STAMP_REAL(i)
#define AUD_STAMP_FLUSH(a, wave_tile, lane)
#else
#define AUD_STAMP_DECL
#define AUD_STAMP_PARAM
#define AUD_STAMP_ARG
#define AUD_STAMP(i)
#define AUD_STAMP_REAL(i)
#define AUD_STAMP_FLUSH(a, wave_tile, lane)
#endif

// n / a.tiles for n < 2^31 (Granlund-Montgomery round-up reciprocal: tile_mul = ceil(2^(31 + l) / tiles), l = ceil(log2 tiles),
// tile_shift = l - 1; launch_melspec_wave)
__device__ __forceinline__ unsigned tile_div(const MelspecArgs& a, unsigned n) {
    return a.tile_shift < 0 ? n : __umulhi(n, a.tile_mul) >> a.tile_shift;
}

struct GaborArgs {
    const float* mel;  // [n_items, rows, cols]
    int n_items, rows, cols;
    const void* k;     // [nG, SY, SX] TT
    const float* k32;  // the same taps as float32 (the LDS-staged kernel)
    int nG, SX, SY, stx, sty;
    double gain;
    int rank;  // 2 or 4
    int d0, d1, d2, d3;
    int by_time;
    int nT, nF, t_max_strides;
    float* out;
    int mode;  // plan option "gabor_kernel": -1 = float64 plans k_gabor, float32 plans LDS-staged; 0 = LDS-staged where it fits; 1 = k_gabor
};

// Arguments of the wave-autonomous kernels (melspec_wave.hip).  All read-only tables live in ONE device blob laid out
// exactly like its LDS copy -- [per-group weight rows | slot records | pass twiddles | split twiddles | column pairs] -- so
// staging is a flat 16-byte-piece copy.  Offsets are bytes from the start of dynamic LDS / the blob.
// The mel epilogue: a lane owns (frame, filter group); a group's filters sit in SLOTS, slot k of every group takes the
// same number of 4-bin chunk steps slot_steps[k] (filters are dealt to groups in order of width, so the padding is
// small), a filter's chunks are consecutive in P, its weights consecutive in the group's weight row.  Weights and the
// power spectrum in LDS are float32 for both compute types (float64 plans scale every frame's spectrum by a power of two
// first, device_common.h frame_scale), weights carry the 1/4 of the real-FFT split.
struct WaveArgs {
    const void* blob;      // device
    int blob_bytes;        // multiple of 16
    int w4_off;            // float32 weight rows
    int w_stride;          // bytes between the weight rows of two groups (an odd number of 16-byte pieces: bank spread)
    int slots_off;         // uint32 [n_groups][n_slots]: first P chunk | filter id << 16 (0xFFFF = empty); w64x16: two words
    int n_slots;
    int n_groups;          // filter groups of the epilogue (64 / frames per wave)
    unsigned char slot_steps[8];
    int twa_off;           // pass twiddles, C2<TT> [K1 - 1][lanes per frame]: W^(2 j k1)
    int tws_off;           // split twiddles, C2<TT> [N/4 + 1]: W_N^k
    int pairs_off;         // w64x16: uint16 [64][4]: base bins of the lane's two column pairs
    const void* gtab;      // w64x16: device, lane-ordered twiddles read from global memory
    int xch_off;           // byte offset of the first wave's private region inside dynamic LDS
    unsigned lds_bytes;    // dynamic LDS of the launch
    int waves;             // waves per workgroup of the launch
    int wgs_per_cu;        // the runtime's occupancy answer (aud_plan_get_info)
    int stash_off;         // w64x16: byte offset of the first wave's [n_slots][64][4] float32 output stash inside dynamic LDS; -1: none
    int dct_off;           // fused segment tail: TT [nf][kDctPitch], row f = column f of the DCT-I matrix (coefficient c at
                           // [c], zero beyond n_coefs); -1: the plan has no fused tail
};
// Arguments of the workgroup-per-item variant of the N = 400 kernel (melspec_w20.hip k_melspec_w20_item): one workgroup takes
// ALL frames of one work item -- its waves walk the item's tiles -- so the item's whole mel matrix [nf, T] can stay in LDS
// behind the frame loop and agabor.Convolve (agabor/gabor.go:225-315) runs on it without a second launch or a re-read from
// memory.  nG == 0: no gabor stage (the mel-only item kernel, plan option "item_kernel").
struct ItemArgs {
    int waves;           // waves per workgroup of the launch
    int mel_off;         // byte offset of the item's [nf][T] float32 mel matrix inside dynamic LDS (NaN already read as 0.5)
    unsigned lds_bytes;  // dynamic LDS of the launch
    int wgs_per_cu;      // the runtime's occupancy answer
    // gabor stage (the fused kernel: rank-4 output [d0, d1, 2, nG], the shape aud_process_batch_dev asks for; the LDS-staged
    // stand-alone kernel of gabor.hip: any output Convolve accepts)
    const float* k32;    // [nG][SY][SX] taps, float32 copy (the kernels take it as a direct restrict parameter: scalar loads)
    int nG, SX, SY, stx, sty;
    double gain;
    int rank;            // 2 or 4
    int d0, d1, d2, d3;  // output shape (rank 2: d0, d1)
    int by_time, t_max_strides;
    int nT, nF;          // iteration space (aud_gabor_iter_space)
    float* out;          // [n_items, output shape]
};
constexpr int kDctCoefs = 13;  // coefficients the fused tail carries per lane (the reference's default NCoefs, mel.go:71)
constexpr int kDctPitch = 14;  // row pitch of the table (16-byte rows in float64)

// PrevSmooth != 0 mode: scan along the steps of a stored power tensor
struct SmoothArgs {
    const aud_item* items;
    int n_items, H, T, N, S, border;
    float* power;      // [n_items, H, T] in/out
    float* log_power;  // [n_items, H, T] or null
    double prev_smooth, cur_smooth, log_off, log_min;
    int comp_log_pow;
};
hipError_t launch_power_smooth(const SmoothArgs& a, int compute_dtype, hipStream_t st);
hipError_t launch_mel_from_power(const MelspecArgs& a, int compute_dtype, hipStream_t st);
hipError_t launch_power_from_coefs(const double* coefs, int H, float* raw, int compute_dtype, hipStream_t st);
hipError_t launch_frame_blend(const float* raw, int raw_stride, const double* carry, int H, int step,
                              double prev, double cur, int comp_log_pow, double log_off, double log_min,
                              double* out_p, double* out_lp, int compute_dtype, hipStream_t st);

// MFCC tail (mel.CepstrumDct + Energy / deltas of SndEnv.ProcessSegment)
struct MfccArgs {
    const aud_item* items;
    int n_items, N, S, T, border, H, nf, n_coefs;
    const void* dct;         // [n_coefs][nf] DCT-I rows (compute type)
    const float* mel;        // [n_items, nf, T]
    const float* log_power;  // [n_items, H, T]
    float* mfcc;             // [n_items, n_coefs, T]
    float* deltas;           // [n_items, n_coefs, T] or null
    float* delta_deltas;     // [n_items, n_coefs, T] or null
    float* energy;           // [n_items, T] or null
};
hipError_t launch_mfcc(const MfccArgs& a, int compute_dtype, hipStream_t st);
// what is left of the tail behind a mel launch that carried mfcc_acc / energy_part: Energy from the per-tile sums, MFCC
// row 0 <- Energy, deltas and delta-deltas from the unrounded coefficients (one workgroup per item)
struct SegmentFinishArgs {
    int n_items, T, n_coefs, tiles;
    const void* mfcc_acc;      // MelspecArgs::mfcc_acc
    const void* energy_part;   // MelspecArgs::energy_part
    float* mfcc;               // [n_items, n_coefs, T]
    float* deltas;             // or null
    float* delta_deltas;       // or null
    float* energy;             // [n_items, T] or null
};
size_t segment_finish_lds_bytes(int n_coefs, int T, int compute_dtype);
// the any-N kernel can carry the tail: its power buffer has room for the F x nf unrounded log-mel values behind the F spectra
bool melspec_generic_tail_fits(int M, int F, int H, int nf, int compute_dtype, int bl_L, bool bl_inplace);
hipError_t launch_segment_finish(const SegmentFinishArgs& a, int compute_dtype, hipStream_t st);
hipError_t launch_mfcc_dct(const MfccArgs& a, int compute_dtype, hipStream_t st);

// generic any-N kernel (Stockham in LDS, radix 2/4 + per-output generic radix)
// float32 device results -> float64 in pinned, device-visible host memory (aud_host_alloc), stored over the link by the kernel
hipError_t launch_widen_to_host(const float* src, double* dst_host, size_t n, hipStream_t st);
size_t melspec_generic_lds_bytes(int M, int F, int compute_dtype, bool bluestein);
bool melspec_generic_bluestein_inplace(int L);
int melspec_generic_pick_F(int M, int compute_dtype);
// smooth lengths in place: frames per workgroup F (0: the length has a factor the in-place stages do not run, or nothing fits),
// the stage radices and the launch's LDS; forced_F > 0 asks for that F (plan option "plain_frames")
int melspec_generic_plain_inplace(int M, int H, int nf, int T, int compute_dtype, int forced_F, int* fac, int* nfac, size_t* lds);
int melspec_generic_bluestein_L(int M, int compute_dtype);
hipError_t melspec_generic_prepare(size_t lds_bytes);
hipError_t launch_melspec_generic(const MelspecArgs& a, int compute_dtype, hipStream_t st);

// window lengths nothing else runs (melspec_direct.hip): the O(N H) sum, one frame per workgroup, only the spectrum in LDS
size_t melspec_direct_lds_bytes(int H, int nf, int compute_dtype);
hipError_t melspec_direct_prepare();
hipError_t launch_melspec_direct(const MelspecArgs& a, int compute_dtype, hipStream_t st);

// the chirp convolution of fixed length 2304 = 16 x 16 x 9 (melspec_chirp.hip): odd window lengths 1024 < N <= 1152 -- the
// reference's 25 ms at 44.1 kHz = 1103 samples -- with compile-time stage geometry, five LDS round trips instead of six
bool melspec_chirp_serves(int N, int compute_dtype);  // (float64 plans: two frames per transform)
size_t melspec_chirp_lds_bytes();
int melspec_chirp_table_len();
void melspec_chirp_tables(const double* twl, const double* bhat, double* out);
bool melspec_chirp_tail_fits(int H, int nf);
hipError_t launch_melspec_chirp(const MelspecArgs& a, hipStream_t st);

// wave-autonomous kernels (melspec_wave.hip): N = 512 as 16 x 16 (kind 1), N = 400 as 20 x 10 (kind 3), N = 2048 as
// 16 x 16 x 4 with one frame per wave (kind 4); no workgroup barrier behind the table staging.
// *_geometry: filter groups of the epilogue, lanes per frame and twiddle rows of the pass table; *_finish: carve LDS
// behind a blob of blob_bytes, false if it cannot fit; *_prepare: LDS opt-in + the runtime's occupancy answer.
struct WaveGeometry {
    int n_groups, lanes_per_frame, k1_rows, split_count;
};
int melspec_wave_kind(int N);  // 0: this window length has no wave kernel
bool melspec_wave_geometry(int kind, int N, WaveGeometry* g);
bool melspec_wave_finish(int kind, int compute_dtype, WaveArgs* e);
hipError_t melspec_wave_prepare(int kind, int compute_dtype, WaveArgs* e);
hipError_t launch_melspec_wave(int kind, const MelspecArgs& a, const WaveArgs& e, int compute_dtype, hipStream_t st);
int melspec_wave_frames_per_wave(int kind);
// workgroup-per-item variant (N = 400 only): carve LDS behind the plan's blob for `nf x T` mel values, false if it cannot fit;
// `it` keeps the launch shape; the launch
bool melspec_item_finish(int kind, int compute_dtype, const WaveArgs& e, int nf, int T, ItemArgs* it);
hipError_t melspec_item_prepare(int kind, int compute_dtype, const WaveArgs& e, ItemArgs* it);
hipError_t launch_melspec_item(int kind, const MelspecArgs& a, const WaveArgs& e, const ItemArgs& it, int compute_dtype, hipStream_t st);

hipError_t launch_gabor(const GaborArgs& a, int compute_dtype, hipStream_t st);

// k-WTA settling on the gabor output (SndEnv.ApplyKwta, sound/sndenv.go:313-323)
struct KwtaFffb {
    int on;
    float gi, ff, fb, fb_dt, max_vs_avg, ff0;
};
struct KwtaArgs {
    const float* raw;  // [n_items, n]  excitatory conductances (the raw gabor output)
    float* act;        // [n_items, n]  activations: out, and in unless start_from_raw
    int n_items, n;
    int lay_n, pl_n;   // pool level: lay_n pools of pl_n consecutive values; layer level only: lay_n = 0
    int start_from_raw;
    int sum_order;     // 0: running float32 sums in the reference's index order; 1: fixed tree (faster)
    int compact;       // sum_order 0, pool level: the serial layer sum walks an order-preserving list of the non-zeros
    float* state;      // [n_items, lay_n, 2] {FBi, Act.Avg} carried between calls, or null
    int32_t* cycles;   // [n_items] settling cycles run, or null
    int iters;
    float del_act_thr;
    KwtaFffb lay, pool;
    // nxx1.Params and what its Update() derives
    float gain, nvar, interp_range, gain_cor_range, gain_cor;
    float sig_gain_nvar, sig_mult_eff, sig_val_at0, interp_val;
    float gbar_e, gbar_l, gbar_i;
    float erev_sub_thr_i, erev_sub_thr_l, thr_sub_erev_e;
    float act_dt;
    unsigned lds_bytes;
};
size_t kwta_lds_bytes(int n, int lay_n, bool compact);
hipError_t kwta_prepare(unsigned lds_bytes);
hipError_t launch_kwta(const KwtaArgs& a, hipStream_t st);

}  // namespace aud
