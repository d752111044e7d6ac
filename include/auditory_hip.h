/*
 * auditory_hip.h -- C ABI of libauditory_hip.so: the MI355X (gfx950) implementation of
 * emer/auditory's signal -> framed FFT -> power/log-power -> mel filterbank -> 2-D gabor
 * hot path.  This is the drop-in boundary: the reference has no FFI of its own (it is
 * pure Go), so each entry point names the exported Go symbol (file:line, relative to the
 * reference root) whose body a cgo shim routes here.  See INTEGRATION.md for the shim.
 *
 * Conventions
 *  - extern "C", plain C types, explicit sizes; every function returns an int status
 *    (AUD_OK == 0) unless documented otherwise.  No C++ exceptions cross the boundary.
 *  - "_dev" entry points take DEVICE pointers and a hipStream_t passed as void*; they
 *    only enqueue work (no allocation, no synchronisation: they are graph-capturable).
 *  - "_host" entry points take HOST pointers (float64 in, float64/float32 out, matching
 *    the reference's etensor.Float64 / etensor.Float32 buffers), stage through a
 *    ctx-owned workspace, and return after the result is in the caller's buffer.  The
 *    library never keeps a caller pointer after return (cgo rule).
 *  - There is NO CPU fallback anywhere in this library: without a usable HIP device
 *    aud_init fails with AUD_EHIP and nothing else can be called.
 *  - One aud_ctx per device; one process per GPU is the intended deployment.
 */
#ifndef AUDITORY_HIP_H
#define AUDITORY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AUD_VERSION 210 /* 0.2.1: + aud_signal_sync, aud_melspec_batch_live, aud_melspec_mfcc_batch_live (0.2.0: AUD_F64 is 0 -- a zero-initialised aud_plan_desc is the float64 plan --, AUD_F32 is 1) */

/* status codes */
#define AUD_OK 0
#define AUD_EINVAL 1 /* bad argument / shape outside the supported envelope */
#define AUD_EHIP 2   /* HIP runtime error (see aud_last_error) */
#define AUD_ERCCL 3  /* RCCL error */
#define AUD_ENOMEM 4
#define AUD_ESHORT 5 /* sndenv.go:458-460 "end beyond signal length" (per-step API only) */
#define AUD_EBROKEN 6 /* the direct gather is unusable until aud_gather_destroy / aud_gather_create: a peer's slab never arrived
                       * (the missing slots of the slab were filled with NaN), or a call failed half-way */

/* element types of signal buffers / compute.  AUD_F64 is ZERO on purpose: the reference computes in float64 throughout
 * (dft/dft.go:42-85, mel/mel.go:120-153), so a zero-initialised aud_plan_desc, a memset, a Go zero value all select the plan
 * that meets the 1e-5 criterion on every element.  float32 compute is an explicit opt-in (AUD_FAST_F32): ~1.35x faster, a few
 * elements per million past 1e-5 of the reference. */
#define AUD_F64 0
#define AUD_F32 1
#define AUD_I16 2 /* raw 16-bit PCM; normalised on device by /0x7FFF (sound.go:138) */
#define AUD_FAST_F32 AUD_F32 /* aud_plan_desc.compute_dtype: the non-conforming fast plan, never a default */

typedef struct aud_ctx aud_ctx;
typedef struct aud_plan aud_plan;

/* ---- parameter blocks (field-for-field mirrors of the Go structs) ---------------- */

/* sound.Params, sound/sndenv.go:24-61 */
typedef struct {
    double win_ms;           /* WinMs */
    double step_ms;          /* StepMs */
    double segment_ms;       /* SegmentMs */
    double stride_ms;        /* StrideMs */
    int32_t border_steps;    /* BorderSteps */
    int32_t channel;         /* Channel */
    int32_t win_samples;     /* WinSamples     (derived) */
    int32_t step_samples;    /* StepSamples    (derived) */
    int32_t segment_samples; /* SegmentSamples (derived) */
    int32_t stride_samples;  /* StrideSamples  (derived) */
    int32_t segment_steps;   /* SegmentSteps   (derived) */
} aud_sound_params;

/* dft.Params, dft/dft.go:15-31 */
typedef struct {
    int32_t comp_log_pow; /* CompLogPow */
    double log_min;       /* LogMin */
    double log_offset;    /* LogOffSet */
    double prev_smooth;   /* PrevSmooth */
    double cur_smooth;    /* CurSmooth */
} aud_dft_params;

/* mel.FilterBank, mel/mel.go:16-44 */
typedef struct {
    int32_t n_filters;   /* NFilters */
    double lo_hz;        /* LoHz */
    double hi_hz;        /* HiHz */
    double log_off;      /* LogOff */
    double log_min;      /* LogMin */
    int32_t renorm;      /* Renorm */
    double renorm_min;   /* RenormMin */
    double renorm_max;   /* RenormMax */
    double renorm_scale; /* RenormScale */
} aud_mel_fbank;

/* agabor.Filter, agabor/gabor.go:17-42 */
typedef struct {
    int32_t off;         /* Off */
    double wave_len;     /* WaveLen */
    double orientation;  /* Orientation */
    double sigma_width;  /* SigmaWidth */
    double sigma_length; /* SigmaLength */
    double phase_offset; /* PhaseOffset */
    int32_t circle_edge; /* CircleEdge */
    int32_t circular;    /* Circular */
} aud_gabor_spec;

/* agabor.FilterSet (scalar part), agabor/gabor.go:45-70 */
typedef struct {
    int32_t size_x;     /* SizeX */
    int32_t size_y;     /* SizeY */
    int32_t stride_x;   /* StrideX */
    int32_t stride_y;   /* StrideY */
    double gain;        /* Gain */
    int32_t distribute; /* Distribute */
} aud_gabor_set;

/* One work item of a batch = one segment of one mono stream.
 * Frame s (0 <= s < segment_steps) covers samples
 *   [start0 + step_samples*(s - border_steps), ... + win_samples)
 * of the stream that begins at element sig_off of the signal buffer and is sig_len
 * samples long.  start0 = segment*StrideSamples + MSecToSamples(add) (sndenv.go:440-441).
 * Negative positions read as zero; a frame whose end exceeds sig_len is masked to zero,
 * as are all later frames of the segment (sndenv.go:354-358, :458-460).
 * sig_stride is the distance, in elements of the signal buffer, between successive samples of the
 * stream: 1 (or 0) for a mono buffer; 2 with sig_off = 0 / 1 addresses the left / right channel of an
 * interleaved stereo PCM buffer (sound.go:116-127) as two mono streams without a de-interleaving copy.
 * sig_len counts samples of the stream, not buffer elements. */
typedef struct {
    int64_t sig_off;
    int32_t sig_len;
    int32_t start0;
    int32_t sig_stride;
    int32_t reserved; /* 0 */
} aud_item;

/* The scalar part of a plan.  It holds NO pointers: a cgo caller builds it in Go memory and may pass it by address
 * (the cgo rules forbid handing C a Go pointer to memory that itself contains Go pointers).  The tables -- built
 * host-side by aud_mel_init_filters / aud_gabor_to_tensor so device code never re-derives them -- are separate
 * arguments of aud_plan_create. */
typedef struct {
    int32_t win_samples;         /* N: FFT length = window length, no taper (sndenv.go:470) */
    int32_t step_samples;        /* S */
    int32_t segment_steps;       /* T */
    int32_t border_steps;
    aud_dft_params dft;
    aud_mel_fbank mel;
    int32_t n_gabor;             /* 0 = no gabor stage */
    aud_gabor_set gabor;
    int32_t compute_dtype;       /* 0 = AUD_F64 (default: the reference's arithmetic) or AUD_FAST_F32 (opt-in) */
    int32_t mfcc_coefs;          /* mel.Params.NCoefs if the MFCC tail is wanted (Mel.MFCC), else 0 */
} aud_plan_desc;

/* ---- host-side setup: no GPU needed --------------------------------------------- */

int aud_version(void);
const char* aud_status_string(int status);

/* sound.MSecToSamples, sound/sndenv.go:522-524 */
int aud_msec_to_samples(double ms, int rate);
/* sound.SamplesToMSec, sound/sndenv.go:527-529 */
double aud_samples_to_msec(int samples, int rate);
/* SndEnv.ParamDefaults, sound/sndenv.go:64-71 */
void aud_sound_params_defaults(aud_sound_params* p);
/* the derivations of SndEnv.Init, sound/sndenv.go:202-207; AUD_EINVAL if sample_rate <= 0 (:196-201) */
int aud_sound_params_derive(aud_sound_params* p, int sample_rate);
/* SndEnv.Init's SegCnt, sound/sndenv.go:263-265 */
int aud_seg_cnt(int signal_len, int segment_samples, int stride_samples, int channels);
/* SndEnv.Tail / SndEnv.Pad length, sound/sndenv.go:503-519 */
int aud_tail(int signal_len, int segment_samples, int stride_samples);
int aud_pad_len(int signal_len, int segment_samples, int stride_samples, int step_samples);
/* SndEnv.AdjustForSilence, sound/sndenv.go:274-294: returns the Go function's `offset` (ms, truncated toward
 * zero; -1 if sample_rate <= 0) and in *delta_samples what happens to the front of the signal:
 * < 0 trim that many samples, > 0 prepend that many zeros, 0 leave it alone. */
int aud_adjust_for_silence(double add_ms, double existing_ms, int sample_rate, int* delta_samples);
/* Wave.GetFloatAtIdx, sound/sound.go:130-141 */
double aud_pcm_to_float(int value, int bit_depth);

/* dft.Params.Defaults, dft/dft.go:33-39 */
void aud_dft_defaults(aud_dft_params* d);
/* mel.FilterBank.Defaults, mel/mel.go:171-180 */
void aud_mel_defaults(aud_mel_fbank* m);
/* mel.FreqToMel / MelToFreq / FreqToBin, mel/mel.go:156-168 */
double aud_freq_to_mel(double freq);
double aud_mel_to_freq(double mel);
int aud_freq_to_bin(double freq, double n_fft, double sample_rate);
/* mel.Params.InitFilters, mel/mel.go:77-117.  bin_pts [nf+2], hz_pts [nf+2] (may be NULL),
 * filters [nf, nf+2].  Sets m->renorm = 0 as mel.go:80 does.  AUD_EINVAL where the Go
 * code would index past the end of the filter tensor (panic). */
int aud_mel_init_filters(aud_mel_fbank* m, int dft_size, int sample_rate, int32_t* bin_pts,
                         double* hz_pts, double* filters);

/* agabor.Active, agabor/gabor.go:329-336: compacts the !Off specs into `active`, returns the count */
int aud_gabor_active(const aud_gabor_spec* specs, int n, aud_gabor_spec* active);
/* agabor.ToTensor, agabor/gabor.go:89-222 (applies Filter.Defaults :73-86 to each active spec).
 * specs: ALL specs (Off ones are skipped); out: [n_active, size_y, size_x].  Returns n_active via *n_out. */
int aud_gabor_to_tensor(const aud_gabor_spec* specs, int n, const aud_gabor_set* set, double* out,
                        int* n_out);
/* Iteration space of agabor.Convolve, agabor/gabor.go:231-262: number of time / frequency
 * positions visited for a [mel_rows, mel_cols] input and an output of the given rank/shape.
 * AUD_EINVAL for the shapes Convolve rejects (cols < SizeX, rank not 2 or 4). */
int aud_gabor_iter_space(const aud_gabor_set* set, int mel_rows, int mel_cols, int out_rank,
                         const int32_t* out_shape, int32_t* n_t, int32_t* n_f,
                         int32_t* t_max_strides);

/* ---- device context / plans ------------------------------------------------------ */

int aud_init(int device_id, aud_ctx** ctx);
int aud_shutdown(aud_ctx* ctx);
const char* aud_last_error(const aud_ctx* ctx); /* never NULL */
int aud_device_id(const aud_ctx* ctx);

/* bin_pts      [n_filters + 2]            mel.Params.BinPts          (mel/mel.go:16-31)
 * mel_filters  [n_filters, n_filters + 2] SndEnv.MelFilters.Values   (sound/sndenv.go:139; built by mel.go:77-117)
 * gabor_filters [n_gabor, size_y, size_x] FilterSet.Filters.Values   (agabor/gabor.go:45-70), NULL when n_gabor = 0
 * The tables are copied to the device before the call returns; nothing is retained. */
int aud_plan_create(aud_ctx* ctx, const aud_plan_desc* desc, const int32_t* bin_pts, const double* mel_filters,
                    const double* gabor_filters, aud_plan** plan);
int aud_plan_destroy(aud_plan* plan);
/* which frame->mel kernel family the plan selected: "w16x16" (N = 512), "w20x10" (N = 400), "w64x16" (N = 2048),
 * "chirp2304" (float64 plans with a window up to 1152 samples that has no smooth in-place route: dft.go:42-50 at the reference's own N = 1103, at 551, 1001 ...), "generic" (any other N
 * whose transform fits a workgroup's LDS, and any plan with "kernel" = 1) or "direct" (every other N: the O(N H) sum) -- diagnostic */
const char* aud_plan_kernel_name(const aud_plan* plan);
/* Tuning / diagnostic switches; results are identical whatever they are set to (up to the last-place effects of a
 * different summation order between the two kernel families).
 *   "kernel"    0 automatic (default: the wave-autonomous kernel of the plan's N where there is one), 1 the generic any-N
 *               kernel (the A/B and parity baseline of the wave kernels)
 *   "xcd_remap" 1 (default) workgroups that share an XCD take one contiguous run of tiles (L2 reuse of the
 *               samples neighbouring tiles share), 0 tiles in workgroup-id order
 *   "item_kernel"  -1 / 0 (default) the tile kernel; 1 the workgroup-per-item kernel where the plan has one (N = 400): a
 *               aud_process_batch_dev call becomes ONE launch with Convolve behind the frame loop (measured slower at 256 items per
 *               launch: DESIGN.md 4.5)
 *   "gabor_kernel" -1 (default) by compute type: float64 plans the all-float64 one-thread-per-position kernel (gabor.go:268-283 as
 *               written), float32 plans the LDS-staged kernel; 0 the LDS-staged kernel (float32 taps and row sums: for a float64
 *               plan an explicit opt-in, ~1e-6 of the all-float64 sum); 1 one thread per position
 *   "fused_tail"   -1 / 1 (default) aud_segment_batch_dev lets the plan's mel kernel carry the MFCC tail wherever it can; 0 always
 *               the two launches on the float32-stored tensors
 *   "chirp_kernel" 1 (default) the fixed-geometry chirp kernel of L = 2304 where it serves the plan (see aud_plan_kernel_name);
 *               0 the any-N kernel's Bluestein route (its A/B and parity partner)
 *   "plain_inplace" 1 (default) smooth window lengths run the any-N kernel IN PLACE (the workgroup's frames as one batched transform
 *               in one padded LDS buffer) where every stage fits a thread's registers; 0 the two-buffer autosort route
 *   "plain_frames"  frames per workgroup of that in-place route: 1 / 2 / 4 / 8 / 16 (AUD_EINVAL where the length does not run with
 *               that many, the plan unchanged); the plan picks the largest that keeps four workgroups per CU
 *   "lds_pad"   extra dynamic LDS per workgroup of the wave kernels, bytes (total <= 64 KB): fewer workgroups per CU, i.e. registers
 *               left free for another kernel's waves (occupancy experiments); 0 (default) none
 *   "stamps_lo" / "stamps_hi"  the two halves of a device address for the s_memtime stamps of the DIAGNOSTIC build
 *               (-DAUD_STAMPS, tools/stamp_profile.py); unknown to the product build
 * AUD_EINVAL for an unknown name. */
int aud_plan_set_option(aud_plan* plan, const char* name, int value);

/* Launch facts of the kernel the plan will run for the mel path (diagnostics: profiles/, bench.py).
 *   "lds_bytes"        LDS per workgroup
 *   "waves_per_wg"     waves of 64 lanes per workgroup
 *   "wgs_per_cu"       workgroups resident per compute unit (the runtime's occupancy answer at plan time)
 *   "frames_per_wave"  frames one wave transforms together (0: the generic kernel)
 *   "epilogue_steps"   wave kernels: filter steps of the mel epilogue, summed over its slots (padding included)
 *   "bluestein_L"      generic kernel: length of the transforms of its Bluestein route (0: direct factorisation)
 *   "bluestein_inplace" 1 if that route runs in one padded buffer (stages through registers)
 *   "generic_frames_per_wg"  frames a workgroup of the generic kernel transforms at once
 *   "chirp_kernel"     1 if the fixed-geometry chirp kernel runs the plan ("chirp2304")
 *   "plain_inplace"    1 if the generic kernel runs the plan's smooth window length in place
 *   "item_kernel"      1 if the plan has the workgroup-per-item kernel ("item_waves", "item_lds_bytes": its launch shape)
 * AUD_EINVAL for an unknown name. */
int aud_plan_get_info(const aud_plan* plan, const char* name, int64_t* value);

/* ---- hot path, device-resident (what bench.py times) ----------------------------- */

/* The SndEnv.ProcessSegment frame loop (sound/sndenv.go:342-359 -> :438-478) fused with
 * dft.Params.Filter (dft/dft.go:42-85) and mel.Params.FilterDft (mel/mel.go:120-153)
 * for n_items segments at once.
 *   sig        device, element type sig_dtype (AUD_F32 / AUD_F64 / AUD_I16)
 *   items      device, [n_items]
 *   mel        device, float32 [n_items, n_filters, T]   (MelFBankSegment per item)
 *   power      device, float32 [n_items, H, T] or NULL   (PowerSegment;    H = N/2+1)
 *   log_power  device, float32 [n_items, H, T] or NULL   (LogPowerSegment; needs CompLogPow)
 * Every cell of every output is written (masked frames as zero). */
int aud_melspec_batch_dev(aud_plan* plan, const void* sig, int sig_dtype, const aud_item* items,
                          int n_items, float* mel, float* power, float* log_power, void* stream);

/* agabor.Convolve (agabor/gabor.go:225-315) for n_items mel matrices at once.
 *   mel   device float32 [n_items, mel_rows, mel_cols]
 *   out   device float32 [n_items, out_shape...]; rank 2 ([2*nFy, nFx*nG], gbv.go:800-813) or
 *         rank 4 ([PoolsY, PoolsX, UnitsY, UnitsX], sndenv.go:218-220).  Only the cells the
 *         reference writes are written; the rest keep their contents.
 * AUD_EINVAL (nothing written) for the shapes Convolve rejects and for shapes where the
 * Go code would index out of range. */
int aud_gabor_batch_dev(aud_plan* plan, const float* mel, int n_items, int mel_rows, int mel_cols,
                        int out_rank, const int32_t* out_shape, int by_time, float* out,
                        void* stream);

/* ProcessSegment + ApplyGabor (sound/sndenv.go:481-497, 4-D pooled output) in one call:
 * mel as above, gabor float32 [n_items, pools_y, pools_x, 2, n_gabor]. */
int aud_process_batch_dev(aud_plan* plan, const void* sig, int sig_dtype, const aud_item* items,
                          int n_items, float* mel, int pools_y, int pools_x, float* gabor,
                          void* stream);

/* The MFCC tail of SndEnv.ProcessSegment on stored tensors (needs desc.mfcc_coefs > 0):
 * mel.Params.CepstrumDct (mel/mel.go:192-212) per processed step, Energy and the row-0 overwrite
 * (sound/sndenv.go:360-372, with the reference's axis quirk), deltas and delta-deltas
 * (sound/sndenv.go:378-431, running sums carried across coefficients as in the reference).
 *   mel [n_items, nf, T] and log_power [n_items, H, T] as written by aud_melspec_batch_dev;
 *   mfcc [n_items, NCoefs, T]; deltas / delta_deltas [n_items, NCoefs, T] or NULL (delta_deltas
 *   needs deltas); energy [n_items, T] or NULL.  AUD_EINVAL if T > H (the Go code indexes out of range). */
int aud_mfcc_batch_dev(aud_plan* plan, const aud_item* items, int n_items, const float* mel,
                       const float* log_power, float* mfcc, float* deltas, float* delta_deltas,
                       float* energy, void* stream);

/* The whole of SndEnv.ProcessSegment (sound/sndenv.go:342-431) with Mel.MFCC on, device-resident: the frame loop of
 * aud_melspec_batch_dev and the tail of aud_mfcc_batch_dev as ONE pass over the spectrum.  Where the plan runs a wave
 * kernel for N = 400 / N = 512, has at most 13 coefficients and dft.PrevSmooth == 0, the frame->mel kernel itself leaves
 * the CepstrumDct of the UNROUNDED log-mel values (the reference's tensors are float64, sndenv.go:106-136; the float32
 * mel tensor of this boundary is only what is stored) and per-tile Energy sums in `workspace`, and a small second launch
 * finishes Energy, the row-0 overwrite, deltas and delta-deltas from them; any other plan runs the two calls above.
 *   mel as in aud_melspec_batch_dev; power / log_power [n_items, H, T] or NULL; mfcc [n_items, NCoefs, T];
 *   deltas / delta_deltas [n_items, NCoefs, T] or NULL; energy [n_items, T] or NULL;
 *   workspace: device, 16-byte aligned, aud_segment_workspace_bytes(plan, n_items) bytes, contents undefined afterwards
 *   (caller-owned so that the call can sit inside a stream capture); it also holds the LogPowerSegment and -- with
 *   dft.PrevSmooth != 0, whose scan runs on the stored tensor -- the PowerSegment a caller passing NULL does not keep.
 * Needs desc.mfcc_coefs > 0 and dft.CompLogPow; AUD_EINVAL if T > H (the Go code indexes out of range). */
int aud_segment_workspace_bytes(const aud_plan* plan, int n_items, int64_t* bytes);
int aud_segment_batch_dev(aud_plan* plan, const void* sig, int sig_dtype, const aud_item* items, int n_items,
                          float* mel, float* power, float* log_power, float* mfcc, float* deltas,
                          float* delta_deltas, float* energy, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- hot path, host buffers (what the cgo shim binds) ---------------------------- */

/* Same as aud_melspec_batch_dev on host memory: sig float64 (SndEnv.Signal values), items on
 * the host, outputs float64 [n_items, nf, T] / [n_items, H, T] (NULL to skip). */
int aud_melspec_batch_host(aud_plan* plan, const double* sig, int64_t sig_total,
                           const aud_item* items, int n_items, double* mel, double* power,
                           double* log_power);

/* A signal kept RESIDENT on the device between calls.  SndEnv.ProcessSegment runs once per segment on the SAME Signal
 * tensor (sound/sndenv.go:342-359: every step reads se.Signal.Values), and the host entry points above move the whole
 * tensor over the link again on every call -- for 256 s of float64 audio that is 0.6 of the call's 0.9 ms.  Upload once
 * (after ToTensor / Pad / AdjustForSilence, sound/sndenv.go:274-300), then call the _sig variants per segment or batch.
 *   samples: host; sample_dtype AUD_F64 (the Signal tensor's values), AUD_F32, or AUD_I16: the WAV's own 16-bit PCM,
 *   normalised /0x7FFF on the device exactly as sound.go:138 does in float64 (2 bytes per sample over the link).
 * The copy is a snapshot: re-upload after changing the samples.  aud_signal_destroy waits for the context's stream. */
/* Result tensors in PINNED host memory.  The host entry points return float64 tensors (the Go tensors' type): the float32
 * results cross the link into a staging buffer and this thread widens them into the caller's memory -- for 256 utterances
 * a third of the call.  Tensors that live in memory from aud_host_alloc are written by the DEVICE instead: a kernel widens
 * the float32 results and stores the float64 values straight into them over the link (no staging copy, no CPU pass).  Every
 * _host / _sig entry point takes that route when ALL the output tensors it is given lie in such memory, and the staging route
 * otherwise; the values are the same.  A Go caller points etensor.Float64.Values at it (unsafe.Slice over C memory is
 * within the cgo rules; go/sound does).  aud_host_free after the last call that writes the memory; aud_shutdown frees what
 * is left. */
int aud_host_alloc(aud_ctx* ctx, int64_t bytes, void** ptr);
int aud_host_free(aud_ctx* ctx, void* ptr);
/* The same for memory the CALLER owns: aud_host_register pins [ptr, ptr + bytes) and makes it device-visible; result tensors
 * inside it are then written by the device like those of aud_host_alloc.  Meant for a mapping that SEVERAL processes share
 * (POSIX shared memory): one process per GPU, each registering the mapping and passing its shard's slice of one
 * [B, ...] float64 tensor as the output of its _host / _sig calls -- the features of the whole batch end in one host tensor
 * with no collective and no copy on the host (the host mode of the reference's multi-GPU use: sound/sndenv.go has no
 * counterpart).  aud_host_unregister before unmapping; aud_shutdown unregisters what is left. */
int aud_host_register(aud_ctx* ctx, void* ptr, int64_t bytes);
int aud_host_unregister(aud_ctx* ctx, void* ptr);
typedef struct aud_signal aud_signal;
int aud_signal_upload(aud_ctx* ctx, const void* samples, int sample_dtype, int64_t n_samples, aud_signal** out);
int aud_signal_destroy(aud_signal* sig);
int64_t aud_signal_len(const aud_signal* sig);
/* The EXACT form of residency -- what the host mirrors of SndEnv use by default.  The reference reads the live tensor at every
 * step (sound/sndenv.go:455-478: SndToWindow slices se.Signal.Values), so a caller may edit samples in place between two
 * ProcessSegment calls and must get the edited signal's features.  aud_signal_sync makes the device copy equal to `samples`:
 * the handle keeps a host SHADOW of what the device holds, the caller's tensor is compared with it byte for byte
 * (memcmp over 4 KB blocks: 20 us for a 3 s sound), and the span from the first to the last differing block -- nothing when
 * they are equal, everything on the first call or when type or length changed -- is uploaded and copied into the shadow.
 *   *sig: NULL on the first call (the handle is created), the handle afterwards; uploaded_bytes (may be NULL): what crossed
 *   the link.  No probability involved: any edit, anywhere, is seen.
 * The compare costs host time in proportion to the tensor and the shadow doubles its host memory, so the mirrors use it up
 * to AUD_RESIDENT_AUTO_BYTES of samples; above that they copy per call unless the caller opts in to a resident copy it
 * keeps current itself (SndEnv.SignalToDevice / SignalChanged: aud_signal_upload, a snapshot). */
#define AUD_RESIDENT_AUTO_BYTES (8 << 20)
int aud_signal_sync(aud_ctx* ctx, aud_signal** sig, const void* samples, int sample_dtype, int64_t n_samples,
                    int64_t* uploaded_bytes);
/* SndEnv.ProcessSegment on the LIVE Signal tensor (sound/sndenv.go:342-359, :455-478) in ONE call: aud_signal_sync restricted to
 * the 4 KB blocks the frames of `items` read under this plan (start0 - S Border ... start0 + S (T - 1 - Border) + N of each
 * stream), then aud_melspec_batch_sig / aud_melspec_mfcc_batch_sig on the resident copy.  The result is exactly what the
 * copy-per-call entries give on `samples` as they are NOW; the cost of being exact is a compare of what the call reads -- 16 KB
 * for one 100 ms segment of a sound, whatever the sound's length -- and an edit elsewhere in the tensor is found by the call
 * that reads it.  *sig as in aud_signal_sync (NULL at first); samples: float64 [n_samples], the Signal tensor's Values. */
int aud_melspec_batch_live(aud_plan* plan, aud_signal** sig, const double* samples, int64_t n_samples, const aud_item* items,
                           int n_items, double* mel, double* power, double* log_power, int64_t* uploaded_bytes);
int aud_melspec_mfcc_batch_live(aud_plan* plan, aud_signal** sig, const double* samples, int64_t n_samples,
                                const aud_item* items, int n_items, double* mel, double* power, double* log_power,
                                double* mfcc, double* deltas, double* delta_deltas, double* energy, int64_t* uploaded_bytes);
/* aud_melspec_batch_host / aud_melspec_mfcc_batch_host on a resident signal (items index ITS samples) */
int aud_melspec_batch_sig(aud_plan* plan, const aud_signal* sig, const aud_item* items, int n_items, double* mel,
                          double* power, double* log_power);
int aud_melspec_mfcc_batch_sig(aud_plan* plan, const aud_signal* sig, const aud_item* items, int n_items, double* mel,
                               double* power, double* log_power, double* mfcc, double* deltas, double* delta_deltas,
                               double* energy);

/* aud_segment_batch_dev on host memory (SndEnv.ProcessSegment with Mel.MFCC on):
 * outputs float64; power / log_power / deltas / delta_deltas / energy may be NULL. */
int aud_melspec_mfcc_batch_host(aud_plan* plan, const double* sig, int64_t sig_total, const aud_item* items,
                                int n_items, double* mel, double* power, double* log_power, double* mfcc,
                                double* deltas, double* delta_deltas, double* energy);

/* ---- the reference's per-step API, one frame per call ---------------------------------------
 * Kept so that code written against dft.Params.Filter / mel.Params.FilterDft / SndEnv.ProcessStep
 * keeps working unchanged.  Each call is a GPU round trip for a single frame (hundreds of
 * microseconds for microseconds of work): correct, never fast -- batch at ProcessSegment level. */

/* SndEnv.SndToWindow, sound/sndenv.go:455-478: window[N] <- signal[start, start+N) with left zero pad;
 * AUD_ESHORT ("end beyond signal length") if start + N > sig_len.  Host-side, no arithmetic. */
int aud_snd_to_window(const double* signal, int64_t sig_len, int64_t start, int win_samples, double* window);

/* dft.Params.Filter (+FftReal, Power), dft/dft.go:42-85, for one step: window [N]; power [H] is the
 * carry of the previous step on entry (used when step > 0 and PrevSmooth != 0) and this step's power on
 * return; log_power [H]; power_seg / log_power_seg [H, T] get column `step` (log_* may be NULL). */
int aud_dft_filter_host(aud_plan* plan, int step, const double* window, double* power, double* log_power,
                        double* power_seg, double* log_power_seg);

/* dft.Params.Power, dft/dft.go:62-85, for one step, on coefficients the caller computed: fft_coefs is complex128
 * [>= H] as (re, im) pairs; the other arguments as aud_dft_filter_host. */
int aud_dft_power_host(aud_plan* plan, int step, const double* fft_coefs, double* power, double* log_power,
                       double* power_seg, double* log_power_seg);

/* mel.Params.CepstrumDct, mel/mel.go:192-212, for one step (plan created with mfcc_coefs = NCoefs): fbank [nf];
 * mfcc_seg [NCoefs, T] gets column `step`; mfcc_dct [nf] (may be NULL) ends as a copy of fbank, as in the Go code. */
int aud_cepstrum_dct_host(aud_plan* plan, int step, const double* fbank, double* mfcc_seg, double* mfcc_dct);

/* mel.Params.FilterDft, mel/mel.go:120-153, for one step: power [H]; segment [nf, T] gets column `step`,
 * fbank [nf] the same values. */
int aud_mel_filter_dft_host(aud_plan* plan, int step, const double* power, double* segment, double* fbank);

/* agabor.Convolve on host memory: mel float64 [n_items, rows, cols], out float32 in/out. */
int aud_gabor_batch_host(aud_plan* plan, const double* mel, int n_items, int mel_rows,
                         int mel_cols, int out_rank, const int32_t* out_shape, int by_time,
                         float* out);

/* ---- k-WTA settling of the gabor output ---------------------------------------------------
 * SndEnv.ApplyKwta, sound/sndenv.go:313-323, which ApplyGabor (:481-497) runs when Kwta.On: the
 * activation tensor starts as a copy of the raw gabor output (:315) and settles under layer- and
 * pool-level FFFB inhibition.  The algorithm is third-party code that is not in the reference tree:
 * kwta.KWTA / KWTAPool / KWTALayer of github.com/emer/vision v1.1.15 over fffb.Params, nxx1.Params and
 * chans.Chans of github.com/emer/leabra v1.1.48 (go.mod:8-9).  The structs below carry those types'
 * exported fields one for one so that a Go binding passes se.Kwta through unchanged; the defaults
 * function restates KWTA.Defaults() as far as it is known here ("k-WTA").
 * NeighInhib (sndenv.go:303-311) is not built: the external-inhibition tensor is taken as all zeros,
 * which is what the reference has whenever NeighInhib.On is false (its zero value; :484-488). */
typedef struct {
    int32_t on;
    float gi, ff, fb, fb_tau, max_vs_avg, ff0;
} aud_fffb_params; /* leabra fffb.Params */

typedef struct {
    float thr, gain, nvar, vm_act_thr, sig_mult, sig_mult_pow, sig_gain, interp_range, gain_cor_range, gain_cor;
} aud_nxx1_params; /* leabra nxx1.Params (settable fields; the derived ones are recomputed per call) */

typedef struct {
    int32_t on;    /* KWTA.On: informational here -- the caller decides whether to call */
    int32_t iters; /* KWTA.Iters */
    float del_act_thr;
    aud_fffb_params lay_fffb, pool_fffb;
    aud_nxx1_params xx1;
    float act_tau;
    float gbar[4]; /* chans.Chans E, L, I, K */
    float erev[4];
} aud_kwta_params;

void aud_kwta_defaults(aud_kwta_params* k);

/* raw, act: float32 [n_items, d0, d1, d2, d3] in device memory, act != raw.
 * pool_level 1: KWTAPool (layer level over all values + pool level inside each (d0, d1) cell);
 *            0: KWTALayer (layer level only; the four extents only give the value count).
 * start_from_raw 1: act is output only and settling starts from act = raw (what ApplyKwta does);
 *                0: act holds the starting activations on entry.
 * pool_state: float32 [n_items, d0*d1, 2] = {FBi, Act.Avg} of every pool's fffb.Inhib, read on entry
 *   and written on return -- the SndEnv.Inhibs slice (sndenv.go:166) that KWTAPool carries from call to
 *   call; NULL = a fresh slice.  Ignored at layer level.
 * sum_order 0: running float32 sums in the reference's index order (bit-faithful, the layer sum is
 *   sequential); 1: fixed reduction tree (deterministic, differs from the reference's sums in the last ulps).
 * cycles: int32 [n_items] settling cycles each item ran, or NULL.
 * One workgroup per item with the activations in LDS: needs (32 + n + 4 d0 d1) * 4 bytes <= 160 KB. */
int aud_kwta_batch_dev(aud_ctx* ctx, const aud_kwta_params* k, const float* raw, float* act, int n_items, int d0,
                       int d1, int d2, int d3, int pool_level, int start_from_raw, float* pool_state,
                       int sum_order, int32_t* cycles, void* stream);
/* the same on host memory (copies in and out on the context's stream, synchronous) */
int aud_kwta_batch_host(aud_ctx* ctx, const aud_kwta_params* k, const float* raw, float* act, int n_items, int d0,
                        int d1, int d2, int d3, int pool_level, int start_from_raw, float* pool_state,
                        int sum_order, int32_t* cycles);

/* ---- multi-GPU reassembly (one process per GPU, RCCL over xGMI) ------------------ */

/* 128-byte RCCL unique id, created on rank 0 and distributed by the host program */
int aud_comm_unique_id(char id[128]);
int aud_comm_init(aud_ctx* ctx, int n_ranks, int rank, const char id[128]);
int aud_comm_destroy(aud_ctx* ctx);
/* in-place capable all-gather of `count` float32 per rank: recv[rank*count ...] <- send */
int aud_allgather_dev(aud_ctx* ctx, const float* send, float* recv, int64_t count, void* stream);

/* The same reassembly as DIRECT device-to-device copies (the 8 GPUs of a node are fully connected, 7 xGMI
 * links per GPU, so an all-gather is one push per peer, each on its own link and its own stream -- what a ring collective,
 * per-link bound with 7 serial hops, cannot do for messages of a few MB).  Set-up, once per (context, shape):
 *   aud_gather_create   allocates this rank's receive area -- TWO slabs [2][n_ranks][slab_floats] float32 that consecutive
 *                       steps alternate between -- and a block of arrival flags on the device, and exports their two 64-byte
 *                       inter-process handles (128 bytes); the host program distributes the handles (any channel)
 *   aud_gather_open_peer  maps peer `peer`'s receive area and flags from its handles (not needed, and refused, for peer == rank)
 * Per step (every rank makes the same sequence of calls):
 *   aud_allgather_direct_dev  copies send[0 .. count) into slot `rank` of the step's slab of EVERY rank (its own included):
 *                       n_ranks - 1 peer copies on n_ranks - 1 internal streams forked from and joined back into `stream`, each
 *                       followed by a one-thread kernel that stores this rank's step number into the peer's flag for it
 *                       (system-scope release).  count <= slab_floats.  *slab_index (may be NULL) = the slab this step
 *                       uses: 0, 1, 0, ... per call.  When `stream` has passed the call THIS rank's pushes are done.
 *   aud_gather_wait_dev  the ARRIVAL side of the same step: a kernel on `stream` polls this rank's flags until every peer's
 *                       has reached this rank's step number, i.e. every peer's slab of the step has landed -- behind it the
 *                       step's slab is complete and kernels queued on `stream` may read it (what ncclAllGather's return means).
 *                       The poll is bounded (default 30 s; AUD_GATHER_WAIT_MS sets another bound): a peer that never arrives
 *                       ends the kernel -- never a hung queue -- and the time-out is STICKY and VISIBLE: the kernel fills the
 *                       missing peers' slots of the step's slab with NaN (what is queued behind the wait computes NaN, not
 *                       plausible numbers from a stale slab), counts the time-out, and sets a status word in host-mapped memory;
 *                       the next aud_allgather_direct_dev / aud_gather_wait_dev on this context returns AUD_EBROKEN without
 *                       queueing anything (no synchronisation needed to notice), and so does every later one until
 *                       aud_gather_destroy / aud_gather_create.  aud_gather_timeouts reads the count (it synchronises).
 *   Both are capturable into a hipGraph (step numbers live in device memory and advance per replay; a captured sequence
 *   must hold an EVEN number of steps so that the slabs keep alternating across replays: the library counts the calls of a
 *   capture and the first call behind an odd one returns AUD_EBROKEN).  Reuse: the peers' step i + 2
 *   overwrites the slab of step i, and a peer cannot start step i + 2 before this rank has signalled step i + 1 -- so
 *   everything that reads the slab of step i must be ordered on `stream` before this rank's call of step i + 1.
 *   A call that fails behind its first enqueue leaves this rank's step count out of step with its peers': the gather is
 *   marked broken (AUD_EBROKEN from then on) instead of running on with slabs and step numbers that no longer match.
 *   The arrival flags need fine-grained (coherent) device memory that can be exported to other processes; where the runtime has
 *   none aud_gather_create FAILS (peers' stores into ordinary device memory are not guaranteed to become visible to a kernel
 *   that polls through its L2) unless AUD_GATHER_COARSE_FLAGS=1 is set; aud_gather_flags_fine says which was taken.
 *   aud_gather_destroy  unmaps the peers, frees the buffers. */
int aud_gather_create(aud_ctx* ctx, int n_ranks, int rank, int64_t slab_floats, float** recv, char handle[128]);
int aud_gather_open_peer(aud_ctx* ctx, int peer, const char handle[128]);
int aud_allgather_direct_dev(aud_ctx* ctx, const float* send, int64_t count, int* slab_index, void* stream);
int aud_gather_wait_dev(aud_ctx* ctx, void* stream);
/* waits that ended on their poll bound since aud_gather_create (synchronises `ctx`'s device first) */
int aud_gather_timeouts(aud_ctx* ctx, int* n);
/* 1: the arrival flags live in fine-grained memory; 0: ordinary device memory (AUD_GATHER_COARSE_FLAGS=1); -1: no gather */
int aud_gather_flags_fine(aud_ctx* ctx);
int aud_gather_destroy(aud_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* AUDITORY_HIP_H */
