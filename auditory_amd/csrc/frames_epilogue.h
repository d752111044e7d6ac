// The part of the any-N kernels behind the power spectrum: optional PowerSegment / LogPowerSegment outputs, the fused segment
// tail's Energy sums, the mel triangles + log, and the fused tail's CepstrumDct rows.  Shared by k_melspec_generic
// (melspec_generic.hip) and the fixed-geometry chirp kernel of L = 2304 (melspec_chirp.hip): both leave the workgroup's F
// power spectra as P[f][k] (row pitch H | 1) in LDS, synchronised, with room for F x nf values behind them.
//
// Reference semantics: dft/dft.go:62-85 (power, log(power + offset)), mel/mel.go:120-153 (triangle sums, log, renorm),
// sound/sndenv.go:360-372 (Energy with its axis quirk), mel/mel.go:192-212 (CepstrumDct).
#pragma once
#include "device_common.h"

namespace aud {

// ln v for the epilogue's values.  float32 plans: logf.  float64 plans: v = m 2^e with sqrt(1/2) <= m < sqrt(2),
// ln m = 2 atanh(t), t = (m - 1) / (m + 1), |t| <= 0.1716 -- the odd series through t^15 (what is dropped: < 3e-13), the quotient
// by the hardware's reciprocal with two Newton steps and one residual correction.  ~35 vector instructions against the ~100
// of the library's log(); absolute error < 1e-12 -- these values are stored as float32 (6e-8) or feed sums that are compared at
// 3e-7 (Energy, the DCT rows).  v = 0 never arrives (the callers return LogMin for it: dft.go:79, mel.go:135); NaN, negative and
// infinite arguments take the library's route.  A whole ProcessSegment workgroup at N = 1103 takes 1 104 of these logarithms.
__device__ __forceinline__ float epi_log(float v) { return logf(v); }
__device__ __forceinline__ double epi_log(double v) {
    if (!(v > 0.0) || v > 1.0e308) return log(v);
    int e = 0;
    double m = frexp(v, &e);                                    // [1/2, 1)
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;                                        // [sqrt(1/2), sqrt(2))
    e -= low ? 1 : 0;
    const double num = m - 1.0, den = m + 1.0;                  // (exact: m has 53 bits in [0.7, 1.42))
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double t = num * r;
    t = fma(fma(-den, t, num), r, t);                           // t = num / den to ~1 ulp
    const double t2 = t * t;
    double p = 1.0 / 15.0;
    p = fma(p, t2, 1.0 / 13.0);
    p = fma(p, t2, 1.0 / 11.0);
    p = fma(p, t2, 1.0 / 9.0);
    p = fma(p, t2, 1.0 / 7.0);
    p = fma(p, t2, 1.0 / 5.0);
    p = fma(p, t2, 1.0 / 3.0);
    const double two_t = t + t;
    const double lnm = fma(two_t * t2, p, two_t);               // 2 t (1 + t^2 / 3 + ... + t^14 / 15)
    return fma(double(e), 0.693147180559945309417, lnm);
}

// WALK: how the mel loop deals its work -- true: a slot of four lanes keeps one filter and walks the workgroup's frames (the any-N
// and direct kernels: up to sixteen frames per workgroup); false: a slot per (frame, filter) pair (the chirp kernel: two frames,
// one pass either way, and this form is 3 % faster there: profiles/round6_epilogue_walk_ab.txt)
template <typename TT, bool WALK = true>
__device__ __forceinline__ void frames_epilogue(const MelspecArgs& a, const aud_item& it, int item, int tiles, int t0, TT* P, int tid) {
    const int F = a.F, N = a.N, H = a.H, T = a.T;
    const int Hp = H | 1;
    // ---- optional PowerSegment / LogPowerSegment (dft.go:70-83) -----------------------
    if (a.power || a.log_power) {
        const TT off = TT(a.dft_log_off), lmin = TT(a.dft_log_min);
        const int f_log = 31 - __builtin_clz(unsigned(F));  // (frames per workgroup are a power of two on every route: 1 .. 16)
        // the item's [H][T] tensors as pointers, 32-bit offsets inside them (the loop had carried two 64-bit products and a
        // division by F per value)
        float* __restrict__ pw_out = a.power ? a.power + size_t(item) * H * T : nullptr;
        float* __restrict__ lp_out = a.log_power ? a.log_power + size_t(item) * H * T : nullptr;
        for (int w = tid; w < F * H; w += blockDim.x) {
            const int k = w >> f_log, f = w - (k << f_log);
            const int sstep = t0 + f;
            if (sstep >= T) continue;
            const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
            const bool live = start + N <= int64_t(it.sig_len);
            const TT pw = P[f * Hp + k];
            const int o = k * T + sstep;
            if (pw_out) pw_out[o] = live ? float(pw) : 0.f;
            if (lp_out) {
                float lp = 0.f;
                if (live && a.comp_log_pow) {
                    const TT v = pw + off;
                    lp = float(v == TT(0) ? lmin : epi_log(v));
                }
                lp_out[o] = lp;
            }
        }
    }

    // ---- fused segment tail, part 1 (aud_segment_batch_dev; sndenv.go:360-366 with its axis quirk): Energy[s] sums
    // the log-power of BIN s over the steps of the segment -- this workgroup's share, from the UNROUNDED values, for s < T
    if (a.energy_part) {
        const TT off = TT(a.dft_log_off), lmin = TT(a.dft_log_min);
        TT* ep = static_cast<TT*>(a.energy_part) + (size_t(item) * tiles + size_t(t0 / F)) * T;
        for (int s = tid; s < T; s += blockDim.x) {  // (T <= H: the host checks, the Go code panics otherwise)
            TT e = TT(0);
            for (int f = 0; f < F; ++f) {
                const int sstep = t0 + f;
                const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
                if (sstep < T && start + N <= int64_t(it.sig_len)) {  // (a step the loop never reached left LogPowerSegment at 0)
                    const TT v = P[size_t(f) * Hp + s] + off;
                    e += v == TT(0) ? lmin : epi_log(v);
                }
            }
            ep[s] = e;
        }
    }

    // ---- mel triangles + log (mel.go:120-153) ---------------------------------------
    TT* melL = P + size_t(F) * Hp;  // fused tail: the workgroup's [F][nf] log-mel values before their float32 rounding
    {
        const TT* __restrict__ filt = static_cast<const TT*>(a.filt);
        const int cols = a.nf + 2;
        const TT loff = TT(a.mel_log_off), lmin = TT(a.mel_log_min);
        if constexpr (WALK) {
            // FOUR lanes per filter, a lane takes every fourth tap; the four partial sums meet by two lane exchanges, (p0 + p2) +
            // (p1 + p3) on every lane.  A slot (= four lanes) keeps ONE filter and walks the workgroup's frames with it: the filter's
            // bin range and tap weights are requested once, together, before the first is used -- as a loop over (frame, filter)
            // pairs every pass waited for its own table loads, two L2 round trips in a row, F / 2 passes (a third of the kernel at
            // N = 200, F = 8: profiles/round6_plain_inplace_ablation.txt).  With fewer filters than slots the slots split the frames
            // between them (G groups: 32 filters, 64 slots: even and odd frames).  Same products, same order of additions.
            const int slots = int(blockDim.x) >> 2, slot = tid >> 2, part = tid & 3;
            const int G = a.nf < slots ? slots / a.nf : 1;            // frame groups of a pass
            const int per_group = (F + G - 1) / G;                    // frames a slot walks (uniform trip count: whole waves stay in the exchanges)
            for (int base = 0; base < a.nf; base += slots) {
                const int g = G > 1 ? slot / a.nf : 0;
                const int flt_raw = base + (G > 1 ? slot - g * a.nf : slot);
                const bool on = flt_raw < a.nf && g < G;
                const int flt = on ? flt_raw : a.nf - 1;
                const int lo = a.bin_pts[flt], hi = a.bin_pts[flt + 2];
                const TT* wrow = filt + size_t(flt) * cols;
                constexpr int kJ = 9;  // taps of a lane where a table row has at most 36 columns (nf <= 34: the reference's 32)
                const bool batched = cols <= 4 * kJ;
                TT wv[kJ];
    #pragma unroll
                for (int j = 0; j < kJ; ++j) {
                    const int bin = lo + part + 4 * j;
                    wv[j] = batched ? wrow[bin <= hi ? bin - lo : 0] : TT(0);
                }
                for (int i = 0; i < per_group; ++i) {
                    const int f_raw = g + i * G;
                    const bool valid = on && f_raw < F;
                    const int f = f_raw < F ? f_raw : F - 1;
                    const int sstep = t0 + f;
                    const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
                    const bool live = sstep < T && start + N <= int64_t(it.sig_len);
                    const TT* prow = P + size_t(f) * Hp;
                    TT sum = TT(0);
                    if (live && batched) {
    #pragma unroll
                        for (int j = 0; j < kJ; ++j) {
                            const int bin = lo + part + 4 * j;
                            if (bin <= hi) sum += wv[j] * prow[bin];
                        }
                    } else if (live) {
                        for (int bin = lo + part; bin <= hi; bin += 4) sum += wrow[bin - lo] * prow[bin];
                    }
                    sum += __shfl_xor(sum, 2, 64);
                    sum += __shfl_xor(sum, 1, 64);
                    if (part != 0 || !valid) continue;
                    float res = 0.f;
                    TT val = TT(0);
                    if (live) {
                        sum += loff;
                        val = (sum == TT(0)) ? lmin : epi_log(sum);
                        if (a.renorm) {
                            val -= TT(a.renorm_min);
                            if (val < TT(0)) val = TT(0);
                            val *= TT(a.renorm_scale);
                            if (val > TT(1)) val = TT(1);
                        }
                        res = float(val);
                    }
                    if (a.mfcc_acc) melL[f * a.nf + flt] = val;  // (0 for a step the loop never reached: its MFCC column stays 0)
                    if (sstep < T) a.mel[(size_t(item) * a.nf + flt) * T + sstep] = res;
                }
            }
        } else {
            // FOUR lanes per (frame, filter): the workgroup has few frames (one or two with Bluestein) and a filter's taps are a
            // serial chain of table loads -- one lane per filter left three of the four waves idle through the kernel's tail.  A
            // lane takes every fourth tap; the four partial sums meet by two lane exchanges, (p0 + p2) + (p1 + p3) on every lane.
            const int n_work = F * a.nf;
            for (int w0 = tid >> 2; w0 < ((n_work + 63) & ~63); w0 += blockDim.x >> 2) {  // (whole waves stay in the exchanges)
                const int w = w0 < n_work ? w0 : n_work - 1, part = tid & 3;
                const int flt = w / F, f = w - flt * F;
                const int sstep = t0 + f;
                const int64_t start = int64_t(it.start0) + int64_t(a.S) * (sstep - a.border);
                const bool live = sstep < T && start + N <= int64_t(it.sig_len);
                TT sum = TT(0);
                {
                    const int lo = a.bin_pts[flt], hi = a.bin_pts[flt + 2];
                    const TT* wrow = filt + size_t(flt) * cols;
                    const TT* prow = P + size_t(f) * Hp;
                    constexpr int kJ = 9;  // taps of a lane where a table row has at most 36 columns (nf <= 34: the reference's 32)
                    if (live && cols <= 4 * kJ) {
                        // every tap's weight requested before the first is used: as a loop with a data-dependent trip count each
                        // iteration waited for its own table load -- five to nine L2 round trips in a row (round 6).  Same products,
                        // same order of additions
                        TT wv[kJ];
    #pragma unroll
                        for (int j = 0; j < kJ; ++j) {
                            const int bin = lo + part + 4 * j;
                            wv[j] = wrow[bin <= hi ? bin - lo : 0];
                        }
    #pragma unroll
                        for (int j = 0; j < kJ; ++j) {
                            const int bin = lo + part + 4 * j;
                            if (bin <= hi) sum += wv[j] * prow[bin];
                        }
                    } else if (live) {
                        for (int bin = lo + part; bin <= hi; bin += 4) sum += wrow[bin - lo] * prow[bin];
                    }
                }
                sum += __shfl_xor(sum, 2, 64);
                sum += __shfl_xor(sum, 1, 64);
                if (part != 0 || w0 >= n_work) continue;
                float res = 0.f;
                TT val = TT(0);
                if (live) {
                    sum += loff;
                    val = (sum == TT(0)) ? lmin : epi_log(sum);
                    if (a.renorm) {
                        val -= TT(a.renorm_min);
                        if (val < TT(0)) val = TT(0);
                        val *= TT(a.renorm_scale);
                        if (val > TT(1)) val = TT(1);
                    }
                    res = float(val);
                }
                if (a.mfcc_acc) melL[f * a.nf + flt] = val;  // (0 for a step the loop never reached: its MFCC column stays 0)
                if (sstep < T) a.mel[(size_t(item) * a.nf + flt) * T + sstep] = res;
            }
        }
    }

    // ---- fused segment tail, part 2: mel.Params.CepstrumDct (mel.go:192-212) on the unrounded log-mel values, rows 1 .. (row 0 is
    // overwritten with Energy, sndenv.go:368-372: launch_segment_finish does that and the deltas)
    if (a.mfcc_acc) {
        __syncthreads();
        const TT* __restrict__ D = static_cast<const TT*>(a.dct_rows);
        TT* acc = static_cast<TT*>(a.mfcc_acc) + size_t(item) * a.n_coefs * T;
        for (int w = tid; w < F * (a.n_coefs - 1); w += blockDim.x) {
            const int c = 1 + w / F, f = w - (c - 1) * F, sstep = t0 + f;
            if (sstep >= T) continue;
            const TT* drow = D + size_t(c) * a.nf;
            const TT* mrow = melL + f * a.nf;
            TT sum = TT(0);
            for (int j = 0; j < a.nf; ++j) sum = mad(drow[j], mrow[j], sum);
            acc[size_t(c) * T + sstep] = sum;
        }
    }
}

}  // namespace aud
