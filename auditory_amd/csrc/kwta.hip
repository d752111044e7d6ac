// k-WTA settling of the gabor output: SndEnv.ApplyKwta (sound/sndenv.go:313-323), i.e. KWTAPool /
// KWTALayer of github.com/emer/vision v1.1.15 (kwta/kwta.go) on top of leabra v1.1.48's FFFB inhibition
// (fffb/fffb.go) and noisy x/(x+1) rate code (nxx1/nxx1.go).  Those modules are not part of the
// reference tree; the algorithm is restated from their published sources ("k-WTA", says what
// anchors it and what is unverified).
//
// One 512-thread workgroup settles one item.  All arithmetic is float32 in the reference's operation order with
// contraction off, FastExp is the integer bit trick of goki/mat32, so in sum_order 0 the result is meant
// to equal the reference's bit for bit.  What cannot be parallelised under that constraint is the
// layer-level running sum (a float32 accumulation over every value in index order): thread 0 does it from
// LDS while the others wait (about 4 cycles per value and settling cycle).  sum_order 1 replaces it by a
// fixed reduction tree over per-thread partial sums: deterministic, a few ulp away from the reference's
// sum, and the whole workgroup takes part.
//
// In the pool-level mode the serial sum skips exact zeros.  That is exact: the running sum starts at +0 and can
// never become -0 (x + y is -0 only for x = y = -0), and s + (+-0) == s for every other s, so the running sum
// over the non-zero activations, taken in index order, equals the running sum over all of them bit for bit.
// The rectified gabor output is exactly zero in half of its cells and those stay zero while they settle, so an
// order-preserving compaction (per-pool counts, a wave scan, scattered writes: all parallel) halves the serial
// section.  It needs n more floats of LDS; when they do not fit, the sum walks every value.
//
// Per settling cycle: layer FFFB (thread 0) -> every pool, one thread each: pool FFFB, gi = max(layer,
// pool), threshold, the pool's units in order (noisy XX1, ActDt integration, running pool sum) -> max |dAct|
// and the layer sum -> stop when cycle > 2 and max |dAct| < DelActThr.  Activations live in LDS.
#include "kernels.h"

#pragma clang fp contract(off)

namespace aud {
namespace {

constexpr int kNT = 512;   // 352 pools of a 1 s segment in one round; two waves per SIMD keep the vector pipe fed
constexpr int kNW = kNT / 64;
constexpr int kCtrl = 32;  // control words at the head of dynamic LDS (floats)
// ctrl[0] layer gi   ctrl[1] stop flag   ctrl[2] layer FBi   ctrl[3] layer Ge.Avg   ctrl[4] layer Ge.Max
// ctrl[5] layer Act.Avg   ctrl[8..15] per-wave max |dAct|   ctrl[16..23] per-wave partial sums

// goki/mat32 FastExp (Schraudolph's quartic spline on the float32 bit pattern); |arg| stays far inside
// the int32 range here (callers pass 0 < x <= 50)
__device__ __forceinline__ float fast_exp(float x) {
    if (x <= -88.76731f) return 0.f;
    int32_t i = int32_t(12102203.0f * x) + 127 * (1 << 23);
    const int32_t m = (i >> 7) & 0xFFFF;
    i += (((((((((((3537 * m) >> 16) + 13668) * m) >> 18) + 15817) * m) >> 14) - 80470) * m) >> 11);
    return __int_as_float(i);
}

__device__ __forceinline__ float xx1(float x) { return x / (x + 1.f); }

__device__ __forceinline__ float xx1_gain_cor(const KwtaArgs& a, float x) {
    const float fact = (a.gain_cor_range - (x / a.nvar)) / a.gain_cor_range;
    if (fact < 0.f) return xx1(a.gain * x);
    const float new_gain = a.gain * (1.f - a.gain_cor * fact);
    return xx1(new_gain * x);
}

__device__ __forceinline__ float noisy_xx1(const KwtaArgs& a, float x) {
    if (x < 0.f) {
        const float ex = -(x * a.sig_gain_nvar);
        if (ex > 50.f) return 0.f;
        return a.sig_mult_eff / (1.f + fast_exp(ex));
    } else if (x < a.interp_range) {
        const float interp = 1.f - ((a.interp_range - x) / a.interp_range);
        return a.sig_val_at0 + interp * a.interp_val;
    }
    return xx1_gain_cor(a, x);
}

// fffb.Params.Inhib on the fields that feed back into the computation
__device__ __forceinline__ void fffb_inhib(const KwtaFffb& p, float ge_avg, float ge_max, float act_avg, float& fbi,
                                           float& gi) {
    if (!p.on) {
        fbi = 0.f;
        gi = 0.f;
        return;
    }
    const float ff_netin = ge_avg + p.max_vs_avg * (ge_max - ge_avg);
    float ffi = 0.f;
    if (ff_netin > p.ff0) ffi = p.ff * (ff_netin - p.ff0);
    const float nfb = p.fb * act_avg;
    fbi += p.fb_dt * (nfb - fbi);
    gi = p.gi * (ffi + fbi);
}

__device__ __forceinline__ float ge_thr_from_g(const KwtaArgs& a, float gi) {
    return (a.gbar_i * gi * a.erev_sub_thr_i + a.gbar_l * a.erev_sub_thr_l) / a.thr_sub_erev_e;
}

__device__ __forceinline__ float unit_update(const KwtaArgs& a, float ge, float ge_thr, float ac, float& mx) {
    float nw = noisy_xx1(a, ge * a.gbar_e - ge_thr);
    const float del = a.act_dt * (nw - ac);
    nw = ac + del;
    mx = fmaxf(mx, fabsf(del));
    return nw;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// fixed tree: lane pairs at distance 32, 16, ..., 1 (the same association on every run)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o);
    return v;
}

// the per-wave slots combined in a fixed order
__device__ __forceinline__ float wave_slots_sum(const float* v) {
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}
__device__ __forceinline__ float wave_slots_max(const float* v) {
    return fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
}
static_assert(kNW == 8, "wave_slots_* combine 8 slots");

// AvgMax32.CalcAvg
__device__ __forceinline__ float calc_avg(float sum, int n) { return n > 0 ? sum / float(n) : sum; }

__global__ __launch_bounds__(kNT) void k_kwta(const KwtaArgs a) {
    float* ctrl = reinterpret_cast<float*>(dyn_lds());
    const int n = a.n, lay_n = a.lay_n, pl_n = a.pl_n;
    const int n_pad = (n + 3) & ~3;
    float* acts = ctrl + kCtrl;     // [n]
    float* p_ge_avg = acts + n_pad;  // [lay_n] x 4
    float* p_ge_max = p_ge_avg + lay_n;
    float* p_fbi = p_ge_max + lay_n;
    float* p_act_avg = p_fbi + lay_n;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int item = blockIdx.x;
    const bool tree = a.sum_order != 0;
    int* p_off = reinterpret_cast<int*>(p_act_avg + lay_n);  // [lay_n + 1] (compaction only)
    float* packed = reinterpret_cast<float*>(p_off + ((lay_n + 1 + 3) & ~3));  // [n] (compaction only)
    const bool compact = a.compact != 0 && lay_n > 0 && !tree;
    const float* __restrict__ raw = a.raw + size_t(item) * n;
    float* act_g = a.act + size_t(item) * n;

    // ---- Ge statistics: layer (all values, index order) and pools ----
    for (int i = tid; i < n; i += kNT) acts[i] = a.start_from_raw ? raw[i] : act_g[i];
    float gmax = -3.402823466e+38f, gpart = 0.f;
    if (lay_n > 0) {
        for (int pi = tid; pi < lay_n; pi += kNT) {
            float s = 0.f, m = -3.402823466e+38f;
            for (int ui = 0; ui < pl_n; ++ui) {
                const float ge = raw[pi * pl_n + ui];
                s += ge;
                m = ge > m ? ge : m;
            }
            gpart += s;
            gmax = fmaxf(gmax, m);
            p_ge_avg[pi] = calc_avg(s, pl_n);
            p_ge_max[pi] = pl_n > 0 ? m : p_ge_avg[pi];
            p_fbi[pi] = a.state ? a.state[(size_t(item) * lay_n + pi) * 2] : 0.f;
            p_act_avg[pi] = a.state ? a.state[(size_t(item) * lay_n + pi) * 2 + 1] : 0.f;
        }
    } else {
        for (int i = tid; i < n; i += kNT) {
            const float ge = raw[i];
            gpart += ge;
            gmax = fmaxf(gmax, ge);
        }
    }
    gmax = wave_max(gmax);
    gpart = wave_sum(gpart);
    if (lane == 0) {
        ctrl[8 + wave] = gmax;
        ctrl[16 + wave] = gpart;
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        if (tree) {
            s = wave_slots_sum(ctrl + 16);
        } else if (a.start_from_raw) {
            for (int i = 0; i < n; ++i) s += acts[i];  // acts == raw here
        } else {
            for (int i = 0; i < n; ++i) s += raw[i];
        }
        const float ge_avg = calc_avg(s, n);
        const float ge_max = n > 0 ? wave_slots_max(ctrl + 8) : ge_avg;
        float fbi = 0.f, gi = 0.f;  // a fresh layer-level fffb.Inhib on every call
        fffb_inhib(a.lay, ge_avg, ge_max, 0.f, fbi, gi);
        ctrl[0] = gi;
        ctrl[1] = 0.f;
        ctrl[2] = fbi;
        ctrl[3] = ge_avg;
        ctrl[4] = ge_max;
    }
    __syncthreads();

    int cy = 0;
    for (; cy < a.iters; ++cy) {
        const float lay_gi = ctrl[0];
        float mx = 0.f, part = 0.f;
        if (lay_n > 0) {
            for (int pi = tid; pi < lay_n; pi += kNT) {
                float fbi = p_fbi[pi], gi;
                fffb_inhib(a.pool, p_ge_avg[pi], p_ge_max[pi], p_act_avg[pi], fbi, gi);
                p_fbi[pi] = fbi;
                // the external-inhibition tensor is all zeros on this path: max(gi, Pool.Gi * FFInhib(0, 0)) = gi
                const float ge_thr = ge_thr_from_g(a, fmaxf(lay_gi, gi));
                float s = 0.f;
                int nz = 0;
                for (int ui = 0; ui < pl_n; ++ui) {
                    const int idx = pi * pl_n + ui;
                    const float nw = unit_update(a, raw[idx], ge_thr, acts[idx], mx);
                    s += nw;
                    nz += nw != 0.f ? 1 : 0;  // a NaN counts: it must reach the sum
                    acts[idx] = nw;
                }
                part += s;
                p_act_avg[pi] = calc_avg(s, pl_n);
                if (compact) p_off[pi] = nz;
            }
        } else {
            const float ge_thr = ge_thr_from_g(a, lay_gi);
            for (int i = tid; i < n; i += kNT) {
                const float nw = unit_update(a, raw[i], ge_thr, acts[i], mx);
                part += nw;
                acts[i] = nw;
            }
        }
        mx = wave_max(mx);
        part = wave_sum(part);
        if (lane == 0) {
            ctrl[8 + wave] = mx;
            ctrl[16 + wave] = part;
        }
        __syncthreads();
        int n_sum = n;
        const float* sum_src = acts;
        if (compact) {
            // exclusive scan of the per-pool non-zero counts in pool order (wave 0, 64 pools per step) ...
            if (wave == 0) {
                int base = 0;
                for (int p0 = 0; p0 < lay_n; p0 += 64) {
                    const int pi = p0 + lane;
                    const int c = pi < lay_n ? p_off[pi] : 0;
                    int incl = c;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int up = __shfl_up(incl, o);
                        if (lane >= o) incl += up;
                    }
                    if (pi < lay_n) p_off[pi] = base + incl - c;
                    base += __shfl(incl, 63);
                }
                if (lane == 0) p_off[lay_n] = base;
            }
            __syncthreads();
            // ... and every pool's non-zero activations to their place, order kept
            for (int pi = tid; pi < lay_n; pi += kNT) {
                int o = p_off[pi];
                for (int ui = 0; ui < pl_n; ++ui) {
                    const float v = acts[pi * pl_n + ui];
                    if (v != 0.f) packed[o++] = v;
                }
            }
            __syncthreads();
            n_sum = p_off[lay_n];
            sum_src = packed;
        }
        if (tid == 0) {
            const float max_del = wave_slots_max(ctrl + 8);
            float s = 0.f;
            if (tree) {
                s = wave_slots_sum(ctrl + 16);
            } else {
#pragma unroll 8
                for (int i = 0; i < n_sum; ++i) s += sum_src[i];
            }
            const float act_avg = calc_avg(s, n);
            ctrl[5] = act_avg;
            const bool stop = cy > 2 && max_del < a.del_act_thr;
            ctrl[1] = stop ? 1.f : 0.f;
            if (!stop) {
                float fbi = ctrl[2], gi;
                fffb_inhib(a.lay, ctrl[3], ctrl[4], act_avg, fbi, gi);
                ctrl[0] = gi;
                ctrl[2] = fbi;
            }
        }
        __syncthreads();
        if (ctrl[1] != 0.f) {  // uniform: every thread reads the same word after the barrier
            ++cy;
            break;
        }
    }

    for (int i = tid; i < n; i += kNT) act_g[i] = acts[i];
    if (a.state)
        for (int pi = tid; pi < lay_n; pi += kNT) {
            a.state[(size_t(item) * lay_n + pi) * 2] = p_fbi[pi];
            a.state[(size_t(item) * lay_n + pi) * 2 + 1] = p_act_avg[pi];
        }
    if (a.cycles && tid == 0) a.cycles[item] = cy;
}

}  // namespace

size_t kwta_lds_bytes(int n, int lay_n, bool compact) {
    size_t words = size_t(kCtrl) + size_t((n + 3) & ~3) + 4 * size_t(lay_n);
    if (compact) words += size_t((lay_n + 1 + 3) & ~3) + size_t(n);
    return words * sizeof(float);
}

hipError_t kwta_prepare(unsigned lds_bytes) {
    // the attribute belongs to the kernel, not to a call: only ever raised, to the device's limit
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kwta), hipFuncAttributeMaxDynamicSharedMemorySize,
                               lds_bytes > 0 ? 160 * 1024 : 0);
}

hipError_t launch_kwta(const KwtaArgs& a, hipStream_t st) {
    if (a.n_items == 0) return hipSuccess;
    hipLaunchKernelGGL(k_kwta, dim3(unsigned(a.n_items)), dim3(kNT), a.lds_bytes, st, a);
    return hipGetLastError();
}

}  // namespace aud
