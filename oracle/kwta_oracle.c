/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the k-WTA stage that follows the gabor convolution
 * in SndEnv.ApplyGabor (sound/sndenv.go:481-497 -> ApplyKwta :313-323).  Nothing in the product may
 * link, import or call this file; tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
 *
 * PARITY UNPINNED, and one step further from the source than the rest of the oracle: the algorithm
 * lives in third-party modules that are NOT under /root/reference and cannot be fetched offline --
 *     github.com/emer/vision v1.1.15   kwta/kwta.go      (KWTA, KWTALayer, KWTAPool)
 *     github.com/emer/leabra v1.1.48   fffb/fffb.go, fffb/inhib.go, nxx1/nxx1.go, chans/chans.go
 *     github.com/emer/etable v1.1.7    minmax/avgmax.go  (AvgMax32)
 *     github.com/goki/mat32  v1.0.12   fastexp.go        (FastExp)
 * (versions from the reference's go.mod:5-22).  What follows restates their published algorithm as
 * the author of this file knows it; it has not been compared with those sources in this image.  The
 * reference's own call sites are what anchor it: the activation tensor starts as a COPY of the raw
 * gabor output (sndenv.go:315), the external-inhibition tensor is all zeros unless NeighInhib.On
 * (:484-488), pools are the outer two dimensions of the 4-D gabor output (:318), and the pool-level
 * fffb.Inhibs slice lives in the SndEnv (:166) and is therefore carried from one call to the next.
 *
 * Everything here is float32, evaluated in the reference's order, no contraction (Go on amd64 does
 * not fuse a*b+c), so a faithful device implementation can be compared bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    int32_t on;
    float gi, ff, fb, fb_tau, max_vs_avg, ff0; /* fffb.Params */
} orc_fffb;

typedef struct {
    float thr, gain, nvar, vm_act_thr, sig_mult, sig_mult_pow, sig_gain, interp_range, gain_cor_range, gain_cor;
} orc_nxx1;

typedef struct {
    int32_t on, iters;
    float del_act_thr;
    orc_fffb lay, pool;
    orc_nxx1 xx1;
    float act_tau;
    float gbar[4]; /* E, L, I, K  (chans.Chans) */
    float erev[4];
} orc_kwta;

/* what Params.Update() derives */
typedef struct {
    float lay_fb_dt, pool_fb_dt;
    float sig_gain_nvar, sig_mult_eff, sig_val_at0, interp_val;
    float erev_sub_thr[4], thr_sub_erev[4];
    float act_dt;
} orc_kwta_derived;

/* goki/mat32 FastExp: Schraudolph's quartic-spline exp on the float32 bit pattern */
float orc_fast_exp(float x) {
    if (x <= -88.76731f) return 0.0f;
    int32_t i = (int32_t)(12102203.0f * x) + 127 * (1 << 23);
    int32_t m = (i >> 7) & 0xFFFF;
    i += (((((((((((3537 * m) >> 16) + 13668) * m) >> 18) + 15817) * m) >> 14) - 80470) * m) >> 11);
    float r;
    uint32_t u = (uint32_t)i;
    memcpy(&r, &u, 4);
    return r;
}

static float xx1(float x) { return x / (x + 1.0f); }

static float xx1_gain_cor(const orc_nxx1* p, float x) {
    float fact = (p->gain_cor_range - (x / p->nvar)) / p->gain_cor_range;
    if (fact < 0.0f) return xx1(p->gain * x);
    float new_gain = p->gain * (1.0f - p->gain_cor * fact);
    return xx1(new_gain * x);
}

void orc_kwta_update(const orc_kwta* k, orc_kwta_derived* d) {
    d->lay_fb_dt = 1.0f / k->lay.fb_tau;
    d->pool_fb_dt = 1.0f / k->pool.fb_tau;
    d->sig_gain_nvar = k->xx1.sig_gain / k->xx1.nvar;
    d->sig_mult_eff = k->xx1.sig_mult * (float)pow((double)(k->xx1.gain * k->xx1.nvar), (double)k->xx1.sig_mult_pow);
    d->sig_val_at0 = 0.5f * d->sig_mult_eff;
    d->interp_val = xx1_gain_cor(&k->xx1, k->xx1.interp_range) - d->sig_val_at0;
    for (int c = 0; c < 4; ++c) {
        d->erev_sub_thr[c] = k->erev[c] - k->xx1.thr;
        d->thr_sub_erev[c] = k->xx1.thr - k->erev[c];
    }
    d->act_dt = 1.0f / k->act_tau;
}

float orc_noisy_xx1(const orc_kwta* k, const orc_kwta_derived* d, float x) {
    if (x < 0.0f) {
        float ex = -(x * d->sig_gain_nvar);
        if (ex > 50.0f) return 0.0f;
        return d->sig_mult_eff / (1.0f + orc_fast_exp(ex));
    } else if (x < k->xx1.interp_range) {
        float interp = 1.0f - ((k->xx1.interp_range - x) / k->xx1.interp_range);
        return d->sig_val_at0 + interp * d->interp_val;
    }
    return xx1_gain_cor(&k->xx1, x);
}

void orc_kwta_defaults(orc_kwta* k) {
    memset(k, 0, sizeof(*k));
    k->on = 1;
    k->iters = 20;
    k->del_act_thr = 0.005f;
    orc_fffb f = {1, 1.8f, 1.0f, 1.0f, 1.4f, 0.0f, 0.1f};
    k->lay = f;
    k->pool = f;
    k->pool.gi = 2.0f;
    orc_nxx1 x = {0.5f, 100.0f, 0.005f, 0.01f, 0.33f, 0.8f, 3.0f, 0.01f, 10.0f, 0.1f};
    k->xx1 = x;
    k->xx1.gain = 80.0f;
    k->xx1.nvar = 0.01f;
    k->act_tau = 3.0f;
    const float gbar[4] = {0.5f, 0.1f, 1.0f, 1.0f}, erev[4] = {1.0f, 0.3f, 0.25f, 0.1f};
    memcpy(k->gbar, gbar, sizeof gbar);
    memcpy(k->erev, erev, sizeof erev);
}

/* minmax.AvgMax32 */
typedef struct {
    float avg, sum, max;
    int n;
} avgmax;
static void am_init(avgmax* a) {
    a->avg = 0;
    a->sum = 0;
    a->n = 0;
    a->max = -3.402823466e+38f;
}
static void am_update(avgmax* a, float v) {
    a->sum += v;
    a->n++;
    if (v > a->max) a->max = v;
}
static void am_calc(avgmax* a) {
    if (a->n > 0) {
        a->avg = a->sum / (float)a->n;
    } else {
        a->avg = a->sum;
        a->max = a->avg;
    }
}

/* fffb.Inhib: only the fields the computation reads back */
typedef struct {
    float fbi, gi;
    avgmax ge, act;
} inhib_t;

static void fffb_inhib(const orc_fffb* p, float fb_dt, inhib_t* inh) {
    if (!p->on) { /* Inhib.Zero() */
        inh->fbi = 0;
        inh->gi = 0;
        return;
    }
    float ff_netin = inh->ge.avg + p->max_vs_avg * (inh->ge.max - inh->ge.avg);
    float ffi = 0.0f;
    if (ff_netin > p->ff0) ffi = p->ff * (ff_netin - p->ff0);
    float fbi = p->fb * inh->act.avg;
    inh->fbi += fb_dt * (fbi - inh->fbi);
    inh->gi = p->gi * (ffi + inh->fbi);
}

static float ge_thr_from_g(const orc_kwta* k, const orc_kwta_derived* d, float gi) {
    return ((k->gbar[2] * gi * d->erev_sub_thr[2] + k->gbar[1] * d->erev_sub_thr[1]) / d->thr_sub_erev[0]);
}

/* KWTALayer: one inhibition level over all n values.  act: in = starting activations (the caller's
 * copy of raw), out = settled ones.  Returns the number of cycles run. */
int orc_kwta_layer(const orc_kwta* k, const float* raw, float* act, int n) {
    orc_kwta_derived d;
    orc_kwta_update(k, &d);
    inhib_t inh;
    memset(&inh, 0, sizeof inh);
    am_init(&inh.ge);
    for (int i = 0; i < n; ++i) am_update(&inh.ge, raw[i]);
    am_calc(&inh.ge);
    int cy = 0;
    for (; cy < k->iters; ++cy) {
        fffb_inhib(&k->lay, d.lay_fb_dt, &inh);
        am_init(&inh.act);
        float max_del = 0.0f;
        for (int i = 0; i < n; ++i) {
            float ge_thr = ge_thr_from_g(k, &d, inh.gi);
            float nw = orc_noisy_xx1(k, &d, raw[i] * k->gbar[0] - ge_thr);
            float del = d.act_dt * (nw - act[i]);
            nw = act[i] + del;
            max_del = fmaxf(max_del, fabsf(del));
            am_update(&inh.act, nw);
            act[i] = nw;
        }
        am_calc(&inh.act);
        if (cy > 2 && max_del < k->del_act_thr) {
            ++cy;
            break;
        }
    }
    return cy;
}

/* KWTAPool over a [d0, d1, d2, d3] tensor: layer level over everything, pool level inside each
 * (d0, d1) cell, effective inhibition = max of the two.  state: [d0*d1][2] = {FBi, Act.Avg} of each
 * pool's fffb.Inhib carried between calls (the SndEnv.Inhibs slice); NULL = a fresh slice (zeros). */
int orc_kwta_pool(const orc_kwta* k, const float* raw, float* act, int d0, int d1, int d2, int d3, float* state) {
    orc_kwta_derived d;
    orc_kwta_update(k, &d);
    const int lay_n = d0 * d1, pl_n = d2 * d3;
    inhib_t lay;
    memset(&lay, 0, sizeof lay);
    inhib_t pl[lay_n > 0 ? lay_n : 1];
    memset(pl, 0, sizeof pl);
    am_init(&lay.ge);
    for (int pi = 0; pi < lay_n; ++pi) {
        if (state) {
            pl[pi].fbi = state[2 * pi];
            pl[pi].act.avg = state[2 * pi + 1];
        }
        am_init(&pl[pi].ge);
        for (int ui = 0; ui < pl_n; ++ui) {
            float ge = raw[pi * pl_n + ui];
            am_update(&lay.ge, ge);
            am_update(&pl[pi].ge, ge);
        }
        am_calc(&pl[pi].ge);
    }
    am_calc(&lay.ge);
    int cy = 0;
    for (; cy < k->iters; ++cy) {
        fffb_inhib(&k->lay, d.lay_fb_dt, &lay);
        am_init(&lay.act);
        float max_del = 0.0f;
        for (int pi = 0; pi < lay_n; ++pi) {
            fffb_inhib(&k->pool, d.pool_fb_dt, &pl[pi]);
            float gi_pool = fmaxf(lay.gi, pl[pi].gi);
            am_init(&pl[pi].act);
            for (int ui = 0; ui < pl_n; ++ui) {
                int idx = pi * pl_n + ui;
                /* extGi is all zeros on this path: max(gi, Pool.Gi * FFInhib(0, 0)) = gi */
                float ge_thr = ge_thr_from_g(k, &d, gi_pool);
                float nw = orc_noisy_xx1(k, &d, raw[idx] * k->gbar[0] - ge_thr);
                float del = d.act_dt * (nw - act[idx]);
                nw = act[idx] + del;
                max_del = fmaxf(max_del, fabsf(del));
                am_update(&lay.act, nw);
                am_update(&pl[pi].act, nw);
                act[idx] = nw;
            }
            am_calc(&pl[pi].act);
        }
        am_calc(&lay.act);
        if (cy > 2 && max_del < k->del_act_thr) {
            ++cy;
            break;
        }
    }
    if (state)
        for (int pi = 0; pi < lay_n; ++pi) {
            state[2 * pi] = pl[pi].fbi;
            state[2 * pi + 1] = pl[pi].act.avg;
        }
    return cy;
}
