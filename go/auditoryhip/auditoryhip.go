// Package auditoryhip is the cgo binding of libauditory_hip.so (include/auditory_hip.h).
//
// NOT COMPILED IN THIS PIPELINE: the build image has no Go toolchain and the reference's module
// dependencies (etable, gonum, ...) cannot be fetched offline.  The file is what a maintainer of
// emer/auditory adds next to the existing packages; INTEGRATION.md shows how dft/mel/agabor/sound
// call into it.  It follows the cgo rules the C ABI was designed for: C never keeps a Go pointer
// after a call returns, all sizes are explicit, every call returns a status code.
package auditoryhip

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../auditory_amd -lauditory_hip -Wl,-rpath,${SRCDIR}/../../auditory_amd
#include <stdlib.h>
#include "auditory_hip.h"
*/
import "C"

import (
	"github.com/emer/leabra/fffb"
	"github.com/emer/vision/kwta"
	"errors"
	"fmt"
	"unsafe"
)

// Ctx owns one GPU (one process per GPU is the intended deployment).
type Ctx struct{ h *C.aud_ctx }

// Plan holds the device-resident tables of one parameter set.
type Plan struct {
	h        *C.aud_plan
	ctx      *Ctx
	NFilters int
	Steps    int // SegmentSteps (T)
	Bins     int // WinSamples/2 + 1 (H)
	NGabor   int
}

// Item is one segment of one mono stream: see aud_item in auditory_hip.h.
type Item struct {
	SigOff int64
	SigLen int32
	Start0 int32
}

func status(ctx *Ctx, rc C.int) error {
	if rc == C.AUD_OK {
		return nil
	}
	msg := C.GoString(C.aud_status_string(rc))
	if ctx != nil && ctx.h != nil {
		if m := C.GoString(C.aud_last_error(ctx.h)); m != "" {
			msg = m
		}
	}
	return fmt.Errorf("auditory_hip: status %d: %s", int(rc), msg)
}

// Init opens the device; there is no CPU fallback, an error here is final.
func Init(device int) (*Ctx, error) {
	c := &Ctx{}
	if rc := C.aud_init(C.int(device), &c.h); rc != C.AUD_OK {
		return nil, status(nil, rc)
	}
	return c, nil
}

func (c *Ctx) Close() { C.aud_shutdown(c.h); c.h = nil }

// MSecToSamples is sound.MSecToSamples (sound/sndenv.go:522-524).
func MSecToSamples(ms float64, rate int) int { return int(C.aud_msec_to_samples(C.double(ms), C.int(rate))) }

// MelInitFilters is the arithmetic of mel.Params.InitFilters (mel/mel.go:77-117): binPts has
// nf+2 entries, filters nf*(nf+2); renorm is cleared like mel.go:80 does.
func MelInitFilters(fb *C.aud_mel_fbank, dftSize, sampleRate int, binPts []int32, hzPts, filters []float64) error {
	rc := C.aud_mel_init_filters(fb, C.int(dftSize), C.int(sampleRate),
		(*C.int32_t)(unsafe.Pointer(&binPts[0])), (*C.double)(unsafe.Pointer(&hzPts[0])),
		(*C.double)(unsafe.Pointer(&filters[0])))
	return status(nil, rc)
}

// GaborToTensor is agabor.ToTensor (agabor/gabor.go:89-222); out has nActive*SizeY*SizeX entries.
func GaborToTensor(specs []C.aud_gabor_spec, set *C.aud_gabor_set, out []float64) (int, error) {
	var n C.int
	rc := C.aud_gabor_to_tensor(&specs[0], C.int(len(specs)), set, (*C.double)(unsafe.Pointer(&out[0])), &n)
	return int(n), status(nil, rc)
}

// NewPlan uploads the tables.  desc's pointer fields must point at Go slices that stay alive for
// the duration of this call only (they are copied to the device before it returns).
func (c *Ctx) NewPlan(desc *C.aud_plan_desc) (*Plan, error) {
	p := &Plan{ctx: c, NFilters: int(desc.mel.n_filters), Steps: int(desc.segment_steps),
		Bins: int(desc.win_samples)/2 + 1, NGabor: int(desc.n_gabor)}
	if rc := C.aud_plan_create(c.h, desc, &p.h); rc != C.AUD_OK {
		return nil, status(c, rc)
	}
	return p, nil
}

func (p *Plan) Close() { C.aud_plan_destroy(p.h); p.h = nil }

// MelSpec runs the ProcessSegment frame loop (sound/sndenv.go:342-359) for all items in one
// launch.  sig is SndEnv.Signal.Values; mel receives [len(items)][NFilters][Steps] float64;
// power / logPower may be nil.
func (p *Plan) MelSpec(sig []float64, items []Item, mel, power, logPower []float64) error {
	if len(items) == 0 {
		return nil
	}
	if len(mel) < len(items)*p.NFilters*p.Steps {
		return errors.New("auditory_hip: mel buffer too small")
	}
	var pp, lp *C.double
	if power != nil {
		pp = (*C.double)(unsafe.Pointer(&power[0]))
	}
	if logPower != nil {
		lp = (*C.double)(unsafe.Pointer(&logPower[0]))
	}
	rc := C.aud_melspec_batch_host(p.h, (*C.double)(unsafe.Pointer(&sig[0])), C.int64_t(len(sig)),
		(*C.aud_item)(unsafe.Pointer(&items[0])), C.int(len(items)),
		(*C.double)(unsafe.Pointer(&mel[0])), pp, lp)
	return status(p.ctx, rc)
}

// Convolve is agabor.Convolve (agabor/gabor.go:225-315) for nItems mel matrices at once; out is
// in/out (cells the reference does not write keep their values).
func (p *Plan) Convolve(mel []float64, nItems, rows, cols int, outShape []int32, byTime bool, out []float32) error {
	bt := C.int(0)
	if byTime {
		bt = 1
	}
	rc := C.aud_gabor_batch_host(p.h, (*C.double)(unsafe.Pointer(&mel[0])), C.int(nItems), C.int(rows), C.int(cols),
		C.int(len(outShape)), (*C.int32_t)(unsafe.Pointer(&outShape[0])), bt, (*C.float)(unsafe.Pointer(&out[0])))
	return status(p.ctx, rc)
}

// SamplesToMSec is sound.SamplesToMSec (sound/sndenv.go:527-529).
func SamplesToMSec(samples, rate int) float64 {
	return float64(C.aud_samples_to_msec(C.int(samples), C.int(rate)))
}

// Power is dft.Params.Power (dft/dft.go:62-85) for one step on coefficients the caller computed; power is the
// carry of the previous step on entry.  logPower / logPowerSeg may be nil.
func (p *Plan) Power(step int, fftCoefs []complex128, power, logPower, powerSeg, logPowerSeg []float64) error {
	ptr := func(s []float64) *C.double {
		if len(s) == 0 {
			return nil
		}
		return (*C.double)(unsafe.Pointer(&s[0]))
	}
	rc := C.aud_dft_power_host(p.h, C.int(step), (*C.double)(unsafe.Pointer(&fftCoefs[0])), ptr(power), ptr(logPower),
		ptr(powerSeg), ptr(logPowerSeg))
	return status(p.ctx, rc)
}

// CepstrumDct is mel.Params.CepstrumDct (mel/mel.go:192-212) for one step; the plan must have been created with
// mfcc_coefs = NCoefs.  mfccSeg is [NCoefs][Steps] row-major; mfccDct (may be nil) ends as a copy of fbank.
func (p *Plan) CepstrumDct(step int, fbank, mfccSeg, mfccDct []float64) error {
	var work *C.double
	if len(mfccDct) > 0 {
		work = (*C.double)(unsafe.Pointer(&mfccDct[0]))
	}
	rc := C.aud_cepstrum_dct_host(p.h, C.int(step), (*C.double)(unsafe.Pointer(&fbank[0])),
		(*C.double)(unsafe.Pointer(&mfccSeg[0])), work)
	return status(p.ctx, rc)
}

// KwtaParams copies the exported fields of a kwta.KWTA (github.com/emer/vision/kwta) into the C struct,
// one for one.  The derived fields (ErevSubThr, ThrSubErev, ActDt, the nxx1 Sig* values, FBDt) are
// recomputed inside the library, like KWTA.Update() does.
func KwtaParams(k *kwta.KWTA) C.aud_kwta_params {
	fffb := func(p *fffb.Params) C.aud_fffb_params {
		on := C.int32_t(0)
		if p.On {
			on = 1
		}
		return C.aud_fffb_params{on: on, gi: C.float(p.Gi), ff: C.float(p.FF), fb: C.float(p.FB),
			fb_tau: C.float(p.FBTau), max_vs_avg: C.float(p.MaxVsAvg), ff0: C.float(p.FF0)}
	}
	var c C.aud_kwta_params
	if k.On {
		c.on = 1
	}
	c.iters = C.int32_t(k.Iters)
	c.del_act_thr = C.float(k.DelActThr)
	c.lay_fffb, c.pool_fffb = fffb(&k.LayFFFB), fffb(&k.PoolFFFB)
	x := &k.XX1
	c.xx1 = C.aud_nxx1_params{thr: C.float(x.Thr), gain: C.float(x.Gain), nvar: C.float(x.NVar),
		vm_act_thr: C.float(x.VmActThr), sig_mult: C.float(x.SigMult), sig_mult_pow: C.float(x.SigMultPow),
		sig_gain: C.float(x.SigGain), interp_range: C.float(x.InterpRange),
		gain_cor_range: C.float(x.GainCorRange), gain_cor: C.float(x.GainCor)}
	c.act_tau = C.float(k.ActTau)
	c.gbar = [4]C.float{C.float(k.Gbar.E), C.float(k.Gbar.L), C.float(k.Gbar.I), C.float(k.Gbar.K)}
	c.erev = [4]C.float{C.float(k.Erev.E), C.float(k.Erev.L), C.float(k.Erev.I), C.float(k.Erev.K)}
	return c
}

// Kwta is SndEnv.ApplyKwta's KWTAPool / KWTALayer call (sound/sndenv.go:313-323) on one or more tensors:
// raw and act are [nItems][shape...] float32, act in/out (the caller has copied raw into it, :315);
// state is the flattened {FBi, Act.Avg} of the pool-level fffb.Inhibs (nil = fresh), see PoolState.
func (c *Ctx) Kwta(k *kwta.KWTA, raw, act []float32, nItems int, shape [4]int, pool bool, state []float32) error {
	kp := KwtaParams(k)
	pl := C.int(0)
	if pool {
		pl = 1
	}
	var st *C.float
	if len(state) > 0 {
		st = (*C.float)(unsafe.Pointer(&state[0]))
	}
	rc := C.aud_kwta_batch_host(c.h, &kp, (*C.float)(unsafe.Pointer(&raw[0])), (*C.float)(unsafe.Pointer(&act[0])),
		C.int(nItems), C.int(shape[0]), C.int(shape[1]), C.int(shape[2]), C.int(shape[3]), pl, 0, st, 0, nil)
	return status(c, rc)
}

// PoolState / SetPoolState move the two fields of fffb.Inhib that KWTAPool carries between calls
// (FBi and Act.Avg) between the Go slice (SndEnv.Inhibs) and the flat float32 layout of the C ABI.
func PoolState(inh fffb.Inhibs, state []float32) {
	for i := range inh {
		state[2*i], state[2*i+1] = inh[i].FBi, inh[i].Act.Avg
	}
}

func SetPoolState(inh fffb.Inhibs, state []float32) {
	for i := range inh {
		inh[i].FBi, inh[i].Act.Avg = state[2*i], state[2*i+1]
	}
}
