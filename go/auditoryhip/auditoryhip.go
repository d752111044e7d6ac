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
	"sync"
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
	NCoefs   int // mfcc_coefs of the plan (0: no MFCC tail)
}

// Item is one segment of one mono stream: see aud_item in auditory_hip.h (same layout, 24 bytes).
// SigStride 0 or 1 = contiguous mono; 2 with SigOff 0 / 1 = left / right channel of interleaved stereo PCM.
type Item struct {
	SigOff    int64
	SigLen    int32
	Start0    int32
	SigStride int32
	Reserved  int32
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

// NewPlan uploads the tables.  desc holds scalars only (no pointers), so passing its address is within the cgo
// pointer rules; binPts, melFilters and gaborFilters are Go slices passed as direct call arguments, which cgo
// pins for the duration of the call, and the library copies them to the device before it returns.
// binPts: mel.Params.BinPts [nf+2]; melFilters: SndEnv.MelFilters.Values [nf*(nf+2)];
// gaborFilters: FilterSet.Filters.Values [nG*SizeY*SizeX] or nil.
func (c *Ctx) NewPlan(desc *C.aud_plan_desc, binPts []int32, melFilters, gaborFilters []float64) (*Plan, error) {
	p := &Plan{ctx: c, NFilters: int(desc.mel.n_filters), Steps: int(desc.segment_steps),
		Bins: int(desc.win_samples)/2 + 1, NGabor: int(desc.n_gabor), NCoefs: int(desc.mfcc_coefs)}
	var gk *C.double
	if len(gaborFilters) > 0 {
		gk = (*C.double)(unsafe.Pointer(&gaborFilters[0]))
	}
	rc := C.aud_plan_create(c.h, desc, (*C.int32_t)(unsafe.Pointer(&binPts[0])),
		(*C.double)(unsafe.Pointer(&melFilters[0])), gk, &p.h)
	if rc != C.AUD_OK {
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
	if nItems <= 0 {
		return nil
	}
	if len(mel) < nItems*rows*cols || len(outShape) == 0 || len(out) == 0 {
		return errors.New("auditory_hip: Convolve: empty or short tensor") // (the reference logs and returns: gabor.go:226-229)
	}
	rc := C.aud_gabor_batch_host(p.h, (*C.double)(unsafe.Pointer(&mel[0])), C.int(nItems), C.int(rows), C.int(cols),
		C.int(len(outShape)), (*C.int32_t)(unsafe.Pointer(&outShape[0])), bt, (*C.float)(unsafe.Pointer(&out[0])))
	return status(p.ctx, rc)
}

// SamplesToMSec is sound.SamplesToMSec (sound/sndenv.go:527-529).
func SamplesToMSec(samples, rate int) float64 {
	return float64(C.aud_samples_to_msec(C.int(samples), C.int(rate)))
}

// KwtaParams copies the exported fields of a kwta.KWTA (github.com/emer/vision/kwta) into the C struct,
// one for one.  The derived fields (ErevSubThr, ThrSubErev, ActDt, the nxx1 Sig* values, FBDt) are
// recomputed inside the library, like KWTA.Update() does.
func KwtaParams(k *kwta.KWTA) C.aud_kwta_params {
	toC := func(p *fffb.Params) C.aud_fffb_params {
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
	c.lay_fffb, c.pool_fffb = toC(&k.LayFFFB), toC(&k.PoolFFFB)
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

// ---- per-step entry points (one frame per call: correct, never fast; see auditory_hip.h) -------------------

// DftFilter is dft.Params.Filter for one step (dft/dft.go:42-85): window [N]; power [H] carries the previous
// step's power in and this step's out; powerSeg / logPowerSeg are the [H, T] segment tensors (column `step`).
func (p *Plan) DftFilter(step int, window, power, logPower, powerSeg, logPowerSeg []float64) error {
	var lp, lps *C.double
	if logPower != nil {
		lp = (*C.double)(unsafe.Pointer(&logPower[0]))
	}
	if logPowerSeg != nil {
		lps = (*C.double)(unsafe.Pointer(&logPowerSeg[0]))
	}
	rc := C.aud_dft_filter_host(p.h, C.int(step), (*C.double)(unsafe.Pointer(&window[0])),
		(*C.double)(unsafe.Pointer(&power[0])), lp, (*C.double)(unsafe.Pointer(&powerSeg[0])), lps)
	return status(p.ctx, rc)
}

// DftPower is dft.Params.Power (dft/dft.go:62-85) on coefficients the caller computed.
func (p *Plan) DftPower(step int, fftCoefs []complex128, power, logPower, powerSeg, logPowerSeg []float64) error {
	var lp, lps *C.double
	if logPower != nil {
		lp = (*C.double)(unsafe.Pointer(&logPower[0]))
	}
	if logPowerSeg != nil {
		lps = (*C.double)(unsafe.Pointer(&logPowerSeg[0]))
	}
	rc := C.aud_dft_power_host(p.h, C.int(step), (*C.double)(unsafe.Pointer(&fftCoefs[0])),
		(*C.double)(unsafe.Pointer(&power[0])), lp, (*C.double)(unsafe.Pointer(&powerSeg[0])), lps)
	return status(p.ctx, rc)
}

// MelFilterDft is mel.Params.FilterDft for one step (mel/mel.go:120-153).
func (p *Plan) MelFilterDft(step int, power, segment, fbank []float64) error {
	rc := C.aud_mel_filter_dft_host(p.h, C.int(step), (*C.double)(unsafe.Pointer(&power[0])),
		(*C.double)(unsafe.Pointer(&segment[0])), (*C.double)(unsafe.Pointer(&fbank[0])))
	return status(p.ctx, rc)
}

// CepstrumDct is mel.Params.CepstrumDct for one step (mel/mel.go:192-212); the plan needs mfcc_coefs = NCoefs.
func (p *Plan) CepstrumDct(step int, fbank, mfccSeg, mfccDct []float64) error {
	var md *C.double
	if mfccDct != nil {
		md = (*C.double)(unsafe.Pointer(&mfccDct[0]))
	}
	rc := C.aud_cepstrum_dct_host(p.h, C.int(step), (*C.double)(unsafe.Pointer(&fbank[0])),
		(*C.double)(unsafe.Pointer(&mfccSeg[0])), md)
	return status(p.ctx, rc)
}

// SndToWindow is SndEnv.SndToWindow (sound/sndenv.go:455-478); ErrShort mirrors its "end beyond signal length".
var ErrShort = errors.New("SndToWindow: end beyond signal length!!")

func SndToWindow(signal []float64, start, winSamples int, window []float64) error {
	rc := C.aud_snd_to_window((*C.double)(unsafe.Pointer(&signal[0])), C.int64_t(len(signal)), C.int64_t(start),
		C.int(winSamples), (*C.double)(unsafe.Pointer(&window[0])))
	if rc == C.AUD_ESHORT {
		return ErrShort
	}
	return status(nil, rc)
}

// Default is the process-wide context the drop-in packages (go/dft, go/mel, go/agabor, go/sound) share.
var (
	defaultCtx  *Ctx
	defaultErr  error
	defaultOnce sync.Once
)

func Default() (*Ctx, error) {
	defaultOnce.Do(func() { defaultCtx, defaultErr = Init(0) }) // goroutines race to the first call
	return defaultCtx, defaultErr
}

// ---- plain-Go mirrors of the C parameter blocks, for the drop-in packages (no C types in their signatures) -------

type DftParams struct {
	CompLogPow                               bool
	LogMin, LogOffSet, PrevSmooth, CurSmooth float64
}
type MelFBank struct {
	NFilters                                              int
	LoHz, HiHz, LogOff, LogMin                            float64
	Renorm                                                bool
	RenormMin, RenormMax, RenormScale                     float64
}
type SoundParams struct {
	WinMs, StepMs, SegmentMs, StrideMs                                        float64
	BorderSteps, Channel                                                      int
	WinSamples, StepSamples, SegmentSamples, StrideSamples, SegmentSteps     int
}
type GaborSpec struct {
	Off                                                           bool
	WaveLen, Orientation, SigmaWidth, SigmaLength, PhaseOffset    float64
	CircleEdge, Circular                                          bool
}

func b2i(b bool) C.int32_t {
	if b {
		return 1
	}
	return 0
}

// DftDefaults is dft.Params.Defaults (dft/dft.go:33-39).
func DftDefaults() DftParams {
	var d C.aud_dft_params
	C.aud_dft_defaults(&d)
	return DftParams{d.comp_log_pow != 0, float64(d.log_min), float64(d.log_offset), float64(d.prev_smooth), float64(d.cur_smooth)}
}

// MelDefaults is mel.FilterBank.Defaults (mel/mel.go:171-180).
func MelDefaults() MelFBank {
	var m C.aud_mel_fbank
	C.aud_mel_defaults(&m)
	return MelFBank{int(m.n_filters), float64(m.lo_hz), float64(m.hi_hz), float64(m.log_off), float64(m.log_min),
		m.renorm != 0, float64(m.renorm_min), float64(m.renorm_max), float64(m.renorm_scale)}
}

func (m MelFBank) c() C.aud_mel_fbank {
	return C.aud_mel_fbank{n_filters: C.int32_t(m.NFilters), lo_hz: C.double(m.LoHz), hi_hz: C.double(m.HiHz),
		log_off: C.double(m.LogOff), log_min: C.double(m.LogMin), renorm: b2i(m.Renorm),
		renorm_min: C.double(m.RenormMin), renorm_max: C.double(m.RenormMax), renorm_scale: C.double(m.RenormScale)}
}

// MelInitFiltersGo is mel.Params.InitFilters' arithmetic (mel/mel.go:77-117) on Go slices.
func MelInitFiltersGo(fb *MelFBank, dftSize, sampleRate int, binPts []int32, hzPts, filters []float64) error {
	cfb := fb.c()
	err := MelInitFilters(&cfb, dftSize, sampleRate, binPts, hzPts, filters)
	fb.Renorm = cfb.renorm != 0
	return err
}

func FreqToMel(f float64) float64 { return float64(C.aud_freq_to_mel(C.double(f))) }
func MelToFreq(m float64) float64 { return float64(C.aud_mel_to_freq(C.double(m))) }
func FreqToBin(f, nFft, sr float64) int {
	return int(C.aud_freq_to_bin(C.double(f), C.double(nFft), C.double(sr)))
}

// SoundParamDefaults / SoundParamsDerive: SndEnv.ParamDefaults and the derivations of SndEnv.Init (sndenv.go:64-71, :202-207).
func SoundParamDefaults() SoundParams {
	var p C.aud_sound_params
	C.aud_sound_params_defaults(&p)
	return SoundParams{WinMs: float64(p.win_ms), StepMs: float64(p.step_ms), SegmentMs: float64(p.segment_ms),
		StrideMs: float64(p.stride_ms), BorderSteps: int(p.border_steps), Channel: int(p.channel)}
}
func SoundParamsDerive(winMs, stepMs, segMs, strideMs float64, border, rate int) (SoundParams, error) {
	p := C.aud_sound_params{win_ms: C.double(winMs), step_ms: C.double(stepMs), segment_ms: C.double(segMs),
		stride_ms: C.double(strideMs), border_steps: C.int32_t(border)}
	if rc := C.aud_sound_params_derive(&p, C.int(rate)); rc != C.AUD_OK {
		return SoundParams{}, status(nil, rc)
	}
	return SoundParams{winMs, stepMs, segMs, strideMs, border, 0, int(p.win_samples), int(p.step_samples),
		int(p.segment_samples), int(p.stride_samples), int(p.segment_steps)}, nil
}
func SegCnt(sigLen, segSamples, strideSamples, channels int) int {
	return int(C.aud_seg_cnt(C.int(sigLen), C.int(segSamples), C.int(strideSamples), C.int(channels)))
}
func Tail(sigLen, segSamples, strideSamples int) int {
	return int(C.aud_tail(C.int(sigLen), C.int(segSamples), C.int(strideSamples)))
}
func PadLen(sigLen, segSamples, strideSamples, stepSamples int) int {
	return int(C.aud_pad_len(C.int(sigLen), C.int(segSamples), C.int(strideSamples), C.int(stepSamples)))
}
func AdjustForSilence(add, existing float64, rate int) (offset, delta int) {
	var d C.int
	off := C.aud_adjust_for_silence(C.double(add), C.double(existing), C.int(rate), &d)
	return int(off), int(d)
}

func GaborSpecOf(off bool, waveLen, orient, sw, sl, phase float64, circleEdge, circular bool) GaborSpec {
	return GaborSpec{off, waveLen, orient, sw, sl, phase, circleEdge, circular}
}
func (g GaborSpec) c() C.aud_gabor_spec {
	return C.aud_gabor_spec{off: b2i(g.Off), wave_len: C.double(g.WaveLen), orientation: C.double(g.Orientation),
		sigma_width: C.double(g.SigmaWidth), sigma_length: C.double(g.SigmaLength), phase_offset: C.double(g.PhaseOffset),
		circle_edge: b2i(g.CircleEdge), circular: b2i(g.Circular)}
}

// GaborSpecDefaults is agabor.Filter.Defaults (agabor/gabor.go:73-86): zero fields get their defaults, with the notice.
func GaborSpecDefaults(g *GaborSpec, i int) {
	if g.WaveLen == 0 {
		fmt.Printf("filter %v: WaveLen is 0 -- using default of 2\n", i)
		g.WaveLen = 2
	}
	if g.SigmaLength == 0 && !g.Circular {
		fmt.Printf("filter %v: SigmaLength is 0 -- using default of 0.5\n", i)
		g.SigmaLength = 0.5
	}
	if g.SigmaWidth == 0 {
		fmt.Printf("filter %v: SigmaWidth is 0 -- using default of 0.5\n", i)
		g.SigmaWidth = 0.5
	}
}

func gaborSet(sx, sy, stx, sty int, gain float64, distribute bool) C.aud_gabor_set {
	return C.aud_gabor_set{size_x: C.int32_t(sx), size_y: C.int32_t(sy), stride_x: C.int32_t(stx), stride_y: C.int32_t(sty),
		gain: C.double(gain), distribute: b2i(distribute)}
}

// GaborToTensorGo is agabor.ToTensor (agabor/gabor.go:89-222) on Go values.
func GaborToTensorGo(specs []GaborSpec, sx, sy, stx, sty int, gain float64, distribute bool, out []float64) (int, error) {
	cs := make([]C.aud_gabor_spec, len(specs))
	for i, g := range specs {
		cs[i] = g.c()
	}
	set := gaborSet(sx, sy, stx, sty, gain, distribute)
	return GaborToTensor(cs, &set, out)
}

func planDesc(n, s, t, border int, d DftParams, m MelFBank, f64 bool, mfcc int) C.aud_plan_desc {
	desc := C.aud_plan_desc{win_samples: C.int32_t(n), step_samples: C.int32_t(s), segment_steps: C.int32_t(t),
		border_steps: C.int32_t(border), mfcc_coefs: C.int32_t(mfcc)}
	desc.dft = C.aud_dft_params{comp_log_pow: b2i(d.CompLogPow), log_min: C.double(d.LogMin), log_offset: C.double(d.LogOffSet),
		prev_smooth: C.double(d.PrevSmooth), cur_smooth: C.double(d.CurSmooth)}
	desc.mel = m.c()
	desc.compute_dtype = C.AUD_F64 // == 0: the zero value is the plan that computes as the reference does
	if !f64 {
		desc.compute_dtype = C.AUD_FAST_F32 // explicit opt-in (SndEnv.ComputeF32)
	}
	return desc
}

// NewSndEnvPlan is what SndEnv.Init builds: the segment geometry, dft / mel parameters and tables, and the gabor set.
func (c *Ctx) NewSndEnvPlan(sp SoundParams, compLogPow bool, logMin, logOff, prevSmooth, curSmooth float64,
	nf int, loHz, hiHz, melLogOff, melLogMin float64, renorm bool, renormMin, renormMax, renormScale float64,
	binPts []int32, melFilters []float64, gsx, gsy, gstx, gsty int, gain float64, gaborFilters []float64,
	mfccCoefs int, f64 bool) (*Plan, error) {
	desc := planDesc(sp.WinSamples, sp.StepSamples, sp.SegmentSteps, sp.BorderSteps,
		DftParams{compLogPow, logMin, logOff, prevSmooth, curSmooth},
		MelFBank{nf, loHz, hiHz, melLogOff, melLogMin, renorm, renormMin, renormMax, renormScale}, f64, mfccCoefs)
	if n := gsx * gsy; n > 0 && len(gaborFilters) >= n {
		desc.n_gabor = C.int32_t(len(gaborFilters) / n)
		desc.gabor = gaborSet(gsx, gsy, gstx, gsty, gain, false)
	}
	return c.NewPlan(&desc, binPts, melFilters, gaborFilters)
}

// NewStepPlan serves dft.Params.Filter / Power called on their own: a one-filter dummy mel table keeps the plan valid.
func NewStepPlan(winSamples, steps int, compLogPow bool, logMin, logOff, prevSmooth, curSmooth float64) (*Plan, error) {
	c, err := Default()
	if err != nil {
		return nil, err
	}
	desc := planDesc(winSamples, 1, steps, 0, DftParams{compLogPow, logMin, logOff, prevSmooth, curSmooth},
		MelFBank{NFilters: 1, LogMin: -10}, true, 0)
	return c.NewPlan(&desc, []int32{0, 0, 0}, []float64{1, 0, 0}, nil)
}

// NewMelStepPlan serves mel.Params.FilterDft / CepstrumDct called on their own.
func NewMelStepPlan(winSamples, steps int, fb MelFBank, binPts []int32, filters []float64, nCoefs int) (*Plan, error) {
	c, err := Default()
	if err != nil {
		return nil, err
	}
	desc := planDesc(winSamples, 1, steps, 0, DftDefaults(), fb, true, nCoefs)
	return c.NewPlan(&desc, binPts, filters, nil)
}

// GaborPlan serves agabor.Convolve called on its own (plans cached per tap set would go here).
func GaborPlan(sx, sy, stx, sty int, gain float64, taps []float64) (*Plan, error) {
	c, err := Default()
	if err != nil {
		return nil, err
	}
	desc := planDesc(4, 1, 1, 0, DftDefaults(), MelFBank{NFilters: 1, LogMin: -10}, true, 0)
	desc.n_gabor = C.int32_t(len(taps) / (sx * sy))
	desc.gabor = gaborSet(sx, sy, stx, sty, gain, false)
	return c.NewPlan(&desc, []int32{0, 0, 0}, []float64{1, 0, 0}, taps)
}

// MelSpecMFCC is ProcessSegment with Mel.MFCC on (sndenv.go:342-435) for all items in one call.
func (p *Plan) MelSpecMFCC(sig []float64, items []Item, mel, power, logPower, mfcc, deltas, deltaDeltas, energy []float64) error {
	if len(items) == 0 {
		return nil
	}
	if len(mel) < len(items)*p.NFilters*p.Steps || len(mfcc) < len(items)*p.NCoefs*p.Steps {
		return errors.New("auditory_hip: mel / mfcc buffer too small")
	}
	ptr := func(s []float64) *C.double {
		if len(s) == 0 {
			return nil
		}
		return (*C.double)(unsafe.Pointer(&s[0]))
	}
	rc := C.aud_melspec_mfcc_batch_host(p.h, ptr(sig), C.int64_t(len(sig)), (*C.aud_item)(unsafe.Pointer(&items[0])),
		C.int(len(items)), ptr(mel), ptr(power), ptr(logPower), ptr(mfcc), ptr(deltas), ptr(deltaDeltas), ptr(energy))
	return status(p.ctx, rc)
}

// Signal is a signal kept resident on the device between calls: SndEnv.ProcessSegment runs once per segment on the SAME Signal
// tensor (sound/sndenv.go:342-359), so the host-buffer calls above move the whole tensor over the link again for every
// segment.  Two forms: UploadSignal is a SNAPSHOT (the caller keeps it current); SyncSignal is EXACT -- the library keeps a
// host shadow of what the device holds and compares the caller's slice with it byte for byte on every call.
type Signal struct {
	h   *C.aud_signal
	ctx *Ctx
}

// ResidentAutoBytes is the size up to which SndEnv validates its resident Signal exactly on every call (SyncSignal); a
// larger Signal is copied per call unless the caller opts in to a snapshot.
const ResidentAutoBytes = C.AUD_RESIDENT_AUTO_BYTES

// SyncSignal makes the device copy s (nil: created) EQUAL to sig: aud_signal_sync compares sig with the shadow in 4 KB blocks
// and uploads the span from the first to the last differing block -- nothing when they are equal, everything the first time or
// when the length changed.  Returns the signal and the bytes that crossed the link.  The reference reads the live tensor at
// every step (sound/sndenv.go:455-478); this is what lets a resident copy do the same.
func (c *Ctx) SyncSignal(s *Signal, sig []float64) (*Signal, int64, error) {
	if s == nil {
		s = &Signal{ctx: c}
	}
	var p unsafe.Pointer
	if len(sig) > 0 {
		p = unsafe.Pointer(&sig[0])
	}
	var up C.int64_t
	if err := status(c, C.aud_signal_sync(c.h, &s.h, p, C.AUD_F64, C.int64_t(len(sig)), &up)); err != nil {
		return s, 0, err
	}
	return s, int64(up), nil
}

// HostFloat64 returns n float64 in pinned, device-visible host memory (aud_host_alloc) as a Go slice over C memory.  Result
// tensors whose Values live there are written by the DEVICE (it widens its float32 results and stores them over the link):
// no staging copy, no widening pass on the CPU -- 0.20 ms instead of 0.34 ms per 256 utterances (profiles/round5_host_call_time.txt).
// The library takes that route when ALL the output tensors of a call lie in such memory.  The slice is not garbage-collected
// memory: HostFree it (or let Ctx.Close do it) after the last call that writes it, and do not append to it.
func (c *Ctx) HostFloat64(n int) ([]float64, error) {
	if n <= 0 {
		return nil, nil
	}
	var p unsafe.Pointer
	if err := status(c, C.aud_host_alloc(c.h, C.int64_t(8*n), &p)); err != nil {
		return nil, err
	}
	return unsafe.Slice((*float64)(p), n), nil
}

// HostFree releases a slice of HostFloat64.
func (c *Ctx) HostFree(s []float64) {
	if len(s) > 0 && c != nil && c.h != nil {
		C.aud_host_free(c.h, unsafe.Pointer(&s[0]))
	}
}

// HostRegister pins memory the caller owns -- a mapping several processes share (syscall.Mmap over /dev/shm), say -- and
// makes it device-visible: result tensors whose Values are sub-slices of it are written by the device like HostFloat64 ones.
// One process per GPU, each passing its shard's slice of ONE [B, nf, T] tensor: the batch's features end in one host
// tensor with no collective.  HostUnregister before unmapping.
func (c *Ctx) HostRegister(b []float64) error {
	if len(b) == 0 {
		return errors.New("auditory_hip: HostRegister: empty slice")
	}
	return status(c, C.aud_host_register(c.h, unsafe.Pointer(&b[0]), C.int64_t(8*len(b))))
}

func (c *Ctx) HostUnregister(b []float64) error {
	if len(b) == 0 {
		return nil
	}
	return status(c, C.aud_host_unregister(c.h, unsafe.Pointer(&b[0])))
}

// UploadSignal copies SndEnv.Signal.Values to the device once.
func (c *Ctx) UploadSignal(sig []float64) (*Signal, error) {
	s := &Signal{ctx: c}
	var p unsafe.Pointer
	if len(sig) > 0 {
		p = unsafe.Pointer(&sig[0])
	}
	if err := status(c, C.aud_signal_upload(c.h, p, C.AUD_F64, C.int64_t(len(sig)), &s.h)); err != nil {
		return nil, err
	}
	return s, nil
}

// UploadPCM16 copies the WAV's own 16-bit samples (2 bytes per sample over the link); the device normalises them by
// 0x7FFF exactly as Wave.SoundToTensor does in float64 (sound/sound.go:138).
func (c *Ctx) UploadPCM16(pcm []int16) (*Signal, error) {
	s := &Signal{ctx: c}
	var p unsafe.Pointer
	if len(pcm) > 0 {
		p = unsafe.Pointer(&pcm[0])
	}
	if err := status(c, C.aud_signal_upload(c.h, p, C.AUD_I16, C.int64_t(len(pcm)), &s.h)); err != nil {
		return nil, err
	}
	return s, nil
}

func (s *Signal) Close() {
	if s != nil && s.h != nil {
		C.aud_signal_destroy(s.h)
		s.h = nil
	}
}

func (s *Signal) Len() int { return int(C.aud_signal_len(s.h)) }

// MelSpecSig is MelSpec on a resident signal: only the items go up, only the results come back.
func (p *Plan) MelSpecSig(sig *Signal, items []Item, mel, power, logPower []float64) error {
	if len(items) == 0 {
		return nil
	}
	if len(mel) < len(items)*p.NFilters*p.Steps {
		return errors.New("auditory_hip: mel buffer too small")
	}
	ptr := func(s []float64) *C.double {
		if len(s) == 0 {
			return nil
		}
		return (*C.double)(unsafe.Pointer(&s[0]))
	}
	rc := C.aud_melspec_batch_sig(p.h, sig.h, (*C.aud_item)(unsafe.Pointer(&items[0])), C.int(len(items)), ptr(mel), ptr(power), ptr(logPower))
	return status(p.ctx, rc)
}

// MelSpecMFCCSig is MelSpecMFCC on a resident signal.
func (p *Plan) MelSpecMFCCSig(sig *Signal, items []Item, mel, power, logPower, mfcc, deltas, deltaDeltas, energy []float64) error {
	if len(items) == 0 {
		return nil
	}
	if sig == nil || sig.h == nil {
		return errors.New("auditory_hip: nil signal")
	}
	n, h := len(items), p.Bins
	if len(mel) < n*p.NFilters*p.Steps || len(mfcc) < n*p.NCoefs*p.Steps {
		return errors.New("auditory_hip: mel / mfcc buffer too small")
	}
	for _, b := range [][]float64{power, logPower} {
		if len(b) != 0 && len(b) < n*h*p.Steps {
			return errors.New("auditory_hip: spectrum buffer too small")
		}
	}
	for _, b := range [][]float64{deltas, deltaDeltas} {
		if len(b) != 0 && len(b) < n*p.NCoefs*p.Steps {
			return errors.New("auditory_hip: delta buffer too small")
		}
	}
	if len(energy) != 0 && len(energy) < n*p.Steps {
		return errors.New("auditory_hip: energy buffer too small")
	}
	ptr := func(s []float64) *C.double {
		if len(s) == 0 {
			return nil
		}
		return (*C.double)(unsafe.Pointer(&s[0]))
	}
	rc := C.aud_melspec_mfcc_batch_sig(p.h, sig.h, (*C.aud_item)(unsafe.Pointer(&items[0])), C.int(len(items)), ptr(mel),
		ptr(power), ptr(logPower), ptr(mfcc), ptr(deltas), ptr(deltaDeltas), ptr(energy))
	return status(p.ctx, rc)
}


// MelSpecLive is MelSpec on the LIVE tensor sig through its resident copy dev (nil: created): aud_melspec_batch_live compares the
// 4 KB blocks the items' frames read with the copy's host shadow, uploads what differs and runs on the device copy -- the result
// is MelSpec's on sig as it is now.  Returns the (possibly new) Signal and the bytes that crossed the link.
func (p *Plan) MelSpecLive(dev *Signal, sig []float64, items []Item, mel, power, logPower []float64) (*Signal, int64, error) {
	if dev == nil {
		dev = &Signal{ctx: p.ctx}
	}
	if len(items) == 0 {
		return dev, 0, nil
	}
	if len(mel) < len(items)*p.NFilters*p.Steps || len(sig) == 0 {
		return dev, 0, errors.New("auditory_hip: mel buffer too small, or empty signal")
	}
	ptr := func(s []float64) *C.double {
		if len(s) == 0 {
			return nil
		}
		return (*C.double)(unsafe.Pointer(&s[0]))
	}
	var up C.int64_t
	rc := C.aud_melspec_batch_live(p.h, &dev.h, ptr(sig), C.int64_t(len(sig)), (*C.aud_item)(unsafe.Pointer(&items[0])),
		C.int(len(items)), ptr(mel), ptr(power), ptr(logPower), &up)
	return dev, int64(up), status(p.ctx, rc)
}

// MelSpecMFCCLive is MelSpecMFCC on the live tensor sig through its resident copy (aud_melspec_mfcc_batch_live).
func (p *Plan) MelSpecMFCCLive(dev *Signal, sig []float64, items []Item, mel, power, logPower, mfcc, deltas, deltaDeltas, energy []float64) (*Signal, int64, error) {
	if dev == nil {
		dev = &Signal{ctx: p.ctx}
	}
	if len(items) == 0 {
		return dev, 0, nil
	}
	if len(mel) < len(items)*p.NFilters*p.Steps || len(mfcc) < len(items)*p.NCoefs*p.Steps || len(sig) == 0 {
		return dev, 0, errors.New("auditory_hip: mel / mfcc buffer too small, or empty signal")
	}
	ptr := func(s []float64) *C.double {
		if len(s) == 0 {
			return nil
		}
		return (*C.double)(unsafe.Pointer(&s[0]))
	}
	var up C.int64_t
	rc := C.aud_melspec_mfcc_batch_live(p.h, &dev.h, ptr(sig), C.int64_t(len(sig)), (*C.aud_item)(unsafe.Pointer(&items[0])),
		C.int(len(items)), ptr(mel), ptr(power), ptr(logPower), ptr(mfcc), ptr(deltas), ptr(deltaDeltas), ptr(energy), &up)
	return dev, int64(up), status(p.ctx, rc)
}
