// Package dft is the drop-in for github.com/emer/auditory/dft: same exported type, fields and method signatures
// (dft/dft.go:15-85 of the reference), bodies routed through libauditory_hip.so (go/auditoryhip).
//
// NOT COMPILED IN THIS PIPELINE (no Go toolchain in the build image; etable cannot be fetched).  Struct tags of the
// reference (GUI hints) are left out; everything a caller can name is here.
package dft

import (
	"log"

	"github.com/emer/auditory/go/auditoryhip"
	"github.com/emer/etable/etensor"
)

// Params: the reference's dft.Params, field for field.
type Params struct {
	CompLogPow bool
	LogMin     float64
	LogOffSet  float64
	PrevSmooth float64
	CurSmooth  float64

	plan    *auditoryhip.Plan // per-step plan (one frame per call), rebuilt when the window length changes
	planKey [7]float64
}

// Defaults: dft/dft.go:33-39 (LogOffSet is 1.0 there, whatever the struct tag says).
func (dft *Params) Defaults() {
	d := auditoryhip.DftDefaults()
	dft.CompLogPow, dft.LogMin, dft.LogOffSet = d.CompLogPow, d.LogMin, d.LogOffSet
	dft.PrevSmooth, dft.CurSmooth = d.PrevSmooth, d.CurSmooth
}

func (dft *Params) stepPlan(winSamples, steps int) *auditoryhip.Plan {
	logPow := 0.0
	if dft.CompLogPow {
		logPow = 1
	}
	key := [7]float64{float64(winSamples), float64(steps), dft.LogMin, dft.LogOffSet, dft.PrevSmooth, dft.CurSmooth, logPow}
	if dft.plan == nil || key != dft.planKey {
		if dft.plan != nil {
			dft.plan.Close()
		}
		p, err := auditoryhip.NewStepPlan(winSamples, steps, dft.CompLogPow, dft.LogMin, dft.LogOffSet, dft.PrevSmooth, dft.CurSmooth)
		if err != nil {
			log.Println(err)
			return nil
		}
		dft.plan, dft.planKey = p, key
	}
	return dft.plan
}

// Filter: dft/dft.go:42-50 -- the DFT of the raw window (length = window length, no taper) and Power.
func (dft *Params) Filter(step int, windowIn *etensor.Float64, winSamples int, power *etensor.Float64, logPower *etensor.Float64, powerForSegment *etensor.Float64, logPowerForSegment *etensor.Float64) {
	p := dft.stepPlan(winSamples, powerForSegment.Dim(1))
	if p == nil {
		return
	}
	if err := p.DftFilter(step, windowIn.Values, power.Values, logPower.Values, powerForSegment.Values, logPowerForSegment.Values); err != nil {
		log.Println(err)
	}
}

// FftReal: dft/dft.go:53-59.
func (dft *Params) FftReal(fftCoefs []complex128, in *etensor.Float64) {
	for i := range fftCoefs {
		fftCoefs[i] = complex(in.FloatVal1D(i), 0)
	}
}

// Power: dft/dft.go:62-85, on coefficients the caller computed.
func (dft *Params) Power(step, winSamples int, fftCoefs []complex128, power *etensor.Float64, logPower *etensor.Float64, powerForSegment *etensor.Float64, logPowerForSegment *etensor.Float64) {
	p := dft.stepPlan(winSamples, powerForSegment.Dim(1))
	if p == nil {
		return
	}
	if err := p.DftPower(step, fftCoefs, power.Values, logPower.Values, powerForSegment.Values, logPowerForSegment.Values); err != nil {
		log.Println(err)
	}
}
