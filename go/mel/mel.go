// Package mel is the drop-in for github.com/emer/auditory/mel (mel/mel.go:16-212 of the reference): same exported
// types, fields and signatures; the table arithmetic and the per-step filter run in libauditory_hip.so.
//
// NOT COMPILED IN THIS PIPELINE (no Go toolchain in the build image).
package mel

import (
	"log"

	"github.com/emer/auditory/go/auditoryhip"
	"github.com/emer/etable/etensor"
)

// FilterBank: mel/mel.go:16-43.
type FilterBank struct {
	NFilters    int
	LoHz        float64
	HiHz        float64
	LogOff      float64
	LogMin      float64
	Renorm      bool
	RenormMin   float64
	RenormMax   float64
	RenormScale float64
}

// Params: mel/mel.go:45-66.
type Params struct {
	FBank   FilterBank
	BinPts  []int32
	HzPts   []float64
	MFCC    bool
	Deltas  bool
	NCoefs  int
	plan    *auditoryhip.Plan
	planKey stepKey
	bins    int       // spectrum length and filter table InitFilters / FilterDft last saw: CepstrumDct, which is handed
	tab     []float64 // neither, runs on the same per-step plan
}

// Defaults: mel/mel.go:69-74 and :171-180.
func (mel *Params) Defaults() {
	mel.MFCC, mel.Deltas, mel.NCoefs = true, true, 13
	mel.FBank.Defaults()
}

func (mfb *FilterBank) Defaults() {
	d := auditoryhip.MelDefaults()
	mfb.NFilters, mfb.LoHz, mfb.HiHz = d.NFilters, d.LoHz, d.HiHz
	mfb.LogOff, mfb.LogMin = d.LogOff, d.LogMin
	mfb.Renorm, mfb.RenormMin, mfb.RenormMax, mfb.RenormScale = d.Renorm, d.RenormMin, d.RenormMax, d.RenormScale
}

// InitFilters: mel/mel.go:77-117 -- BinPts and the [nf, nf+2] triangle table (Renorm is forced off, :80).
func (mel *Params) InitFilters(dftSize int, sampleRate int, filters *etensor.Float64) {
	nf := mel.FBank.NFilters
	mel.BinPts = make([]int32, nf+2)
	mel.HzPts = make([]float64, nf+2)
	filters.SetShape([]int{nf, nf + 2}, nil, nil)
	fb := mel.FBank.toC()
	if err := auditoryhip.MelInitFiltersGo(&fb, dftSize, sampleRate, mel.BinPts, mel.HzPts, filters.Values); err != nil {
		log.Println(err) // where the reference would index past the table and panic
	}
	mel.bins, mel.tab = dftSize/2+1, filters.Values
	mel.FBank.Renorm = false
}

// FilterDft: mel/mel.go:120-153, one step.
func (mel *Params) FilterDft(step int, dftPowerOut *etensor.Float64, segmentData *etensor.Float64, fBankData *etensor.Float64, filters *etensor.Float64) {
	mel.bins, mel.tab = dftPowerOut.Len(), filters.Values
	p := mel.stepPlan(segmentData.Dim(1))
	if p == nil {
		return
	}
	if err := p.MelFilterDft(step, dftPowerOut.Values, segmentData.Values, fBankData.Values); err != nil {
		log.Println(err)
	}
}

// FreqToMel / MelToFreq / FreqToBin: mel/mel.go:156-168.
func FreqToMel(freq float64) float64              { return auditoryhip.FreqToMel(freq) }
func MelToFreq(mel float64) float64               { return auditoryhip.MelToFreq(mel) }
func FreqToBin(freq, nFft, sampleRate float64) int { return auditoryhip.FreqToBin(freq, nFft, sampleRate) }

// FftReal: mel/mel.go:183-189.
func (mel *Params) FftReal(out []complex128, in *etensor.Float64) {
	for i := range out {
		out[i] = complex(in.FloatVal1D(i), 0)
	}
}

// CepstrumDct: mel/mel.go:192-212 (DCT-I of the log-mel values, c0 <- ln(1 + c0^2), NCoefs kept).
func (mel *Params) CepstrumDct(step int, fBankData *etensor.Float64, mfccSegment *etensor.Float64, mfccDct *etensor.Float64) {
	p := mel.stepPlan(mfccSegment.Dim(1)) // (the plan FilterDft runs on when the segment tensors have the same steps)
	if p == nil {
		return
	}
	if err := p.CepstrumDct(step, fBankData.Values, mfccSegment.Values, mfccDct.Values); err != nil {
		log.Println(err)
	}
}

func (mfb *FilterBank) toC() auditoryhip.MelFBank {
	return auditoryhip.MelFBank{NFilters: mfb.NFilters, LoHz: mfb.LoHz, HiHz: mfb.HiHz, LogOff: mfb.LogOff,
		LogMin: mfb.LogMin, Renorm: mfb.Renorm, RenormMin: mfb.RenormMin, RenormMax: mfb.RenormMax,
		RenormScale: mfb.RenormScale}
}

type stepKey struct {
	bins, steps, nCoefs int
	fbank               FilterBank
}

func (mel *Params) stepPlan(steps int) *auditoryhip.Plan {
	nf := mel.FBank.NFilters
	if mel.bins < 2 || len(mel.BinPts) != nf+2 || len(mel.tab) < nf*(nf+2) {
		log.Println("mel: InitFilters has not run for these parameters")
		return nil
	}
	nc := 0
	if mel.MFCC {
		nc = mel.NCoefs
	}
	key := stepKey{mel.bins, steps, nc, mel.FBank} // FilterDft reads FBank.LogOff / LogMin / Renorm* at call time (mel.go:133-149)
	if mel.plan == nil || key != mel.planKey {
		if mel.plan != nil {
			mel.plan.Close()
			mel.plan = nil
		}
		p, err := auditoryhip.NewMelStepPlan(2*(mel.bins-1), steps, mel.FBank.toC(), mel.BinPts, mel.tab, nc)
		if err != nil {
			log.Println(err)
			return nil
		}
		mel.plan, mel.planKey = p, key
	}
	return mel.plan
}
