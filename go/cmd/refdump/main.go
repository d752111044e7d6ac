// refdump -- NOT COMPILED IN THIS PIPELINE (no Go toolchain, modules not vendored).  The one program that can PIN the
// oracle: it drives the REAL reference (github.com/emer/auditory v0.9.8) over the WAV inputs written by
// tests/golden/make_ref_inputs.py and dumps what tests/test_golden.py compares the oracle and the HIP path with.
// On a machine with Go and the module cache, FROM THE REPOSITORY ROOT (the paths inside jobs.txt are relative to it):
//     go run ./go/cmd/refdump tests/golden/ref_in/jobs.txt
// (one job per line: wav out-prefix winMs stepMs segMs strideMs border nf loHz hiHz poolsY poolsX seg[,seg...]).
//
// ONE run pins everything the library restates, the k-WTA stage (written from memory of emer/vision) included.  Per job:
//   pass 1, se.Kwta.On = false -- per segment s, the tensors of ProcessSegment (sndenv.go:342-432) and the raw gabor output:
//     <prefix>_s<s>_mel.f64 logpower.f64 energy.f64 mfcc.f64 mfccdeltas.f64 mfccdeltadeltas.f64 gabor.f32
//   pass 2 and 3 (jobs with a gabor set), a fresh SndEnv each with se.Kwta as Defaults() leaves it, KwtaPool true / false,
//   segments in the job's order (KWTAPool carries se.Inhibs from call to call, sndenv.go:166, :313-323) -- what ApplyGabor
//   returns (GborKwta, :490-495):
//     <prefix>_s<s>_kwtapool.f32   <prefix>_s<s>_kwtalayer.f32
//   and once per job the parameter block kwta.KWTA.Defaults() produced, as Go prints it (%+v: field names and values,
//   nested structs in braces) -- tests/golden_cases.py parses it and compares field by field with aud_kwta_defaults:
//     <prefix>_kwta_params.txt
// Files are the tensors' row-major Values, little endian.
package main

import (
	"bufio"
	"encoding/binary"
	"fmt"
	"os"
	"strconv"
	"strings"

	"github.com/emer/auditory/agabor"
	"github.com/emer/auditory/sound"
)

func f(s string) float64 { v, _ := strconv.ParseFloat(s, 64); return v }
func n(s string) int     { v, _ := strconv.Atoi(s); return v }

func dump(fn string, v interface{}) {
	fh, err := os.Create(fn)
	if err != nil {
		panic(err)
	}
	defer fh.Close()
	binary.Write(fh, binary.LittleEndian, v) // []float64 / []float32, little endian, the tensor's row-major Values
}

// setup builds the SndEnv of one job the way a sim does: Defaults, Load, ToTensor, parameters, the gabor set, Init.
func setup(a []string) *sound.SndEnv {
	se := &sound.SndEnv{}
	se.Defaults()
	se.Sound.Load(a[0])
	se.ToTensor()
	se.Params.WinMs, se.Params.StepMs, se.Params.SegmentMs, se.Params.StrideMs = f(a[2]), f(a[3]), f(a[4]), f(a[5])
	se.Params.BorderSteps = n(a[6])
	se.Mel.FBank.NFilters, se.Mel.FBank.LoHz, se.Mel.FBank.HiHz = n(a[7]), f(a[8]), f(a[9])
	py, px := n(a[10]), n(a[11])
	if py > 0 { // the default FilterSet of examples/processspeech/processspeech.go:226-253
		se.GaborFilters.SizeX, se.GaborFilters.SizeY, se.GaborFilters.StrideX, se.GaborFilters.StrideY = 9, 9, 3, 3
		se.GaborFilters.Gain = 2
		for _, o := range []float64{0, 45, 90, 135} {
			for _, ph := range []float64{0, 1.5708} {
				se.GaborSpecs = append(se.GaborSpecs, agabor.Filter{WaveLen: 2, Orientation: o, SigmaWidth: 0.5,
					SigmaLength: 0.5, PhaseOffset: ph, CircleEdge: true})
			}
		}
		se.GborOutPoolsY, se.GborOutPoolsX, se.GborOutUnitsY, se.GborOutUnitsX = py, px, 2, 8
	} else {
		se.GborOutUnitsY, se.GborOutUnitsX = 1, 1
	}
	return se
}

func job(a []string) {
	py := n(a[10])
	segs := strings.Split(a[12], ",")

	// pass 1: the segment loop's tensors and the raw gabor output
	se := setup(a)
	se.Kwta.On = false
	if err := se.Init(); err != nil {
		panic(err)
	}
	for _, s := range segs {
		se.ProcessSegment(n(s), 0)
		dump(fmt.Sprintf("%s_s%s_mel.f64", a[1], s), se.MelFBankSegment.Values)
		dump(fmt.Sprintf("%s_s%s_logpower.f64", a[1], s), se.LogPowerSegment.Values)
		dump(fmt.Sprintf("%s_s%s_energy.f64", a[1], s), se.Energy.Values)
		dump(fmt.Sprintf("%s_s%s_mfcc.f64", a[1], s), se.MFCCSegment.Values)
		dump(fmt.Sprintf("%s_s%s_mfccdeltas.f64", a[1], s), se.MFCCDeltas.Values)
		dump(fmt.Sprintf("%s_s%s_mfccdeltadeltas.f64", a[1], s), se.MFCCDeltaDeltas.Values)
		if py > 0 {
			dump(fmt.Sprintf("%s_s%s_gabor.f32", a[1], s), se.ApplyGabor().Values)
		}
	}
	if py == 0 {
		return
	}

	// pass 2 / 3: ApplyGabor with the k-WTA stage on, pool level and layer level
	for _, pool := range []bool{true, false} {
		se := setup(a) // Defaults(): se.Kwta.Defaults(), se.KwtaPool = true (sndenv.go:185-192)
		if !se.Kwta.On {
			panic("kwta.KWTA.Defaults() left On false: the dump would not exercise the stage")
		}
		se.KwtaPool = pool
		if err := se.Init(); err != nil {
			panic(err)
		}
		kind := "kwtalayer"
		if pool {
			kind = "kwtapool"
			fh, err := os.Create(a[1] + "_kwta_params.txt")
			if err != nil {
				panic(err)
			}
			fmt.Fprintf(fh, "%+v\n", se.Kwta)
			fh.Close()
		}
		for _, s := range segs {
			se.ProcessSegment(n(s), 0)
			tsr := se.ApplyGabor() // GborKwta (sndenv.go:490-495)
			if tsr != &se.GborKwta {
				panic("ApplyGabor did not return GborKwta with Kwta.On")
			}
			dump(fmt.Sprintf("%s_s%s_%s.f32", a[1], s, kind), tsr.Values)
		}
	}
}

func main() {
	fh, err := os.Open(strings.TrimPrefix(os.Args[1], "@"))
	if err != nil {
		panic(err)
	}
	sc := bufio.NewScanner(fh)
	for sc.Scan() {
		if a := strings.Fields(sc.Text()); len(a) == 13 {
			job(a)
		}
	}
}
