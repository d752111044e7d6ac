// refdump -- NOT COMPILED IN THIS PIPELINE (no Go toolchain, modules not vendored).  The one program that can PIN the
// oracle: it drives the REAL reference (github.com/emer/auditory v0.9.8) over the WAV inputs written by
// tests/golden/make_ref_inputs.py and dumps what tests/test_golden.py compares the oracle and the HIP path with.
// On a machine with Go and the module cache, FROM THE REPOSITORY ROOT (the paths inside jobs.txt are relative to it):
//     go run ./go/cmd/refdump tests/golden/ref_in/jobs.txt
// (one job per line: wav out-prefix winMs stepMs segMs strideMs border nf loHz hiHz poolsY poolsX seg[,seg...]).
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

func job(a []string) {
	se := sound.SndEnv{}
	se.Defaults()
	se.Sound.Load(a[0])
	se.ToTensor()
	se.Params.WinMs, se.Params.StepMs, se.Params.SegmentMs, se.Params.StrideMs = f(a[2]), f(a[3]), f(a[4]), f(a[5])
	se.Params.BorderSteps = n(a[6])
	se.Mel.FBank.NFilters, se.Mel.FBank.LoHz, se.Mel.FBank.HiHz = n(a[7]), f(a[8]), f(a[9])
	se.Kwta.On = false
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
	if err := se.Init(); err != nil {
		panic(err)
	}
	for _, s := range strings.Split(a[12], ",") {
		se.ProcessSegment(n(s), 0)
		dump(fmt.Sprintf("%s_s%s_mel.f64", a[1], s), se.MelFBankSegment.Values)
		dump(fmt.Sprintf("%s_s%s_logpower.f64", a[1], s), se.LogPowerSegment.Values)
		dump(fmt.Sprintf("%s_s%s_mfcc.f64", a[1], s), se.MFCCSegment.Values)
		if py > 0 {
			dump(fmt.Sprintf("%s_s%s_gabor.f32", a[1], s), se.ApplyGabor().Values)
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
