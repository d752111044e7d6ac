// NOT COMPILED IN THIS PIPELINE (no Go toolchain in the build image).  For the maintainer: `go test ./go/sound` on a GPU box.
package sound

import (
	"math"
	"testing"

	"github.com/go-audio/audio"
)

// A PrevSmooth set AFTER Init must reach ProcessSegment: Init resets se.DFT (sound/sndenv.go:230), so that is the only
// place a user can set it, and the reference's loop reads dft.PrevSmooth per step (dft/dft.go:62-69).  ProcessSegment
// (one device launch, plan keyed on the current parameters) is compared with the ProcessStep loop (per-step plan of
// go/dft, keyed the same way).
func TestPrevSmoothSetAfterInitReachesProcessSegment(t *testing.T) {
	se := &SndEnv{}
	se.Defaults()
	sr := 16000
	sig := make([]float64, sr/2)
	for i := range sig {
		sig[i] = 0.3*math.Sin(2*math.Pi*440*float64(i)/float64(sr)) + 0.1*math.Sin(2*math.Pi*1830*float64(i)/float64(sr))
	}
	// a 16-bit mono buffer as wav.Decoder.FullPCMBuffer would hand it over (Wave is the reference's own file, sound/sound.go:31-33)
	pcm := make([]int, len(sig))
	for i, v := range sig {
		pcm[i] = int(math.Round(v * 0x7FFF))
	}
	se.Sound.Buf = &audio.IntBuffer{Format: &audio.Format{NumChannels: 1, SampleRate: sr}, Data: pcm, SourceBitDepth: 16}
	se.ToTensor()
	se.Mel.MFCC = false
	if err := se.Init(); err != nil {
		t.Fatal(err)
	}
	se.DFT.PrevSmooth, se.DFT.CurSmooth = 0.25, 0.75 // after Init, as the reference requires

	se.ProcessSegment(0, 0)
	T, H, nf := se.Params.SegmentSteps, se.Params.WinSamples/2+1, se.Mel.FBank.NFilters
	mel := append([]float64(nil), se.MelFBankSegment.Values...)
	pow := append([]float64(nil), se.PowerSegment.Values...)

	se.PowerSegment.SetZeros()
	se.MelFBankSegment.SetZeros()
	se.Power.SetZeros()
	for s := 0; s < T; s++ {
		if err := se.ProcessStep(0, s, 0); err != nil {
			break
		}
	}
	for i := 0; i < H*T; i++ {
		if d := math.Abs(pow[i] - se.PowerSegment.Values[i]); d > 1e-5*math.Max(1, math.Abs(pow[i])) {
			t.Fatalf("PowerSegment[%d]: segment call %g, step loop %g", i, pow[i], se.PowerSegment.Values[i])
		}
	}
	for i := 0; i < nf*T; i++ {
		if d := math.Abs(mel[i] - se.MelFBankSegment.Values[i]); d > 1e-5*math.Max(1, math.Abs(mel[i])) {
			t.Fatalf("MelFBankSegment[%d]: segment call %g, step loop %g", i, mel[i], se.MelFBankSegment.Values[i])
		}
	}
	// and the smoothing really happened: with PrevSmooth = 0 the second live step differs
	se.DFT.PrevSmooth, se.DFT.CurSmooth = 0, 1
	se.ProcessSegment(0, 0)
	same := true
	for i := 0; i < H*T && same; i++ {
		same = pow[i] == se.PowerSegment.Values[i]
	}
	if same {
		t.Fatal("PrevSmooth had no effect on ProcessSegment")
	}
}
