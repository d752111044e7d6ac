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

// The device keeps se.Signal between ProcessSegment calls by default and must never serve a stale copy.  The reference reads
// the live tensor at every step (sound/sndenv.go:455-478), so the resident copy is validated EXACTLY on every call
// (aud_signal_sync: memcmp against a host shadow): a replaced slice of the same length, an in-place overwrite, a ONE-sample
// edit that nobody announces, AdjustForSilence -- each compared with the copy-per-call route (HostSignalPerCall) on the same
// samples.  Then the opt-in snapshot (SignalToDevice) and its SignalChanged contract.
func TestResidentSignalNeverStale(t *testing.T) {
	sr := 16000
	mk := func(f float64) []float64 {
		s := make([]float64, sr/2)
		for i := range s {
			s[i] = 0.3 * math.Sin(2*math.Pi*f*float64(i)/float64(sr))
		}
		return s
	}
	se := &SndEnv{}
	se.Defaults()
	se.Mel.MFCC = false
	se.Signal.SetShape([]int{sr / 2}, nil, nil)
	copy(se.Signal.Values, mk(440))
	se.Sound.Buf = &audio.IntBuffer{Format: &audio.Format{NumChannels: 1, SampleRate: sr}, Data: make([]int, sr/2), SourceBitDepth: 16}
	if err := se.Init(); err != nil {
		t.Fatal(err)
	}
	mel := func(perCall bool) []float64 {
		se.HostSignalPerCall = perCall
		se.ProcessSegment(1, 0)
		return append([]float64(nil), se.MelFBankSegment.Values...)
	}
	check := func(what string) []float64 {
		res, host := mel(false), mel(true)
		for i := range res {
			if res[i] != host[i] {
				t.Fatalf("%s: resident %g, copy per call %g at %d", what, res[i], host[i], i)
			}
		}
		return res
	}
	check("first call")
	if se.devSig == nil {
		t.Fatal("the first ProcessSegment did not keep the Signal on the device")
	}
	mel(false)
	if se.LastUploadedBytes != 0 {
		t.Fatalf("an unchanged Signal moved %d bytes", se.LastUploadedBytes)
	}
	se.Signal.Values = mk(880) // another slice of the same length
	check("replaced slice")
	copy(se.Signal.Values, mk(1320)) // in place, every sample
	before := check("in-place overwrite")
	se.Signal.Values[1601] += 0.25 // in place, ONE sample inside segment 1, NOT announced
	after := check("one-sample edit without SignalChanged")
	same := true
	for i := range after {
		same = same && after[i] == before[i]
	}
	if same {
		t.Fatal("the one-sample edit did not reach the device")
	}
	mel(false)
	se.Signal.Values[3000] -= 0.125
	mel(false)
	if se.LastUploadedBytes != 4096 {
		t.Fatalf("a one-sample edit moved %d bytes, not its 4 KB compare block", se.LastUploadedBytes)
	}
	se.AdjustForSilence(30, 10) // prepends 20 ms
	check("AdjustForSilence")

	// the opt-in snapshot: identity-keyed; an unannounced in-place edit is the caller's business, SignalChanged() retakes it
	if err := se.SignalToDevice(); err != nil {
		t.Fatal(err)
	}
	clean := mel(false)
	se.Signal.Values[2000] += 0.25
	stale := mel(false)
	for i := range stale {
		if stale[i] != clean[i] {
			t.Fatal("a snapshot changed without SignalChanged()")
		}
	}
	se.SignalChanged()
	check("snapshot + SignalChanged")
	se.Close()
}

// PinnedTensors is an opt-in: by default every segment tensor lives on the Go heap, like the reference's.  With it, Init points
// their Values at pinned C memory; the next Init or Close gives the tensors Go-heap copies back (never nil Values under a
// non-empty shape) and frees the blocks.
func TestPinnedTensorsOptInAndLifetime(t *testing.T) {
	sr := 16000
	se := &SndEnv{}
	se.Defaults()
	se.Signal.SetShape([]int{sr / 2}, nil, nil)
	for i := range se.Signal.Values {
		se.Signal.Values[i] = 0.3 * math.Sin(2*math.Pi*440*float64(i)/float64(sr))
	}
	se.Sound.Buf = &audio.IntBuffer{Format: &audio.Format{NumChannels: 1, SampleRate: sr}, Data: make([]int, sr/2), SourceBitDepth: 16}
	if err := se.Init(); err != nil {
		t.Fatal(err)
	}
	if len(se.pinned) != 0 {
		t.Fatal("tensors were pinned without PinnedTensors")
	}
	se.ProcessSegment(1, 0)
	heap := append([]float64(nil), se.MelFBankSegment.Values...)
	se.PinnedTensors = true
	if err := se.Init(); err != nil {
		t.Fatal(err)
	}
	if len(se.pinned) != 7 {
		t.Fatalf("%d tensors pinned, want 7", len(se.pinned))
	}
	se.ProcessSegment(1, 0)
	for i := range heap {
		if heap[i] != se.MelFBankSegment.Values[i] {
			t.Fatalf("pinned route differs at %d", i)
		}
	}
	se.Mel.MFCC = false // the MFCC tensors are not SetShape'd again by this Init: they must keep valid Go-heap Values
	if err := se.Init(); err != nil {
		t.Fatal(err)
	}
	if len(se.MFCCSegment.Values) != se.MFCCSegment.Len() {
		t.Fatal("MFCCSegment lost its Values")
	}
	se.Close()
	if len(se.pinned) != 0 || len(se.MelFBankSegment.Values) != se.MelFBankSegment.Len() {
		t.Fatal("Close left pinned tensors behind")
	}
	for i := range heap {
		_ = se.MelFBankSegment.Values[i] // Go heap again: readable after Close
	}
}
