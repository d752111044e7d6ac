// Package sound is the drop-in for github.com/emer/auditory/sound's SndEnv (sound/sndenv.go:24-536 of the reference):
// same exported types, fields and method signatures.  Init builds a device plan next to the tensors; ProcessSegment is
// ONE call into libauditory_hip.so for the whole segment (all T steps: window extraction, DFT, power, log-power, mel,
// and the MFCC tail when Mel.MFCC is on); ApplyGabor is one more.  The per-step methods keep working through the
// per-step entry points.  ProcessSegments (new) takes every segment of the sound in one launch.
//
// NOT COMPILED IN THIS PIPELINE (no Go toolchain in the build image).  Wave (sound/sound.go) is host I/O and stays the
// reference's own file; only SndEnv is replaced.
package sound

import (
	"errors"
	"fmt"
	"log"
	"runtime"

	"github.com/emer/auditory/go/agabor"
	"github.com/emer/auditory/go/auditoryhip"
	"github.com/emer/auditory/go/dft"
	"github.com/emer/auditory/go/mel"
	"github.com/emer/etable/etable"
	"github.com/emer/etable/etensor"
	"github.com/emer/leabra/fffb"
	"github.com/emer/vision/kwta"
)

// Params: sound/sndenv.go:24-61.
type Params struct {
	WinMs          float64
	StepMs         float64
	SegmentMs      float64
	StrideMs       float64
	BorderSteps    int
	Channel        int
	WinSamples     int
	StepSamples    int
	SegmentSamples int
	StrideSamples  int
	SegmentSteps   int
	Steps          []int
}

// SndEnv: sound/sndenv.go:73-182, field for field.
type SndEnv struct {
	Nm              string
	Dsc             string
	On              bool
	Sound           Wave
	Params          Params
	Signal          etensor.Float64
	SegCnt          int
	Window          etensor.Float64
	DFT             dft.Params
	Power           etensor.Float64
	LogPower        etensor.Float64
	PowerSegment    etensor.Float64
	LogPowerSegment etensor.Float64
	Mel             mel.Params
	MelFBank        etensor.Float64
	MelFBankSegment etensor.Float64
	MelFilters      etensor.Float64
	Energy          etensor.Float64
	MFCCDCT         etensor.Float64
	MFCCSegment     etensor.Float64
	MFCCDeltas      etensor.Float64
	MFCCDeltaDeltas etensor.Float64
	GaborSpecs      []agabor.Filter
	GaborFilters    agabor.FilterSet
	GaborTab        etable.Table
	GborOutPoolsX   int
	GborOutPoolsY   int
	GborOutUnitsX   int
	GborOutUnitsY   int
	GborOutput      etensor.Float32
	GborKwta        etensor.Float32
	Inhibs          fffb.Inhibs
	ExtGi           etensor.Float32
	NeighInhib      kwta.NeighInhib
	Kwta            kwta.KWTA
	KwtaPool        bool
	ByTime          bool

	// ComputeF32 selects the float32 kernels (1.5x faster; a few elements per million miss 1e-5 of the reference).
	// The zero value keeps the reference's arithmetic: float64.
	ComputeF32 bool

	ctx     *auditoryhip.Ctx
	plan    *auditoryhip.Plan       // device plan of planKey's parameters (segmentPlan rebuilds it when they change)
	planKey planKey
	derived auditoryhip.SoundParams // what Init derived (sample counts, steps)
	// ProcessSegment runs once per segment on the SAME Signal, and the reference reads the LIVE tensor at every step
	// (sndenv.go:455-478): the device keeps a copy between calls that is validated EXACTLY on every call.
	//   zero values (default): for a Signal of up to auditoryhip.ResidentAutoBytes every call compares what ITS frames read
	//       (4 KB blocks) byte for byte with the host shadow of the device copy and uploads what differs
	//       (aud_melspec_batch_live / aud_melspec_mfcc_batch_live) -- any in-place edit is seen by the call that reads it; a
	//       larger Signal is copied per call.
	//   ResidentSnapshot = true, or an explicit SignalToDevice(): the caller opts in to a SNAPSHOT it keeps current itself --
	//       re-taken when Signal.Values is other memory or another length, SignalChanged() after an in-place edit.
	//   HostSignalPerCall = true: copy per call, no resident copy at all.
	HostSignalPerCall bool
	ResidentSnapshot  bool
	LastUploadedBytes int64 // what the last ProcessSegment moved of the Signal (diagnostic)
	devSig            *auditoryhip.Signal
	snapshot          bool // devSig is an opted-in snapshot of (snapData, snapN)
	snapData          *float64
	snapN             int
	// PinnedTensors = true (opt-in): the segment tensors ProcessSegment fills (PowerSegment, LogPowerSegment, MelFBankSegment,
	// Energy, the MFCC tensors) keep their Values in pinned host memory (auditoryhip.HostFloat64) and the device writes them
	// directly: 0.20 instead of 0.29 ms per 256 utterances.  LIFETIME: such Values are C memory -- the next Init, Close, or the
	// SndEnv's finalizer frees them, so a caller must not keep a slice of tensor.Values beyond that (copy it).  The zero value
	// leaves every tensor on the Go heap, as the reference's are: results then come through the library's staging buffer.
	PinnedTensors bool
	pinned        []pinnedTensor
	guard         *pinGuard
}

// pinnedTensor: a segment tensor whose Values currently live in C memory
type pinnedTensor struct {
	t *etensor.Float64
	s []float64
}

// pinGuard owns the pinned blocks of one SndEnv.  It is a heap object of its own, so that it can carry a finalizer wherever
// the SndEnv itself lives (embedded by value in a sim's struct, a global, the stack: runtime.SetFinalizer on such a SndEnv
// would panic): when the last SndEnv (copy) that points to it is garbage-collected without Close, the blocks are freed.
type pinGuard struct {
	ctx    *auditoryhip.Ctx
	blocks [][]float64
}

func (g *pinGuard) free() {
	for _, b := range g.blocks {
		g.ctx.HostFree(b)
	}
	g.blocks = nil
}

// ParamDefaults: sound/sndenv.go:64-71.
func (se *SndEnv) ParamDefaults() {
	p := auditoryhip.SoundParamDefaults()
	se.Params.WinMs, se.Params.StepMs, se.Params.SegmentMs, se.Params.StrideMs = p.WinMs, p.StepMs, p.SegmentMs, p.StrideMs
	se.Params.BorderSteps, se.Params.Channel = p.BorderSteps, p.Channel
}

// Defaults: sound/sndenv.go:185-192.
func (se *SndEnv) Defaults() {
	se.ParamDefaults()
	se.DFT.Defaults()
	se.Mel.Defaults()
	se.Kwta.Defaults()
	se.KwtaPool = true
}

// Init: sound/sndenv.go:195-267 -- derived sample counts, tensor shapes, the mel table, SegCnt; then the device plan.
func (se *SndEnv) Init() (err error) {
	sr := se.Sound.SampleRate()
	if sr <= 0 {
		fmt.Println("sample rate <= 0")
		return errors.New("sample rate <= 0")
	}
	d, err := auditoryhip.SoundParamsDerive(se.Params.WinMs, se.Params.StepMs, se.Params.SegmentMs, se.Params.StrideMs, se.Params.BorderSteps, sr)
	if err != nil {
		return err
	}
	se.Params.WinSamples, se.Params.StepSamples = d.WinSamples, d.StepSamples
	se.Params.SegmentSamples, se.Params.StrideSamples, se.Params.SegmentSteps = d.SegmentSamples, d.StrideSamples, d.SegmentSteps
	se.unpinSegmentTensors() // (before any SetShape: a tensor must not keep, or re-slice, memory that is about to be freed)
	T, H, nf := se.Params.SegmentSteps, se.Params.WinSamples/2+1, se.Mel.FBank.NFilters

	// the gabor set and the output tensors' shapes (sndenv.go:209-227): active specs -> taps, 2-D or 4-D output
	specs := agabor.Active(se.GaborSpecs)
	nfilters := len(specs)
	se.GaborFilters.Filters.SetShape([]int{nfilters, se.GaborFilters.SizeY, se.GaborFilters.SizeX}, nil, nil)
	agabor.ToTensor(specs, &se.GaborFilters)
	se.GaborFilters.ToTable(se.GaborFilters, &se.GaborTab)
	if se.GborOutPoolsX == 0 && se.GborOutPoolsY == 0 { // 2D
		se.GborOutput.SetShape([]int{se.GborOutUnitsY, se.GborOutUnitsX}, nil, nil)
		se.ExtGi.SetShape([]int{se.GborOutUnitsY, se.GborOutUnitsX}, nil, nil)
	} else if se.GborOutPoolsX > 0 && se.GborOutPoolsY > 0 { // 4D
		se.GborOutput.SetShape([]int{se.GborOutPoolsY, se.GborOutPoolsX, se.GborOutUnitsY, se.GborOutUnitsX}, nil, nil)
		se.ExtGi.SetShape([]int{se.GborOutPoolsY, se.GborOutPoolsX, 2, nfilters}, nil, nil)
	} else {
		log.Println("GborOutPoolsX & GborOutPoolsY must both be == 0 or > 0 (i.e. 2D or 4D)")
		return err // (nil, as in the reference: :224-226)
	}
	se.GborOutput.SetMetaData("odd-row", "true")
	se.GborOutput.SetMetaData("grid-fill", ".9")
	se.GborKwta.CopyShapeFrom(&se.GborOutput)
	se.GborKwta.CopyMetaData(&se.GborOutput)

	se.DFT.Defaults() // sndenv.go:230: Init RESETS the DFT parameters (a user's PrevSmooth etc. must be set after Init; segmentPlan picks them up)
	se.Mel.InitFilters(se.Params.WinSamples, sr, &se.MelFilters)
	se.Window.SetShape([]int{se.Params.WinSamples}, nil, nil)
	se.Power.SetShape([]int{H}, nil, nil)
	se.LogPower.CopyShapeFrom(&se.Power)
	se.PowerSegment.SetShape([]int{H, T}, nil, nil)
	if se.DFT.CompLogPow {
		se.LogPowerSegment.CopyShapeFrom(&se.PowerSegment)
	}
	se.Params.Steps = make([]int, T)
	for i := range se.Params.Steps {
		se.Params.Steps[i] = se.Params.StepSamples * (i - se.Params.BorderSteps)
	}
	se.MelFBank.SetShape([]int{nf}, nil, nil)
	se.MelFBankSegment.SetShape([]int{nf, T}, nil, nil)
	se.Energy.SetShape([]int{T}, nil, nil)
	if se.Mel.MFCC {
		se.MFCCDCT.SetShape([]int{nf}, nil, nil)
		se.MFCCSegment.SetShape([]int{se.Mel.NCoefs, T}, nil, nil)
		se.MFCCDeltas.SetShape([]int{se.Mel.NCoefs, T}, nil, nil)
		se.MFCCDeltaDeltas.SetShape([]int{se.Mel.NCoefs, T}, nil, nil)
	}
	se.SegCnt = auditoryhip.SegCnt(len(se.Signal.Values), se.Params.SegmentSamples, se.Params.StrideSamples, se.Sound.Channels())

	if se.ctx == nil {
		if se.ctx, err = auditoryhip.Default(); err != nil {
			return err // no HIP device: there is no CPU fallback
		}
	}
	se.pinSegmentTensors()
	if se.plan != nil {
		se.plan.Close()
		se.plan = nil
	}
	se.dropResident() // (a resident copy belongs to the Signal it was taken from)
	se.derived = d
	_, err = se.segmentPlan()
	return err
}

// planKey: every parameter the device plan bakes in that the reference reads at CALL time.  Init resets se.DFT
// (sndenv.go:230), so a user's PrevSmooth / CurSmooth / LogOffSet / LogMin can only be set AFTER Init -- and
// ProcessSegment must see them, as the reference's loop does (dft.go:62-85 reads dft.* per step).
type planKey struct {
	compLogPow                                      bool
	logMin, logOffSet, prevSmooth, curSmooth        float64
	melLogOff, melLogMin                            float64
	renorm                                          bool
	renormMin, renormMax, renormScale               float64
	nCoefs                                          int
	f32                                             bool
}

func (se *SndEnv) curPlanKey() planKey {
	nc := 0
	if se.Mel.MFCC {
		nc = se.Mel.NCoefs
	}
	fb := &se.Mel.FBank
	return planKey{se.DFT.CompLogPow, se.DFT.LogMin, se.DFT.LogOffSet, se.DFT.PrevSmooth, se.DFT.CurSmooth, fb.LogOff, fb.LogMin,
		fb.Renorm, fb.RenormMin, fb.RenormMax, fb.RenormScale, nc, se.ComputeF32}
}

// segmentPlan returns the device plan of the CURRENT parameters, rebuilding it when they differ from the ones it was
// created with (go/dft keys its per-step plan the same way, dft.go stepPlan).
func (se *SndEnv) segmentPlan() (*auditoryhip.Plan, error) {
	key := se.curPlanKey()
	if se.plan != nil && key == se.planKey {
		return se.plan, nil
	}
	if se.plan != nil {
		se.plan.Close()
		se.plan = nil
	}
	p, err := se.ctx.NewSndEnvPlan(se.derived, se.DFT.CompLogPow, se.DFT.LogMin, se.DFT.LogOffSet, se.DFT.PrevSmooth, se.DFT.CurSmooth,
		se.Mel.FBank.NFilters, se.Mel.FBank.LoHz, se.Mel.FBank.HiHz, se.Mel.FBank.LogOff, se.Mel.FBank.LogMin, se.Mel.FBank.Renorm,
		se.Mel.FBank.RenormMin, se.Mel.FBank.RenormMax, se.Mel.FBank.RenormScale, se.Mel.BinPts, se.MelFilters.Values,
		se.GaborFilters.SizeX, se.GaborFilters.SizeY, se.GaborFilters.StrideX, se.GaborFilters.StrideY, se.GaborFilters.Gain,
		se.GaborFilters.Filters.Values, key.nCoefs, !se.ComputeF32)
	if err != nil {
		return nil, err
	}
	se.plan, se.planKey = p, key
	return p, nil
}

// unpinSegmentTensors gives every tensor that was pinned a Go-heap slice of the same length again (the shape stays valid, the
// values are kept) and frees the C memory.  Tensors that were never pinned are not touched.
func (se *SndEnv) unpinSegmentTensors() {
	for _, p := range se.pinned {
		if len(p.t.Values) == len(p.s) && len(p.s) > 0 && &p.t.Values[0] == &p.s[0] { // still the pinned slice
			heap := make([]float64, len(p.s))
			copy(heap, p.s)
			p.t.Values = heap
		}
	}
	se.pinned = nil
	if se.guard != nil {
		se.guard.free()
	}
}

// pinSegmentTensors (PinnedTensors only) moves the Values of every tensor ProcessSegment fills into pinned host memory -- all of
// them or none: the library writes them from the device only when every output of the call lies there.  The tensors keep their
// shapes, strides and names; etensor reads and writes Values through the slice, wherever it lives.
func (se *SndEnv) pinSegmentTensors() {
	se.unpinSegmentTensors()
	if !se.PinnedTensors {
		return
	}
	tensors := []*etensor.Float64{&se.PowerSegment, &se.LogPowerSegment, &se.MelFBankSegment, &se.Energy}
	if se.Mel.MFCC {
		tensors = append(tensors, &se.MFCCSegment, &se.MFCCDeltas, &se.MFCCDeltaDeltas)
	}
	var got []pinnedTensor
	for _, t := range tensors {
		s, err := se.ctx.HostFloat64(len(t.Values))
		if err != nil || (len(t.Values) > 0 && s == nil) {
			for _, g := range got {
				se.ctx.HostFree(g.s)
			}
			return // (the Go-heap tensors of SetShape stay: the staging route)
		}
		got = append(got, pinnedTensor{t, s})
	}
	if se.guard == nil { // a SndEnv that is garbage-collected without Close must not leak its pinned blocks
		se.guard = &pinGuard{ctx: se.ctx}
		runtime.SetFinalizer(se.guard, (*pinGuard).free)
	}
	for _, g := range got {
		for j := range g.s {
			g.s[j] = 0
		}
		g.t.Values = g.s
		se.guard.blocks = append(se.guard.blocks, g.s)
	}
	se.pinned = got
}

// Close (new) releases what the SndEnv holds outside the Go heap: the device plan, the resident Signal and -- with
// PinnedTensors -- the pinned blocks (the tensors get Go-heap copies of their values back, so they stay readable).  A SndEnv
// without PinnedTensors may simply be dropped; Init after Close works.
func (se *SndEnv) Close() {
	se.unpinSegmentTensors()
	se.dropResident()
	if se.plan != nil {
		se.plan.Close()
		se.plan = nil
	}
}

func (se *SndEnv) dropResident() {
	se.devSig.Close()
	se.devSig, se.snapshot, se.snapData, se.snapN = nil, false, nil, 0
}

// SignalChanged (new): after changing samples of se.Signal.Values IN PLACE while a snapshot is resident (ResidentSnapshot or
// SignalToDevice) -- the next ProcessSegment takes the snapshot again.  Not needed in the default mode, which compares the
// whole tensor on every call.
func (se *SndEnv) SignalChanged() { se.snapData = nil }

// SignalToDevice (new): opt in to a resident SNAPSHOT of se.Signal, taken NOW, whatever its size; ProcessSegment /
// ProcessSegments then send only the work items and fetch only the results.  The opt-in lasts until Init.
func (se *SndEnv) SignalToDevice() (err error) {
	if se.ctx == nil {
		if se.ctx, err = auditoryhip.Default(); err != nil {
			return err
		}
	}
	se.dropResident()
	if se.devSig, err = se.ctx.UploadSignal(se.Signal.Values); err == nil {
		se.snapshot, se.snapN = true, len(se.Signal.Values)
		if se.snapN > 0 {
			se.snapData = &se.Signal.Values[0]
		}
		se.LastUploadedBytes = int64(8 * se.snapN)
	}
	return err
}

// resident: how this call reads the Signal -- a snapshot (dev != nil, live false), the default live route (live true: the
// aud_*_live calls compare what they read with the resident copy's shadow and upload what differs, as part of the call), or a
// copy per call (nil, false)
func (se *SndEnv) resident() (dev *auditoryhip.Signal, live bool) {
	se.LastUploadedBytes = 0
	n := len(se.Signal.Values)
	if se.HostSignalPerCall || n == 0 {
		return nil, false
	}
	if se.ResidentSnapshot || se.snapshot { // the opt-in: validated by identity only
		if se.devSig == nil || se.snapData != &se.Signal.Values[0] || se.snapN != n {
			if err := se.SignalToDevice(); err != nil {
				fmt.Println(err)
				return nil, false
			}
		}
		return se.devSig, false
	}
	if 8*n > auditoryhip.ResidentAutoBytes {
		se.dropResident()
		return nil, false
	}
	return se.devSig, true // (devSig may still be nil: the first live call creates it)
}

func (se *SndEnv) item(segment, add int) auditoryhip.Item {
	start0 := segment*se.Params.StrideSamples + MSecToSamples(float64(add), se.Sound.SampleRate()) // sndenv.go:440-441
	return auditoryhip.Item{SigOff: 0, SigLen: int32(len(se.Signal.Values)), Start0: int32(start0), SigStride: 1}
}

// ProcessSegment: sound/sndenv.go:342-435 -- one launch for the T steps of the segment.  A window that runs past the
// signal leaves that step and all later ones zero, as the reference's print-and-break does (:354-358).  The per-step
// tensors Power / LogPower are zeroed like the segment tensors (:343-351; the device path never fills them: they are the
// reference's per-step scratch), and Energy is computed whether or not Mel.MFCC is set (:360-366).
func (se *SndEnv) ProcessSegment(segment, add int) {
	se.Power.SetZeros()
	se.LogPower.SetZeros()
	items := []auditoryhip.Item{se.item(segment, add)}
	plan, err := se.segmentPlan() // the parameters as they are NOW (a PrevSmooth set after Init counts)
	if err != nil {
		fmt.Println(err)
		return
	}
	dev, live := se.resident()
	resident := dev != nil
	if se.Mel.MFCC && se.DFT.CompLogPow { // (the MFCC tail reads LogPowerSegment: include/auditory.hpp takes the same branch)
		if live {
			se.devSig, se.LastUploadedBytes, err = plan.MelSpecMFCCLive(dev, se.Signal.Values, items, se.MelFBankSegment.Values,
				se.PowerSegment.Values, se.LogPowerSegment.Values, se.MFCCSegment.Values, se.MFCCDeltas.Values,
				se.MFCCDeltaDeltas.Values, se.Energy.Values)
		} else if resident {
			err = plan.MelSpecMFCCSig(dev, items, se.MelFBankSegment.Values, se.PowerSegment.Values,
				se.LogPowerSegment.Values, se.MFCCSegment.Values, se.MFCCDeltas.Values, se.MFCCDeltaDeltas.Values, se.Energy.Values)
		} else {
			err = plan.MelSpecMFCC(se.Signal.Values, items, se.MelFBankSegment.Values, se.PowerSegment.Values,
				se.LogPowerSegment.Values, se.MFCCSegment.Values, se.MFCCDeltas.Values, se.MFCCDeltaDeltas.Values, se.Energy.Values)
		}
	} else {
		if live {
			se.devSig, se.LastUploadedBytes, err = plan.MelSpecLive(dev, se.Signal.Values, items, se.MelFBankSegment.Values,
				se.PowerSegment.Values, se.LogPowerSegment.Values)
		} else if resident {
			err = plan.MelSpecSig(dev, items, se.MelFBankSegment.Values, se.PowerSegment.Values, se.LogPowerSegment.Values)
		} else {
			err = plan.MelSpec(se.Signal.Values, items, se.MelFBankSegment.Values, se.PowerSegment.Values, se.LogPowerSegment.Values)
		}
		// Energy[s] = sum over f < SegmentSteps of LogPowerSegment row s (the reference's axis, SURVEY Q8)
		T := se.Params.SegmentSteps
		if !se.DFT.CompLogPow {
			T = 0 // no LogPowerSegment: Energy stays zero
		}
		for s := 0; s < T; s++ {
			e := 0.0
			for f := 0; f < T; f++ {
				e += se.LogPowerSegment.Values[s*T+f]
			}
			se.Energy.Values[s] = e
		}
	}
	if err != nil {
		fmt.Println(err)
	}
}

// ProcessSegments (new): segments first .. first+n-1 of the sound in one launch; mel is [n][nf][T] float64.
func (se *SndEnv) ProcessSegments(first, n, add int, mel []float64) error {
	items := make([]auditoryhip.Item, n)
	for i := range items {
		items[i] = se.item(first+i, add)
	}
	plan, err := se.segmentPlan()
	if err != nil {
		return err
	}
	dev, live := se.resident()
	if live {
		se.devSig, se.LastUploadedBytes, err = plan.MelSpecLive(dev, se.Signal.Values, items, mel, nil, nil)
		return err
	}
	if dev != nil {
		return plan.MelSpecSig(dev, items, mel, nil, nil)
	}
	return plan.MelSpec(se.Signal.Values, items, mel, nil, nil)
}

// ProcessStep: sound/sndenv.go:438-452 (one step: SndToWindow, DFT.Filter, Mel.FilterDft, CepstrumDct).
func (se *SndEnv) ProcessStep(segment, step, add int) error {
	offset := se.Params.Steps[step]
	start := segment*se.Params.StrideSamples + offset + MSecToSamples(float64(add), se.Sound.SampleRate())
	if err := se.SndToWindow(start); err != nil {
		return err
	}
	se.DFT.Filter(step, &se.Window, se.Params.WinSamples, &se.Power, &se.LogPower, &se.PowerSegment, &se.LogPowerSegment)
	se.Mel.FilterDft(step, &se.Power, &se.MelFBankSegment, &se.MelFBank, &se.MelFilters)
	if se.Mel.MFCC {
		se.Mel.CepstrumDct(step, &se.MelFBank, &se.MFCCSegment, &se.MFCCDCT)
	}
	return nil
}

// SndToWindow: sound/sndenv.go:455-478 (left zero pad; "end beyond signal length" past the end).
func (se *SndEnv) SndToWindow(start int) error {
	return auditoryhip.SndToWindow(se.Signal.Values, start, se.Params.WinSamples, se.Window.Values)
}

// ApplyGabor: sound/sndenv.go:481-497.
func (se *SndEnv) ApplyGabor() (tsr *etensor.Float32) {
	shp := make([]int32, se.GborOutput.NumDims())
	for i := range shp {
		shp[i] = int32(se.GborOutput.Dim(i))
	}
	plan, perr := se.segmentPlan()
	if perr != nil {
		log.Println(perr)
		return &se.GborOutput
	}
	if err := plan.Convolve(se.MelFBankSegment.Values, 1, se.MelFBankSegment.Dim(0), se.MelFBankSegment.Dim(1), shp, se.ByTime, se.GborOutput.Values); err != nil {
		log.Println(err) // Convolve logs and returns with rawOut untouched (gabor.go:226-229, :259-262)
	}
	if se.NeighInhib.On {
		se.ApplyNeighInhib()
	} else {
		se.ExtGi.SetZeros()
	}
	if se.Kwta.On {
		se.ApplyKwta()
		return &se.GborKwta
	}
	return &se.GborOutput
}

// ApplyNeighInhib: sound/sndenv.go:303-311 (emer/vision code, not on the device path).
func (se *SndEnv) ApplyNeighInhib() {
	if se.NeighInhib.On {
		se.NeighInhib.Inhib4(&se.GborOutput, &se.ExtGi)
	} else {
		se.ExtGi.SetZeros()
	}
}

// ApplyKwta: sound/sndenv.go:313-323.
func (se *SndEnv) ApplyKwta() {
	se.GborKwta.CopyFrom(&se.GborOutput)
	if !se.Kwta.On {
		return
	}
	if se.NeighInhib.On { // non-zero ExtGi: the Go path of emer/vision
		if se.KwtaPool {
			se.Kwta.KWTAPool(&se.GborOutput, &se.GborKwta, &se.Inhibs, &se.ExtGi)
		} else {
			se.Kwta.KWTALayer(&se.GborOutput, &se.GborKwta, &se.ExtGi)
		}
		return
	}
	shp := [4]int{se.GborOutput.Dim(0), se.GborOutput.Dim(1), se.GborOutput.Dim(2), se.GborOutput.Dim(3)}
	if len(se.Inhibs) < shp[0]*shp[1] {
		se.Inhibs = make(fffb.Inhibs, shp[0]*shp[1])
	}
	st := make([]float32, 2*shp[0]*shp[1])
	auditoryhip.PoolState(se.Inhibs, st)
	if err := se.ctx.Kwta(&se.Kwta, se.GborOutput.Values, se.GborKwta.Values, 1, shp, se.KwtaPool, st); err != nil {
		log.Println(err)
	}
	auditoryhip.SetPoolState(se.Inhibs, st)
}

// ToTensor: sound/sndenv.go:297-300.
func (se *SndEnv) ToTensor() bool {
	se.SignalChanged()
	return se.Sound.SoundToTensor(&se.Signal)
}

// AdjustForSilence / Tail / Pad: sound/sndenv.go:274-294, :503-519.
func (se *SndEnv) AdjustForSilence(add, existing float64) (offset int) {
	off, delta := auditoryhip.AdjustForSilence(add, existing, se.Sound.SampleRate())
	switch {
	case delta < 0:
		se.Signal.Values = se.Signal.Values[-delta:]
	case delta > 0:
		se.Signal.Values = append(make([]float64, delta), se.Signal.Values...)
	}
	se.SignalChanged()
	return off
}

func (se *SndEnv) Tail(signal []float64) int {
	return auditoryhip.Tail(len(signal), se.Params.SegmentSamples, se.Params.StrideSamples)
}

func (se *SndEnv) Pad(signal []float64, value float64) (padded []float64) {
	n := auditoryhip.PadLen(len(signal), se.Params.SegmentSamples, se.Params.StrideSamples, se.Params.StepSamples)
	padded = append(signal, make([]float64, n)...)
	for i := len(signal); i < len(padded); i++ {
		padded[i] = value
	}
	return padded
}

func (se *SndEnv) Name() string { return se.Nm }
func (se *SndEnv) Desc() string { return se.Dsc }

// MSecToSamples / SamplesToMSec: sound/sndenv.go:522-529.
func MSecToSamples(ms float64, rate int) int     { return auditoryhip.MSecToSamples(ms, rate) }
func SamplesToMSec(samples int, rate int) float64 { return auditoryhip.SamplesToMSec(samples, rate) }
