// Package agabor is the drop-in for github.com/emer/auditory/agabor (agabor/gabor.go:17-336 of the reference): same
// exported types, fields and signatures; tap generation and the convolution run in libauditory_hip.so.
//
// NOT COMPILED IN THIS PIPELINE (no Go toolchain in the build image).
package agabor

import (
	"log"

	"github.com/emer/auditory/go/auditoryhip"
	"github.com/emer/etable/etable"
	"github.com/emer/etable/etensor"
)

// Filter: agabor/gabor.go:17-43.
type Filter struct {
	Off         bool
	WaveLen     float64
	Orientation float64
	SigmaWidth  float64
	SigmaLength float64
	PhaseOffset float64
	CircleEdge  bool
	Circular    bool
}

// FilterSet: agabor/gabor.go:45-70.
type FilterSet struct {
	SizeX      int
	SizeY      int
	StrideX    int
	StrideY    int
	Gain       float64
	Distribute bool
	Filters    etensor.Float64
	Table      etable.Table
}

// Defaults: agabor/gabor.go:73-86 -- zero-valued fields get WaveLen 2, sigmas 0.5 (with the reference's notice).
func (f *Filter) Defaults(i int) {
	s := auditoryhip.GaborSpecOf(f.Off, f.WaveLen, f.Orientation, f.SigmaWidth, f.SigmaLength, f.PhaseOffset, f.CircleEdge, f.Circular)
	auditoryhip.GaborSpecDefaults(&s, i)
	f.WaveLen, f.SigmaWidth, f.SigmaLength = s.WaveLen, s.SigmaWidth, s.SigmaLength
}

// Active: agabor/gabor.go:329-336.
func Active(specs []Filter) (active []Filter) {
	for _, s := range specs {
		if !s.Off {
			active = append(active, s)
		}
	}
	return active
}

// ToTensor: agabor/gabor.go:89-222 -- taps of the active specs into set.Filters [nActive, SizeY, SizeX].
func ToTensor(specs []Filter, set *FilterSet) {
	cs := make([]auditoryhip.GaborSpec, len(specs))
	for i, f := range specs {
		cs[i] = auditoryhip.GaborSpecOf(f.Off, f.WaveLen, f.Orientation, f.SigmaWidth, f.SigmaLength, f.PhaseOffset, f.CircleEdge, f.Circular)
	}
	n := len(Active(specs))
	set.Filters.SetShape([]int{n, set.SizeY, set.SizeX}, nil, nil)
	if _, err := auditoryhip.GaborToTensorGo(cs, set.SizeX, set.SizeY, set.StrideX, set.StrideY, set.Gain, set.Distribute, set.Filters.Values); err != nil {
		log.Println(err)
	}
}

// Convolve: agabor/gabor.go:225-315.  filters by value, rawOut by pointer; on a shape the reference rejects it
// logs and returns with rawOut untouched, as the reference does.
func Convolve(melData *etensor.Float64, filters FilterSet, rawOut *etensor.Float32, byTime bool) {
	p, err := auditoryhip.GaborPlan(filters.SizeX, filters.SizeY, filters.StrideX, filters.StrideY, filters.Gain, filters.Filters.Values)
	if err != nil {
		log.Println(err)
		return
	}
	defer p.Close()
	shp := make([]int32, rawOut.NumDims())
	for i := range shp {
		shp[i] = int32(rawOut.Dim(i))
	}
	if err := p.Convolve(melData.Values, 1, melData.Dim(0), melData.Dim(1), shp, byTime, rawOut.Values); err != nil {
		log.Println(err)
	}
}

// ToTable: agabor/gabor.go:320-327 (GUI helper: the filters as a table of [SizeY, SizeX] cells).
func (fs *FilterSet) ToTable(set FilterSet, tab *etable.Table) {
	tab.SetFromSchema(etable.Schema{{"Filter", etensor.FLOAT64, []int{1, fs.SizeX, fs.SizeY}, []string{"Filter", "Y", "X"}}}, fs.Filters.Dim(0))
	tab.Cols[0].SetFloats(set.Filters.Values)
}
