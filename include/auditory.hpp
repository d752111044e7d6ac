// auditory.hpp -- C++ host-side mirror of the reference's Go packages for the hot path, over the
// C ABI of auditory_hip.h.  Header-only; link with -lauditory_hip.
//
// The reference is compiled Go with no FFI; a Go toolchain is not available in this pipeline, so the
// host layer a Go maintainer would write with cgo (go/auditoryhip) is provided here in
// C++ with the reference's package / type / field / method names:
//
//   namespace dft    { struct Params }                       dft/dft.go:15-39
//   namespace mel    { struct FilterBank, struct Params }    mel/mel.go:16-117, :156-180
//   namespace agabor { struct Filter, struct FilterSet, Active, ToTensor, Convolve }   agabor/gabor.go
//   namespace sound  { MSecToSamples, struct Params, struct SndEnv }                   sound/sndenv.go
//
// Tensors are etensor-like: row-major `Values` plus `Shape`.  Error behaviour follows the Go code:
// Init returns an error string ("" = nil); ProcessSegment / ApplyGabor / Convolve print and carry on.
#ifndef AUDITORY_HPP
#define AUDITORY_HPP

#include <algorithm>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "auditory_hip.h"

namespace auditory {

// minimal etensor.Float64 / Float32: row-major values + shape
template <typename T>
struct Tensor {
    std::vector<int> Shape;
    std::vector<T> Values;
    void SetShape(std::vector<int> shp) {
        Shape = std::move(shp);
        size_t n = 1;
        for (int d : Shape) n *= size_t(d);
        Values.assign(n, T(0));  // etensor.SetShape zero-fills
    }
    int NumDims() const { return int(Shape.size()); }
    int Dim(int i) const { return Shape[size_t(i)]; }
    void SetZeros() { Values.assign(Values.size(), T(0)); }
};
using Float64 = Tensor<double>;
using Float32 = Tensor<float>;

// one device context shared by the mirrors (one process per GPU)
inline aud_ctx*& default_ctx() {
    static aud_ctx* ctx = nullptr;
    return ctx;
}
inline int ensure_ctx(int device = 0) {
    if (default_ctx()) return AUD_OK;
    return aud_init(device, &default_ctx());  // AUD_EHIP without a GPU: there is no CPU fallback
}

namespace dft {
struct Params {  // dft/dft.go:15-31
    bool CompLogPow = false;
    double LogMin = 0, LogOffSet = 0, PrevSmooth = 0, CurSmooth = 0;
    void Defaults() {  // dft/dft.go:33-39
        aud_dft_params c{};
        c.prev_smooth = PrevSmooth;
        aud_dft_defaults(&c);
        CompLogPow = c.comp_log_pow != 0;
        LogMin = c.log_min;
        LogOffSet = c.log_offset;
        PrevSmooth = c.prev_smooth;
        CurSmooth = c.cur_smooth;
    }
    aud_dft_params c() const { return aud_dft_params{CompLogPow ? 1 : 0, LogMin, LogOffSet, PrevSmooth, CurSmooth}; }

    // The reference's per-step entry points, one frame per GPU round trip (correct, never fast; batch at
    // ProcessSegment level).  `plan` is the plan of these parameters (SndEnv::plan.p).
    // dft/dft.go:42-50
    int Filter(aud_plan* plan, int step, const Float64& windowIn, Float64* power, Float64* logPower,
               Float64* powerForSegment, Float64* logPowerForSegment) const {
        return aud_dft_filter_host(plan, step, windowIn.Values.data(), power->Values.data(),
                                   logPower ? logPower->Values.data() : nullptr, powerForSegment->Values.data(),
                                   logPowerForSegment ? logPowerForSegment->Values.data() : nullptr);
    }
    // dft/dft.go:53-59
    static void FftReal(std::vector<std::complex<double>>* fftCoefs, const Float64& in) {
        for (size_t i = 0; i < fftCoefs->size(); ++i) (*fftCoefs)[i] = std::complex<double>(in.Values[i], 0.0);
    }
    // dft/dft.go:62-85 on coefficients the caller computed
    int Power(aud_plan* plan, int step, const std::vector<std::complex<double>>& fftCoefs, Float64* power,
              Float64* logPower, Float64* powerForSegment, Float64* logPowerForSegment) const {
        return aud_dft_power_host(plan, step, reinterpret_cast<const double*>(fftCoefs.data()), power->Values.data(),
                                  logPower ? logPower->Values.data() : nullptr, powerForSegment->Values.data(),
                                  logPowerForSegment ? logPowerForSegment->Values.data() : nullptr);
    }
};
}  // namespace dft

namespace mel {
inline double FreqToMel(double f) { return aud_freq_to_mel(f); }                                  // mel.go:156
inline double MelToFreq(double m) { return aud_mel_to_freq(m); }                                  // mel.go:161
inline int FreqToBin(double f, double nFft, double sr) { return aud_freq_to_bin(f, nFft, sr); }   // mel.go:166

struct FilterBank {  // mel/mel.go:16-44
    int NFilters = 0;
    double LoHz = 0, HiHz = 0, LogOff = 0, LogMin = 0;
    bool Renorm = false;
    double RenormMin = 0, RenormMax = 0, RenormScale = 0;
    void Defaults() {  // mel/mel.go:171-180
        aud_mel_fbank c{};
        aud_mel_defaults(&c);
        from(c);
    }
    aud_mel_fbank c() const {
        return aud_mel_fbank{NFilters, LoHz, HiHz, LogOff, LogMin, Renorm ? 1 : 0, RenormMin, RenormMax, RenormScale};
    }
    void from(const aud_mel_fbank& c) {
        NFilters = c.n_filters; LoHz = c.lo_hz; HiHz = c.hi_hz; LogOff = c.log_off; LogMin = c.log_min;
        Renorm = c.renorm != 0; RenormMin = c.renorm_min; RenormMax = c.renorm_max; RenormScale = c.renorm_scale;
    }
};

struct Params {  // mel/mel.go:47-66
    FilterBank FBank;
    std::vector<int32_t> BinPts;
    std::vector<double> HzPts;
    bool MFCC = false, Deltas = false;
    int NCoefs = 0;
    void Defaults() {  // mel/mel.go:69-74
        FBank.Defaults();
        MFCC = true;
        NCoefs = 13;
        Deltas = true;
    }
    // mel/mel.go:77-117.  Returns false where the Go code would panic (triangle past the end of the tensor).
    bool InitFilters(int dftSize, int sampleRate, Float64* filters) {
        const int nf = FBank.NFilters;
        BinPts.assign(size_t(nf) + 2, 0);
        HzPts.assign(size_t(nf) + 2, 0.0);
        filters->SetShape({nf, nf + 2});
        aud_mel_fbank c = FBank.c();
        const int rc = aud_mel_init_filters(&c, dftSize, sampleRate, BinPts.data(), HzPts.data(), filters->Values.data());
        FBank.from(c);
        return rc == AUD_OK;
    }
    // mel/mel.go:120-153 for one step (`filters` lives on the device inside the plan)
    int FilterDft(aud_plan* plan, int step, const Float64& dftPowerOut, Float64* segmentData, Float64* fBankData) const {
        return aud_mel_filter_dft_host(plan, step, dftPowerOut.Values.data(), segmentData->Values.data(),
                                       fBankData ? fBankData->Values.data() : nullptr);
    }
    // mel/mel.go:192-212 for one step (plan created with mfcc_coefs = NCoefs)
    int CepstrumDct(aud_plan* plan, int step, const Float64& fBankData, Float64* mfccSegment, Float64* mfccDct) const {
        return aud_cepstrum_dct_host(plan, step, fBankData.Values.data(), mfccSegment->Values.data(),
                                     mfccDct ? mfccDct->Values.data() : nullptr);
    }
};
}  // namespace mel

namespace agabor {
struct Filter {  // agabor/gabor.go:17-42
    bool Off = false;
    double WaveLen = 0, Orientation = 0, SigmaWidth = 0, SigmaLength = 0, PhaseOffset = 0;
    bool CircleEdge = false, Circular = false;
    aud_gabor_spec c() const {
        return aud_gabor_spec{Off ? 1 : 0, WaveLen, Orientation, SigmaWidth, SigmaLength, PhaseOffset,
                              CircleEdge ? 1 : 0, Circular ? 1 : 0};
    }
};

struct FilterSet {  // agabor/gabor.go:45-70
    int SizeX = 0, SizeY = 0, StrideX = 0, StrideY = 0;
    double Gain = 0;
    bool Distribute = false;
    Float64 Filters;
    aud_gabor_set c() const { return aud_gabor_set{SizeX, SizeY, StrideX, StrideY, Gain, Distribute ? 1 : 0}; }
};

inline std::vector<Filter> Active(const std::vector<Filter>& specs) {  // gabor.go:329-336
    std::vector<Filter> a;
    for (const Filter& s : specs)
        if (!s.Off) a.push_back(s);
    return a;
}

inline void ToTensor(const std::vector<Filter>& specs, FilterSet* set) {  // gabor.go:89-222
    std::vector<aud_gabor_spec> cs;
    for (const Filter& s : specs) cs.push_back(s.c());
    const int n_act = int(Active(specs).size());
    set->Filters.SetShape({n_act, set->SizeY, set->SizeX});
    if (cs.empty()) return;
    aud_gabor_set g = set->c();
    int n_out = 0;
    aud_gabor_to_tensor(cs.data(), int(cs.size()), &g, set->Filters.Values.data(), &n_out);
}
}  // namespace agabor

// a plan bound to one (window, step, segment, mel table, gabor set) parameter set
struct PlanHandle {
    aud_plan* p = nullptr;
    ~PlanHandle() {
        if (p) aud_plan_destroy(p);
    }
    PlanHandle() = default;
    PlanHandle(const PlanHandle&) = delete;
    PlanHandle& operator=(const PlanHandle&) = delete;
};

namespace agabor {
// gabor.go:225-315 on the GPU through an existing plan; logs and returns on rejected shapes, like the Go code
inline void Convolve(aud_plan* plan, const Float64& melData, const FilterSet&, Float32* rawOut, bool byTime) {
    std::vector<int32_t> shp(rawOut->Shape.begin(), rawOut->Shape.end());
    const int rc = aud_gabor_batch_host(plan, melData.Values.data(), 1, melData.Dim(0), melData.Dim(1),
                                        int(shp.size()), shp.data(), byTime ? 1 : 0, rawOut->Values.data());
    if (rc != AUD_OK) std::fprintf(stderr, "agabor.Convolve: %s\n", aud_status_string(rc));
}
}  // namespace agabor

// kwta.KWTA of github.com/emer/vision v1.1.15 (parameters only; the settling runs on the GPU).  A
// default-constructed value is all zeros like the Go zero value; Defaults() as KWTA.Defaults().
namespace kwta {
struct KWTA : aud_kwta_params {
    KWTA() : aud_kwta_params{} {}
    bool On() const { return on != 0; }
    void Defaults() { aud_kwta_defaults(this); }
    // KWTAPool(raw, act, inhib, extGi) / KWTALayer(raw, act, extGi) with extGi == zeros: act is in/out,
    // inhib the carried [pools][2] state (resized like the Go slice)
    int KWTAPool(const Float32& raw, Float32* act, std::vector<float>* inhib) const {
        const size_t pools = size_t(raw.Dim(0)) * raw.Dim(1);
        if (inhib && inhib->size() != 2 * pools) inhib->assign(2 * pools, 0.f);
        int32_t cyc = 0;
        const int rc = aud_kwta_batch_host(default_ctx(), this, raw.Values.data(), act->Values.data(), 1, raw.Dim(0),
                                           raw.Dim(1), raw.Dim(2), raw.Dim(3), 1, 0, inhib ? inhib->data() : nullptr,
                                           0, &cyc);
        if (rc != AUD_OK) std::fprintf(stderr, "kwta.KWTAPool: %s\n", aud_last_error(default_ctx()));
        return cyc;
    }
    int KWTALayer(const Float32& raw, Float32* act) const {
        int32_t cyc = 0;
        const int rc = aud_kwta_batch_host(default_ctx(), this, raw.Values.data(), act->Values.data(), 1,
                                           int(raw.Values.size()), 1, 1, 1, 0, 0, nullptr, 0, &cyc);
        if (rc != AUD_OK) std::fprintf(stderr, "kwta.KWTALayer: %s\n", aud_last_error(default_ctx()));
        return cyc;
    }
};
}  // namespace kwta

namespace sound {
inline int MSecToSamples(double ms, int rate) { return aud_msec_to_samples(ms, rate); }  // sndenv.go:522-524
inline double SamplesToMSec(int samples, int rate) { return aud_samples_to_msec(samples, rate); }  // :527-529

struct Params {  // sound/sndenv.go:24-61
    double WinMs = 0, StepMs = 0, SegmentMs = 0, StrideMs = 0;
    int BorderSteps = 0, Channel = 0;
    int WinSamples = 0, StepSamples = 0, SegmentSamples = 0, StrideSamples = 0, SegmentSteps = 0;
    std::vector<int> Steps;
};

struct SndEnv {  // sound/sndenv.go:73-182 (hot-path fields)
    Params Params_;  // `Params` in Go; renamed only because C++ cannot reuse the type name
    int SampleRate = 0, Channels = 1;  // se.Sound.SampleRate() / Channels()
    Float64 Signal;
    int SegCnt = 0;
    dft::Params DFT;
    mel::Params Mel;
    Float64 MelFilters, PowerSegment, LogPowerSegment, MelFBankSegment;
    Float64 Energy, MFCCSegment, MFCCDeltas, MFCCDeltaDeltas;  // sndenv.go:104-131 (filled when Mel.MFCC)
    std::vector<agabor::Filter> GaborSpecs;
    agabor::FilterSet GaborFilters;
    int GborOutPoolsX = 0, GborOutPoolsY = 0, GborOutUnitsX = 0, GborOutUnitsY = 0;
    Float32 GborOutput;
    Float32 GborKwta;            // post-kwta output (sndenv.go:163)
    std::vector<float> Inhibs;   // pool-level FFFB state carried between calls (fffb.Inhibs, :166)
    Float32 ExtGi;               // stays zero: NeighInhib is not built (:169-172)
    kwta::KWTA Kwta;
    bool KwtaPool = false;
    bool ByTime = false;
    int ComputeDtype = AUD_F64;  // the reference's arithmetic; AUD_FAST_F32 is the explicit opt-in
    PlanHandle plan;
    aud_plan_desc plan_desc_{};  // what `plan` was created from (ensure_plan)
    // ProcessSegment runs once per segment on the SAME Signal, and the reference reads the LIVE tensor at every step
    // (sndenv.go:455-478): the device keeps a copy between calls that is validated EXACTLY on every call.
    //   Residency = Auto (default): for a Signal of up to AUD_RESIDENT_AUTO_BYTES every call compares what ITS frames read (4 KB
    //       blocks) byte for byte with the host shadow of the device copy and uploads what differs (aud_melspec_batch_live /
    //       aud_melspec_mfcc_batch_live) -- any in-place edit is seen by the call that reads it; a larger one is copied per call.
    //   Residency = Snapshot, or an explicit SignalToDevice(): the caller opts in to a snapshot it keeps current itself --
    //       re-taken when Signal.Values is other memory or another length, SignalChanged() after an in-place edit.
    //   Residency = PerCall: copy per call.
    enum ResidencyMode { Auto = 0, Snapshot = 1, PerCall = 2 };
    ResidencyMode Residency = Auto;
    aud_signal* dev_sig_ = nullptr;
    bool snapshot_ = false;        // dev_sig_ is an opted-in snapshot of (snap_data_, snap_n_)
    const double* snap_data_ = nullptr;
    size_t snap_n_ = 0;
    int64_t last_uploaded_bytes = 0;  // what the last ProcessSegment moved of the Signal (diagnostic)
    ~SndEnv() { drop_resident(); }
    void drop_resident() {
        if (dev_sig_) aud_signal_destroy(dev_sig_);
        dev_sig_ = nullptr;
        snapshot_ = false;
        snap_data_ = nullptr;
        snap_n_ = 0;
    }
    // New: after changing samples of Signal.Values IN PLACE while a snapshot is resident (not needed in the default mode)
    void SignalChanged() { snap_data_ = nullptr; }
    // New: opt in to a resident SNAPSHOT of Signal, taken NOW (aud_signal_upload), whatever its size
    bool SignalToDevice() {
        if (ensure_ctx() != AUD_OK) return false;
        drop_resident();
        if (aud_signal_upload(default_ctx(), Signal.Values.data(), AUD_F64, int64_t(Signal.Values.size()), &dev_sig_) != AUD_OK) return false;
        snapshot_ = true;
        snap_data_ = Signal.Values.data();
        snap_n_ = Signal.Values.size();
        last_uploaded_bytes = int64_t(snap_n_ * 8);
        return true;
    }
    bool resident() {  // true: dev_sig_ holds (snapshot) or will hold (live_) exactly what this call must read
        last_uploaded_bytes = 0;
        live_ = false;
        if (Residency == PerCall || Signal.Values.empty()) return false;
        if (Residency == Snapshot || snapshot_) {
            if (!dev_sig_ || snap_data_ != Signal.Values.data() || snap_n_ != Signal.Values.size()) return SignalToDevice();
            return true;
        }
        if (Signal.Values.size() * sizeof(double) > size_t(AUD_RESIDENT_AUTO_BYTES)) {
            drop_resident();
            return false;
        }
        live_ = true;  // the aud_*_live calls validate what they read, exactly, as part of the call
        return true;
    }
    bool live_ = false;  // set by resident(): this call goes through aud_melspec_batch_live / aud_melspec_mfcc_batch_live

    void ParamDefaults() {  // sndenv.go:64-71
        aud_sound_params c{};
        aud_sound_params_defaults(&c);
        Params_.WinMs = c.win_ms; Params_.StepMs = c.step_ms; Params_.SegmentMs = c.segment_ms;
        Params_.StrideMs = c.stride_ms; Params_.Channel = c.channel; Params_.BorderSteps = c.border_steps;
    }
    void Defaults() {  // sndenv.go:185-192
        ParamDefaults();
        Mel.Defaults();
        Kwta.Defaults();
        KwtaPool = true;
        ByTime = false;
    }

    // sndenv.go:195-267.  "" = nil error.
    std::string Init() {
        aud_sound_params c{};
        c.win_ms = Params_.WinMs; c.step_ms = Params_.StepMs; c.segment_ms = Params_.SegmentMs;
        c.stride_ms = Params_.StrideMs; c.border_steps = Params_.BorderSteps; c.channel = Params_.Channel;
        if (aud_sound_params_derive(&c, SampleRate) != AUD_OK) {
            std::printf("sample rate <= 0\n");
            return "sample rate <= 0";
        }
        Params_.WinSamples = c.win_samples; Params_.StepSamples = c.step_samples;
        Params_.SegmentSamples = c.segment_samples; Params_.SegmentSteps = c.segment_steps;
        Params_.StrideSamples = c.stride_samples;

        agabor::ToTensor(agabor::Active(GaborSpecs), &GaborFilters);
        if (GborOutPoolsX == 0 && GborOutPoolsY == 0) GborOutput.SetShape({GborOutUnitsY, GborOutUnitsX});
        else if (GborOutPoolsX > 0 && GborOutPoolsY > 0)
            GborOutput.SetShape({GborOutPoolsY, GborOutPoolsX, GborOutUnitsY, GborOutUnitsX});
        else {
            std::fprintf(stderr, "GborOutPoolsX & GborOutPoolsY must both be == 0 or > 0 (i.e. 2D or 4D)\n");
            return "";
        }
        ExtGi.SetShape(GborOutput.Shape);
        GborKwta.SetShape(GborOutput.Shape);
        const int H = Params_.WinSamples / 2 + 1;
        DFT.Defaults();
        if (!Mel.InitFilters(Params_.WinSamples, SampleRate, &MelFilters)) return "mel filter table overflow";
        PowerSegment.SetShape({H, Params_.SegmentSteps});
        LogPowerSegment.SetShape({H, Params_.SegmentSteps});
        Params_.Steps.clear();
        for (int i = 0; i < Params_.SegmentSteps; ++i) Params_.Steps.push_back(Params_.StepSamples * (i - Params_.BorderSteps));
        MelFBankSegment.SetShape({Mel.FBank.NFilters, Params_.SegmentSteps});
        Energy.SetShape({Params_.SegmentSteps});
        if (Mel.MFCC) {  // sndenv.go:252-256
            MFCCSegment.SetShape({Mel.NCoefs, Params_.SegmentSteps});
            MFCCDeltas.SetShape({Mel.NCoefs, Params_.SegmentSteps});
            MFCCDeltaDeltas.SetShape({Mel.NCoefs, Params_.SegmentSteps});
        }
        SegCnt = aud_seg_cnt(int(Signal.Values.size()), Params_.SegmentSamples, Params_.StrideSamples, Channels);

        if (ensure_ctx() != AUD_OK) return "no HIP device (libauditory_hip has no CPU fallback)";
        if (plan.p) { aud_plan_destroy(plan.p); plan.p = nullptr; }
        drop_resident();  // (a resident copy belongs to the Signal it was taken from)
        return ensure_plan() ? "" : aud_last_error(default_ctx());
    }

    // The device plan bakes the DFT and mel-bank parameters in, while the reference reads se.DFT / se.Mel.FBank at CALL time
    // (Init resets se.DFT, sndenv.go:230, so PrevSmooth etc. can only be set after it): the plan is keyed on them and
    // rebuilt lazily by the call that finds them changed.
    static bool same_plan(const aud_plan_desc& a, const aud_plan_desc& b) {  // field by field: the structs carry padding
        const aud_dft_params &x = a.dft, &y = b.dft;
        const aud_mel_fbank &m = a.mel, &n = b.mel;
        return a.win_samples == b.win_samples && a.step_samples == b.step_samples && a.segment_steps == b.segment_steps &&
               a.border_steps == b.border_steps && x.comp_log_pow == y.comp_log_pow && x.log_min == y.log_min &&
               x.log_offset == y.log_offset && x.prev_smooth == y.prev_smooth && x.cur_smooth == y.cur_smooth &&
               m.n_filters == n.n_filters && m.lo_hz == n.lo_hz && m.hi_hz == n.hi_hz && m.log_off == n.log_off &&
               m.log_min == n.log_min && m.renorm == n.renorm && m.renorm_min == n.renorm_min && m.renorm_max == n.renorm_max &&
               m.renorm_scale == n.renorm_scale && a.n_gabor == b.n_gabor && a.compute_dtype == b.compute_dtype &&
               a.mfcc_coefs == b.mfcc_coefs;  // (the gabor set and the tables only change in Init, which drops the plan)
    }
    bool ensure_plan() {
        aud_plan_desc d{};
        d.win_samples = Params_.WinSamples; d.step_samples = Params_.StepSamples;
        d.segment_steps = Params_.SegmentSteps; d.border_steps = Params_.BorderSteps;
        d.dft = DFT.c(); d.mel = Mel.FBank.c();
        d.n_gabor = GaborFilters.Filters.NumDims() == 3 ? GaborFilters.Filters.Dim(0) : 0;
        d.gabor = GaborFilters.c();
        d.compute_dtype = ComputeDtype;
        d.mfcc_coefs = Mel.MFCC ? Mel.NCoefs : 0;
        if (plan.p && same_plan(d, plan_desc_)) return true;
        if (plan.p) { aud_plan_destroy(plan.p); plan.p = nullptr; }
        if (aud_plan_create(default_ctx(), &d, Mel.BinPts.data(), MelFilters.Values.data(),
                            d.n_gabor ? GaborFilters.Filters.Values.data() : nullptr, &plan.p) != AUD_OK)
            return false;
        plan_desc_ = d;
        return true;
    }

    // sndenv.go:342-359 (frame loop): PowerSegment, LogPowerSegment, MelFBankSegment of one segment
    void ProcessSegment(int segment, int add) {
        aud_item it{0, int32_t(Signal.Values.size()),
                    int32_t(segment * Params_.StrideSamples + MSecToSamples(double(add), SampleRate))};  // :440-441
        int rc;
        if (!ensure_plan()) {
            std::printf("%s\n", aud_last_error(default_ctx()));
            return;
        }
        const bool resident = this->resident();
        double* lp = DFT.CompLogPow ? LogPowerSegment.Values.data() : nullptr;
        if (Mel.MFCC && DFT.CompLogPow) {  // the MFCC tail of the loop too: CepstrumDct, Energy, deltas (:360-432)
            double* dl = Mel.Deltas ? MFCCDeltas.Values.data() : nullptr;
            double* ddl = Mel.Deltas ? MFCCDeltaDeltas.Values.data() : nullptr;
            rc = live_ ? aud_melspec_mfcc_batch_live(plan.p, &dev_sig_, Signal.Values.data(), int64_t(Signal.Values.size()), &it, 1,
                                                     MelFBankSegment.Values.data(), PowerSegment.Values.data(), lp,
                                                     MFCCSegment.Values.data(), dl, ddl, Energy.Values.data(), &last_uploaded_bytes)
                 : resident ? aud_melspec_mfcc_batch_sig(plan.p, dev_sig_, &it, 1, MelFBankSegment.Values.data(), PowerSegment.Values.data(),
                                                       lp, MFCCSegment.Values.data(), dl, ddl, Energy.Values.data())
                          : aud_melspec_mfcc_batch_host(plan.p, Signal.Values.data(), int64_t(Signal.Values.size()), &it, 1,
                                                        MelFBankSegment.Values.data(), PowerSegment.Values.data(), lp,
                                                        MFCCSegment.Values.data(), dl, ddl, Energy.Values.data());
        } else {
            rc = live_ ? aud_melspec_batch_live(plan.p, &dev_sig_, Signal.Values.data(), int64_t(Signal.Values.size()), &it, 1,
                                                MelFBankSegment.Values.data(), PowerSegment.Values.data(), lp, &last_uploaded_bytes)
                 : resident ? aud_melspec_batch_sig(plan.p, dev_sig_, &it, 1, MelFBankSegment.Values.data(), PowerSegment.Values.data(), lp)
                          : aud_melspec_batch_host(plan.p, Signal.Values.data(), int64_t(Signal.Values.size()), &it, 1,
                                                   MelFBankSegment.Values.data(), PowerSegment.Values.data(), lp);
        }
        if (rc != AUD_OK) std::printf("%s\n", aud_last_error(default_ctx()));  // fmt.Println(err), sndenv.go:356
    }

    // sndenv.go:313-323
    void ApplyKwta() {
        GborKwta.Values = GborOutput.Values;  // CopyFrom
        if (!Kwta.On()) return;
        if (KwtaPool) {
            if (GborOutput.NumDims() != 4) {  // the Go code panics here (Dim(2) of a 2-D tensor)
                std::fprintf(stderr, "KwtaPool needs the 4-D gabor output\n");
                return;
            }
            Kwta.KWTAPool(GborOutput, &GborKwta, &Inhibs);
        } else {
            Kwta.KWTALayer(GborOutput, &GborKwta);
        }
    }

    // sndenv.go:481-497; NeighInhib is not built (ExtGi stays zero, the reference's state when it is off)
    Float32* ApplyGabor() {
        (void)ensure_plan();
        agabor::Convolve(plan.p, MelFBankSegment, GaborFilters, &GborOutput, ByTime);
        std::fill(ExtGi.Values.begin(), ExtGi.Values.end(), 0.f);
        if (Kwta.On()) {
            ApplyKwta();
            return &GborKwta;
        }
        return &GborOutput;
    }
};
}  // namespace sound

}  // namespace auditory
#endif  // AUDITORY_HPP
