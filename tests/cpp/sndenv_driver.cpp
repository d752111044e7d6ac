// Drives include/auditory.hpp the way an emergent simulation drives sound.SndEnv:
//   Defaults -> set fields -> Init -> for each segment { ProcessSegment; ApplyGabor }
// argv: <signal.f64> <sample_rate> <out.bin>.  Output: for every segment, float64 mel [nf*T],
// float64 log-power [H*T], float32 raw gabor [8*2*2*8], float32 post-kwta [8*2*2*8], float64 MFCC [13*T],
// MFCC deltas [13*T], Energy [T], preceded by one int32 header {SegCnt, nf, T, H}.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "auditory.hpp"

using namespace auditory;

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    std::FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 3;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    sound::SndEnv se;
    se.Defaults();
    se.Signal.SetShape({int(bytes / 8)});
    if (std::fread(se.Signal.Values.data(), 8, se.Signal.Values.size(), f) != se.Signal.Values.size()) return 4;
    std::fclose(f);
    se.SampleRate = std::atoi(argv[2]);

    // the processspeech filter set (examples/processspeech/processspeech.go:226-253)
    const double orient[] = {0, 45, 90, 135}, phase[] = {0, 1.5708};
    for (double o : orient)
        for (double ph : phase) {
            agabor::Filter g;
            g.WaveLen = 2.0; g.Orientation = o; g.SigmaWidth = 0.5; g.SigmaLength = 0.5;
            g.PhaseOffset = ph; g.CircleEdge = true;
            se.GaborSpecs.push_back(g);
        }
    se.GaborFilters.SizeX = se.GaborFilters.SizeY = 9;
    se.GaborFilters.StrideX = se.GaborFilters.StrideY = 3;
    se.GaborFilters.Gain = 2;
    se.GborOutPoolsX = 2; se.GborOutPoolsY = 8; se.GborOutUnitsX = 8; se.GborOutUnitsY = 2;

    const std::string err = se.Init();
    if (!err.empty()) {
        std::fprintf(stderr, "Init: %s\n", err.c_str());
        return 5;
    }
    std::FILE* o = std::fopen(argv[3], "wb");
    const int32_t hdr[4] = {se.SegCnt, se.Mel.FBank.NFilters, se.Params_.SegmentSteps, se.Params_.WinSamples / 2 + 1};
    std::fwrite(hdr, 4, 4, o);
    for (int seg = 0; seg < se.SegCnt; ++seg) {
        // the device keeps the Signal between calls by default, validated exactly on every call (aud_signal_sync); segment 2
        // takes the copy-per-call route (same results).  Segment 3: ONE sample edited in place and NOT announced must reach the
        // device (the reference reads the live tensor, sndenv.go:455-478) -- and only its 4 KB compare block crosses the link.
        // Segment 4: the opted-in snapshot (SignalToDevice) serves what it holds until SignalChanged().
        se.Residency = seg == 2 ? sound::SndEnv::PerCall : sound::SndEnv::Auto;
        if (seg == 1 && !se.dev_sig_) return 7;
        auto total = [&se] {
            double t = 0.0;
            for (double v : se.MelFBankSegment.Values) t += v;
            return t;
        };
        if (seg == 3 || seg == 4) {
            const size_t at = size_t(seg * se.Params_.StrideSamples + 101);  // (inside the segment)
            const double keep = se.Signal.Values[at];
            if (seg == 4 && !se.SignalToDevice()) return 12;
            se.ProcessSegment(seg, 0);
            if (se.last_uploaded_bytes != 0) return 13;  // the copy was current: nothing moved
            const double clean = total();
            se.Signal.Values[at] = keep + 0.5;  // no SignalChanged()
            se.ProcessSegment(seg, 0);
            if (seg == 3) {
                if (total() == clean) return 11;  // the edit never reached the device
                if (se.last_uploaded_bytes != 4096) return 14;
            } else {
                if (total() != clean) return 15;  // a snapshot is a snapshot ...
                se.SignalChanged();
                se.ProcessSegment(seg, 0);
                if (total() == clean) return 16;  // ... until the caller says so
            }
            se.Signal.Values[at] = keep;
            if (seg == 4) {
                se.SignalChanged();
                se.ProcessSegment(seg, 0);
                se.drop_resident();  // end of the opt-in: the default mode again
            } else {
                se.ProcessSegment(seg, 0);
            }
            if (total() != clean) return 17;
        }
        se.ProcessSegment(seg, 0);
        Float32* g = se.ApplyGabor();
        if (g != &se.GborKwta) return 6;  // Defaults() leaves Kwta.On (sndenv.go:189, :492-494)
        std::fwrite(se.MelFBankSegment.Values.data(), 8, se.MelFBankSegment.Values.size(), o);
        std::fwrite(se.LogPowerSegment.Values.data(), 8, se.LogPowerSegment.Values.size(), o);
        std::fwrite(se.GborOutput.Values.data(), 4, se.GborOutput.Values.size(), o);
        std::fwrite(g->Values.data(), 4, g->Values.size(), o);
        std::fwrite(se.MFCCSegment.Values.data(), 8, se.MFCCSegment.Values.size(), o);
        std::fwrite(se.MFCCDeltas.Values.data(), 8, se.MFCCDeltas.Values.size(), o);
        std::fwrite(se.Energy.Values.data(), 8, se.Energy.Values.size(), o);
    }
    // the reference's per-step calls for step 2 of segment 0 (sndenv.go:438-452, mel.go:192-212): window ->
    // dft.Filter -> mel.FilterDft -> mel.CepstrumDct; appended as mel [nf] and mfcc [NCoefs]
    {
        const int N = se.Params_.WinSamples, H = N / 2 + 1, T = se.Params_.SegmentSteps, nf = se.Mel.FBank.NFilters;
        Float64 window, power, logPower, pSeg, lSeg, fbank, mSeg, mfccSeg, mfccDct;
        window.SetShape({N}); power.SetShape({H}); logPower.SetShape({H}); pSeg.SetShape({H, T}); lSeg.SetShape({H, T});
        fbank.SetShape({nf}); mSeg.SetShape({nf, T}); mfccSeg.SetShape({se.Mel.NCoefs, T}); mfccDct.SetShape({nf});
        const int step = 2;
        if (aud_snd_to_window(se.Signal.Values.data(), int64_t(se.Signal.Values.size()), se.Params_.Steps[step], N,
                              window.Values.data()) != AUD_OK) return 7;
        if (se.DFT.Filter(se.plan.p, step, window, &power, &logPower, &pSeg, &lSeg) != AUD_OK) return 8;
        if (se.Mel.FilterDft(se.plan.p, step, power, &mSeg, &fbank) != AUD_OK) return 9;
        if (se.Mel.CepstrumDct(se.plan.p, step, fbank, &mfccSeg, &mfccDct) != AUD_OK) return 10;
        std::vector<double> col(size_t(se.Mel.NCoefs));
        for (int i = 0; i < se.Mel.NCoefs; ++i) col[size_t(i)] = mfccSeg.Values[size_t(i) * T + step];
        std::fwrite(fbank.Values.data(), 8, fbank.Values.size(), o);
        std::fwrite(col.data(), 8, col.size(), o);
    }
    std::fclose(o);
    aud_shutdown(default_ctx());
    std::printf("CPP-DRIVER-OK %d segments\n", se.SegCnt);
    return 0;
}
