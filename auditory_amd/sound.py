"""Host-side mirror of the reference's orchestrator `sound.SndEnv` (sound/sndenv.go) for the
hot path: Init -> ProcessSegment -> ApplyGabor.  The frame loop itself runs on the GPU as one
batched launch per call; the MFCC tail (CepstrumDct, Energy, deltas) runs as three small kernels
behind it."""
import numpy as np

from . import agabor, capi, dft, kwta, mel, runtime


def MSecToSamples(ms, rate):
    """sound/sndenv.go:522-524"""
    return capi.load().aud_msec_to_samples(float(ms), int(rate))


def SamplesToMSec(samples, rate):
    """sound/sndenv.go:527-529"""
    return capi.load().aud_samples_to_msec(int(samples), int(rate))


# sound.SoundSampleType, sound/sound.go:22-30
Unknown, SignedInt, UnSignedInt, Float = 0, 1, 2, 3


class Wave:
    """sound.Wave, sound/sound.go:32-141 (PCM WAV in, normalised float64 out).  Host I/O: the RIFF parsing
    replaces go-audio's decoder; the int -> float rule is Wave.GetFloatAtIdx (sound.go:130-141)."""

    def __init__(self):
        self.Data = np.zeros(0, np.int64)     # interleaved samples, like audio.IntBuffer.Data
        self.SourceBitDepth = 0
        self._rate = 0
        self._channels = 0

    def Load(self, fn):
        """sound.go:37-51: decode a PCM WAV file (8/16/24/32-bit integer samples)"""
        import struct
        raw = open(fn, "rb").read()
        if raw[:4] != b"RIFF" or raw[8:12] != b"WAVE":
            raise ValueError("not a RIFF/WAVE file: %s" % fn)
        pos, fmt, data = 12, None, None
        while pos + 8 <= len(raw):
            cid, size = raw[pos:pos + 4], struct.unpack("<I", raw[pos + 4:pos + 8])[0]
            if cid == b"fmt ":
                fmt = struct.unpack("<HHIIHH", raw[pos + 8:pos + 24])
            elif cid == b"data":
                data = raw[pos + 8:pos + 8 + size]
            pos += 8 + size + (size & 1)
        if fmt is None or data is None or fmt[0] != 1:
            raise ValueError("only integer PCM WAV is supported")
        self._channels, self._rate, self.SourceBitDepth = fmt[1], fmt[2], fmt[5]
        if self.SourceBitDepth == 8:
            self.Data = np.frombuffer(data, np.uint8).astype(np.int64) - 128
        elif self.SourceBitDepth == 16:
            self.Data = np.frombuffer(data, "<i2").astype(np.int64)
        elif self.SourceBitDepth == 24:
            b = np.frombuffer(data[:len(data) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int64)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            self.Data = np.where(v >= 1 << 23, v - (1 << 24), v)
        elif self.SourceBitDepth == 32:
            self.Data = np.frombuffer(data, "<i4").astype(np.int64)
        else:
            raise ValueError("unsupported bit depth %d" % self.SourceBitDepth)
        return None

    def SampleRate(self):
        """sound.go:79-90"""
        return self._rate

    def Channels(self):
        """sound.go:93-103"""
        return self._channels

    def SampleSize(self):
        """sound.go:88-94 (the reference returns 16 whatever the file holds)"""
        return 16

    def SampleType(self):
        """sound.go:107-109"""
        return SignedInt

    def GetFloatAtIdx(self, idx):
        """sound.go:130-141"""
        return capi.load().aud_pcm_to_float(int(self.Data[idx]), int(self.SourceBitDepth))

    def WriteWave(self, fn):
        """sound.go:55-76: the buffer back to a PCM WAV file at its own rate, depth and channel count"""
        import struct
        depth, ch = int(self.SourceBitDepth), max(int(self._channels), 1)
        if depth == 8:
            body = (self.Data + 128).astype(np.uint8).tobytes()
        elif depth == 16:
            body = self.Data.astype("<i2").tobytes()
        elif depth == 24:
            v = self.Data.astype(np.int64) & 0xFFFFFF
            body = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes()
        elif depth == 32:
            body = self.Data.astype("<i4").tobytes()
        else:
            print("Encoding failed on write: unsupported bit depth %d" % depth)
            return "unsupported bit depth"
        hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(body), b"WAVE", b"fmt ", 16, 1, ch, self._rate,
                          self._rate * ch * depth // 8, ch * depth // 8, depth, b"data", len(body))
        with open(fn, "wb") as f:
            f.write(hdr + body + (b"\0" if len(body) & 1 else b""))
        return None

    def NumFrames(self):
        return len(self.Data) // max(self._channels, 1)

    def SoundToTensor(self):
        """sound.go:116-127: the first NumFrames() entries of the INTERLEAVED buffer, normalised by
        GetFloatAtIdx -- for stereo that is L,R,L,R... of the first half of the clip (SURVEY a-0 quirk);
        mono is what the hot path supports."""
        lib = capi.load()
        n = self.NumFrames()
        scale = {8: 0x7F, 16: 0x7FFF, 24: 0x7FFFFF, 32: 0x7FFFFFFF}.get(self.SourceBitDepth)
        if scale is None:
            return np.zeros(n)
        out = self.Data[:n].astype(np.float64) / float(scale)
        assert n == 0 or out[0] == lib.aud_pcm_to_float(int(self.Data[0]), self.SourceBitDepth)
        return out


class Params:
    """sound.Params, sound/sndenv.go:24-61"""

    def __init__(self):
        self.WinMs = 0.0
        self.StepMs = 0.0
        self.SegmentMs = 0.0
        self.StrideMs = 0.0
        self.BorderSteps = 0
        self.Channel = 0
        self.WinSamples = 0
        self.StepSamples = 0
        self.SegmentSamples = 0
        self.StrideSamples = 0
        self.SegmentSteps = 0
        self.Steps = []

    def to_c(self):
        c = capi.SoundParams()
        c.win_ms, c.step_ms, c.segment_ms, c.stride_ms = self.WinMs, self.StepMs, self.SegmentMs, self.StrideMs
        c.border_steps, c.channel = self.BorderSteps, self.Channel
        return c


class SndEnv:
    """sound.SndEnv, sound/sndenv.go:73-182 (hot-path fields only)"""

    def __init__(self, device=0, compute_dtype=capi.AUD_F64):
        self.Params = Params()
        self.Sound = Wave()
        self.SampleRate = 0                 # se.Sound.SampleRate()
        self.Channels = 1                   # se.Sound.Channels(); mono streams only (SURVEY a-0)
        self.Signal = np.zeros(0, np.float64)
        self.SegCnt = 0
        self.DFT = dft.Params()
        self.Mel = mel.Params()
        self.MelFilters = None
        self.PowerSegment = None
        self.LogPowerSegment = None
        self.MelFBankSegment = None
        self.Window = None
        self.Power = None
        self.LogPower = None
        self.MelFBank = None
        self.Energy = None
        self.MFCCSegment = None
        self.MFCCDeltas = None
        self.MFCCDeltaDeltas = None
        self.GaborSpecs = []
        self.GaborFilters = agabor.FilterSet()
        self.GborOutPoolsX = 0
        self.GborOutPoolsY = 0
        self.GborOutUnitsX = 0
        self.GborOutUnitsY = 0
        self.GborOutput = None
        self.GborKwta = None                # post-kwta output (sndenv.go:163)
        self.Inhibs = None                  # pool-level FFFB state KWTAPool carries between calls (:166)
        self.ExtGi = None                   # stays all zeros: NeighInhib is not built (:169-172)
        self.Kwta = kwta.KWTA()             # zero value (Kwta.On false) until Defaults(), as in Go
        self.KwtaPool = False
        self.ByTime = False
        self._device = device
        self._compute_dtype = compute_dtype
        self._plan = None
        # ProcessSegment runs once per segment on the SAME Signal, and the reference reads the LIVE tensor at every step
        # (sndenv.go:455-478): the device keeps a copy of it between calls that is validated EXACTLY on every call.
        #   ResidentSignal = None (default): for a Signal of up to AUD_RESIDENT_AUTO_BYTES every call compares what ITS frames read
        #       (4 KB blocks) byte for byte with the host shadow of the device copy and uploads what differs
        #       (aud_melspec_batch_live / aud_melspec_mfcc_batch_live) -- any in-place edit is seen by the call that reads it;
        #       a larger Signal is copied per call, as if there were no residency.
        #   ResidentSignal = True, or an explicit SignalToDevice(): the caller opts in to a SNAPSHOT it keeps current itself --
        #       re-taken when Signal is another array, length or type; SignalChanged() after an in-place edit.
        #   ResidentSignal = False: copy per call.
        self.ResidentSignal = None
        self._dev_sig = self._dev_sig_key = None
        self._snapshot = False      # the resident copy is an opted-in snapshot (validated by identity only)

    def ToTensor(self):
        """sound/sndenv.go:297-300: Signal <- Sound.SoundToTensor()"""
        self.Signal = self.Sound.SoundToTensor()
        self.SampleRate, self.Channels = self.Sound.SampleRate(), self.Sound.Channels()
        self.SignalChanged()
        return True

    def ParamDefaults(self):
        """sound/sndenv.go:64-71"""
        c = capi.SoundParams()
        capi.load().aud_sound_params_defaults(c)
        p = self.Params
        p.WinMs, p.StepMs, p.SegmentMs, p.StrideMs = c.win_ms, c.step_ms, c.segment_ms, c.stride_ms
        p.Channel, p.BorderSteps = c.channel, c.border_steps

    def Defaults(self):
        """sound/sndenv.go:185-192"""
        self.ParamDefaults()
        self.Mel.Defaults()
        self.Kwta.Defaults()
        self.KwtaPool = True
        self.ByTime = False

    def Init(self):
        """sound/sndenv.go:195-267.  Returns None, or an error string like the Go `error`."""
        lib = capi.load()
        c = self.Params.to_c()
        if lib.aud_sound_params_derive(c, int(self.SampleRate)) != capi.AUD_OK:
            print("sample rate <= 0")
            return "sample rate <= 0"
        p = self.Params
        p.WinSamples, p.StepSamples, p.SegmentSamples = c.win_samples, c.step_samples, c.segment_samples
        p.SegmentSteps, p.StrideSamples = c.segment_steps, c.stride_samples

        specs = agabor.Active(self.GaborSpecs)
        if specs:
            agabor.ToTensor(specs, self.GaborFilters)
        if self.GborOutPoolsX == 0 and self.GborOutPoolsY == 0:
            self.GborOutput = np.zeros((self.GborOutUnitsY, self.GborOutUnitsX), np.float32)
        elif self.GborOutPoolsX > 0 and self.GborOutPoolsY > 0:
            self.GborOutput = np.zeros((self.GborOutPoolsY, self.GborOutPoolsX, self.GborOutUnitsY,
                                        self.GborOutUnitsX), np.float32)
        else:
            print("GborOutPoolsX & GborOutPoolsY must both be == 0 or > 0 (i.e. 2D or 4D)")
            return None
        self.ExtGi = np.zeros(self.GborOutput.shape, np.float32)
        self.GborKwta = np.zeros(self.GborOutput.shape, np.float32)

        H = p.WinSamples // 2 + 1
        self.DFT.Defaults()
        self.MelFilters = self.Mel.InitFilters(p.WinSamples, self.SampleRate)
        self.PowerSegment = np.zeros((H, p.SegmentSteps))
        self.LogPowerSegment = np.zeros((H, p.SegmentSteps))
        p.Steps = [p.StepSamples * (i - p.BorderSteps) for i in range(p.SegmentSteps)]
        self.MelFBankSegment = np.zeros((self.Mel.FBank.NFilters, p.SegmentSteps))
        self.Energy = np.zeros(p.SegmentSteps)
        if self.Mel.MFCC:
            self.MFCCSegment = np.zeros((self.Mel.NCoefs, p.SegmentSteps))
            self.MFCCDeltas = np.zeros((self.Mel.NCoefs, p.SegmentSteps))
            self.MFCCDeltaDeltas = np.zeros((self.Mel.NCoefs, p.SegmentSteps))
        self.SegCnt = lib.aud_seg_cnt(len(self.Signal), p.SegmentSamples, p.StrideSamples,
                                      self.Channels)
        self._plan_key = None
        self._drop_resident()              # (a resident copy belongs to the Signal it was taken from)
        self._ensure_plan()
        return None

    def _ensure_plan(self):
        """The device plan bakes the DFT and mel-bank parameters in, while the reference reads se.DFT / se.Mel.FBank at CALL
        time (Init resets se.DFT, sndenv.go:230, so PrevSmooth etc. can only be set after it): the plan is keyed on them and
        rebuilt lazily when a call finds them changed."""
        d, fb = self.DFT, self.Mel.FBank
        key = (bool(d.CompLogPow), d.LogMin, d.LogOffSet, d.PrevSmooth, d.CurSmooth, fb.LogOff, fb.LogMin, bool(fb.Renorm),
               fb.RenormMin, fb.RenormMax, fb.RenormScale, bool(self.Mel.MFCC), self.Mel.NCoefs, self._compute_dtype)
        if self._plan is None or key != self._plan_key:
            self._make_plan()
            self._plan_key = key
        return self._plan

    def _make_plan(self):
        if self._plan is not None:
            self._plan.close()
        p = self.Params
        has_g = len(self.GaborFilters.Filters) > 0
        self._plan = runtime.Plan(runtime.get_ctx(self._device), p.WinSamples, p.StepSamples,
                                  p.SegmentSteps, p.BorderSteps, self.DFT.to_c(),
                                  self.Mel.FBank.to_c(), self.Mel.BinPts, self.MelFilters,
                                  self.GaborFilters.to_c() if has_g else None,
                                  self.GaborFilters.Filters if has_g else None, self._compute_dtype,
                                  mfcc_coefs=self.Mel.NCoefs if self.Mel.MFCC else 0)

    def _drop_resident(self):
        if self._dev_sig is not None:
            self._dev_sig.close()
        self._dev_sig = self._dev_sig_key = None
        self._snapshot = False

    def _signal_key(self):
        """what an opted-in SNAPSHOT is valid for: the tensor's memory, length, type and strides (its contents are the
        caller's business: SignalChanged())"""
        sig = self.Signal
        return (sig.__array_interface__["data"][0], len(sig), sig.dtype.str, sig.strides)

    def SignalChanged(self):
        """New: after changing samples of self.Signal IN PLACE while a snapshot is resident (ResidentSignal = True or
        SignalToDevice()): the next ProcessSegment uploads the tensor again.  Not needed in the default mode, which compares
        the whole tensor on every call."""
        self._dev_sig_key = None

    def SignalToDevice(self):
        """New: opt in to a resident SNAPSHOT of self.Signal, taken NOW (aud_signal_upload); ProcessSegment(s) then send only
        the work items and fetch only the results, whatever the Signal's size.  The snapshot is re-taken when Signal is
        another array / length / type; an in-place edit must be announced with SignalChanged()."""
        self._drop_resident()
        self._dev_sig = runtime.Signal(runtime.get_ctx(self._device), np.ascontiguousarray(self.Signal, np.float64))
        self._dev_sig_key = self._signal_key()
        self._snapshot = True

    def _resident(self):
        """the device copy this call may read (None: copy per call)"""
        if self.ResidentSignal is False or len(self.Signal) == 0:
            return None
        if self.ResidentSignal or self._snapshot:                       # opted-in snapshot (until Init, or ResidentSignal = False)
            if self._dev_sig is None or self._dev_sig_key != self._signal_key():
                self.SignalToDevice()
            return self._dev_sig
        if len(self.Signal) * 8 > capi.AUD_RESIDENT_AUTO_BYTES:
            self._drop_resident()
            return None
        if self._dev_sig is None:
            self._dev_sig = runtime.Signal(runtime.get_ctx(self._device))
        return self._dev_sig            # (the default: the _live calls validate what they read, exactly, as part of the call)

    def _item(self, segment, add):
        start0 = segment * self.Params.StrideSamples + MSecToSamples(add, self.SampleRate)
        return (0, len(self.Signal), start0)

    def SndToWindow(self, start):
        """sound/sndenv.go:455-478.  Returns None or the Go error text; fills self.Window."""
        import ctypes as C
        self.Window = np.zeros(self.Params.WinSamples)
        rc = capi.load().aud_snd_to_window(self.Signal.ctypes.data_as(C.c_void_p), len(self.Signal), int(start),
                                           self.Params.WinSamples, self.Window.ctypes.data_as(C.c_void_p))
        return None if rc == capi.AUD_OK else "SndToWindow: end beyond signal length!!"

    def ProcessStep(self, segment, step, add=0):
        """sound/sndenv.go:438-452: one step through dft.Filter and mel.FilterDft (a GPU round trip per
        frame; ProcessSegment is the batched way).  Returns None or the error text."""
        p = self.Params
        if self.Power is None:
            H = p.WinSamples // 2 + 1
            self.Power, self.LogPower, self.MelFBank = np.zeros(H), np.zeros(H), np.zeros(self.Mel.FBank.NFilters)
        offset = p.Steps[step] + MSecToSamples(add, self.SampleRate)
        err = self.SndToWindow(segment * p.StrideSamples + offset)
        if err is None:
            plan = self._ensure_plan()
            self.DFT.Filter(step, self.Window, p.WinSamples, self.Power, self.LogPower, self.PowerSegment,
                            self.LogPowerSegment, plan)
            self.Mel.FilterDft(step, self.Power, self.MelFBankSegment, self.MelFBank, self.MelFilters, plan)
        return err

    def ProcessSegment(self, segment, add=0):
        """sound/sndenv.go:342-359 (frame loop part): fills PowerSegment, LogPowerSegment and
        MelFBankSegment for one segment."""
        self.ProcessSegments([segment], add)

    def ProcessSegments(self, segments, add=0):
        """Batch extension: all requested segments in ONE call.  Returns
        (mel [n, nf, T], power [n, H, T], log_power [n, H, T]); the SndEnv tensors hold the last.
        With Mel.MFCC (the default) the MFCC tail of sndenv.go:360-432 runs too and fills
        MFCCSegment / MFCCDeltas / MFCCDeltaDeltas / Energy."""
        its = [self._item(s, add) for s in segments]
        items = runtime.make_items([i[0] for i in its], [i[1] for i in its], [i[2] for i in its])
        self._ensure_plan()
        res = self._resident()
        live = res is not None and not self._snapshot      # default residency: aud_*_live compares what the call reads
        if self.Mel.MFCC and self.DFT.CompLogPow:
            o = (self._plan.melspec_mfcc_live(res, self.Signal, items, deltas=bool(self.Mel.Deltas)) if live else
                 self._plan.melspec_mfcc_sig(res, items, deltas=bool(self.Mel.Deltas)) if res is not None else
                 self._plan.melspec_mfcc_host(self.Signal, items, deltas=bool(self.Mel.Deltas)))
            m, pw, lp = o["mel"], o["power"], o["log_power"]
            self.MFCCSegment, self.Energy = o["mfcc"][-1], o["energy"][-1]
            if self.Mel.Deltas:
                self.MFCCDeltas, self.MFCCDeltaDeltas = o["deltas"][-1], o["delta_deltas"][-1]
            self._last_mfcc = o
        else:
            m, pw, lp = (self._plan.melspec_live(res, self.Signal, items, True, bool(self.DFT.CompLogPow)) if live else
                         self._plan.melspec_sig(res, items, True, bool(self.DFT.CompLogPow)) if res is not None else
                         self._plan.melspec_host(self.Signal, items, True, bool(self.DFT.CompLogPow)))
        self.MelFBankSegment = m[-1]
        self.PowerSegment = pw[-1]
        if lp is not None:
            self.LogPowerSegment = lp[-1]
        return m, pw, lp

    def ApplyNeighInhib(self):
        """sound/sndenv.go:303-311.  NeighInhib.Inhib4 (emer/vision) is not built: there is no NeighInhib field to
        turn on here, so this is the `else` branch -- ExtGi is cleared."""
        self.ExtGi[...] = 0

    def Name(self):
        return getattr(self, "Nm", "")

    def Desc(self):
        return getattr(self, "Dsc", "")

    def ApplyKwta(self):
        """sound/sndenv.go:313-323: GborKwta <- GborOutput, then KWTAPool / KWTALayer when Kwta.On"""
        self.GborKwta[...] = self.GborOutput
        if self.Kwta.On:
            if self.KwtaPool:
                if self.GborOutput.ndim != 4:
                    # the reference panics here (KWTAPool reads Dim(2), Dim(3) of a 2-D tensor)
                    raise capi.AuditoryError(capi.AUD_EINVAL, "KwtaPool needs the 4-D gabor output")
                n_pools = self.GborOutput.shape[0] * self.GborOutput.shape[1]
                if self.Inhibs is None or self.Inhibs.shape[0] != n_pools:
                    self.Inhibs = np.zeros((n_pools, 2), np.float32)
                self.Kwta.KWTAPool(self.GborOutput, self.GborKwta, self.Inhibs, self.ExtGi, device=self._device)
            else:
                self.Kwta.KWTALayer(self.GborOutput, self.GborKwta, self.ExtGi, device=self._device)

    def ApplyGabor(self):
        """sound/sndenv.go:481-497.  NeighInhib is not built: ExtGi stays zero, which is the reference's
        state whenever NeighInhib.On is false (its zero value; SndEnv.Defaults never turns it on)."""
        agabor.Convolve(self.MelFBankSegment, self.GaborFilters, self.GborOutput, self.ByTime,
                        plan=self._ensure_plan())
        self.ExtGi[...] = 0
        if self.Kwta.On:
            self.ApplyKwta()
            return self.GborKwta
        return self.GborOutput

    def AdjustForSilence(self, add, existing):
        """sound/sndenv.go:274-294: trims or prepends silence at the start of Signal; returns the offset (ms)"""
        import ctypes as C
        delta = C.c_int(0)
        off = capi.load().aud_adjust_for_silence(float(add), float(existing), int(self.SampleRate), C.byref(delta))
        if off < 0:
            print("sample rate <= 0")
            return -1
        if delta.value < 0:
            self.Signal = self.Signal[-delta.value:]
        elif delta.value > 0:
            self.Signal = np.concatenate([np.zeros(delta.value), self.Signal])
        self.SignalChanged()
        return off

    def Tail(self, signal):
        """sound/sndenv.go:503-507"""
        return capi.load().aud_tail(len(signal), self.Params.SegmentSamples, self.Params.StrideSamples)

    def Pad(self, signal, value=0.0):
        """sound/sndenv.go:510-519"""
        n = capi.load().aud_pad_len(len(signal), self.Params.SegmentSamples,
                                    self.Params.StrideSamples, self.Params.StepSamples)
        return np.concatenate([np.asarray(signal, np.float64), np.full(n, value)])
