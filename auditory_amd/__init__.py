"""auditory_amd -- MI355X (gfx950) implementation of emer/auditory's
signal -> framed FFT -> power -> mel -> gabor hot path behind the reference's own API names.

The arithmetic lives in libauditory_hip.so (hand-written HIP, C ABI in include/auditory_hip.h);
these modules mirror the reference's Go packages on the host side.
"""
from . import capi  # noqa: F401

__all__ = ["capi", "runtime", "dft", "mel", "agabor", "sound", "batch", "synth"]
