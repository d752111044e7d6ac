"""Builds auditory_amd/libauditory_hip.so for gfx950 with hipcc (in-tree, so the .so travels
to the GPU box with the repo snapshot)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libauditory_hip.so")
SOURCES = ["host_setup.cpp", "capi.hip", "melspec_generic.hip", "melspec_r16.hip", "melspec_r25.hip", "melspec_r1024.hip", "smooth_mel.hip", "mfcc.hip", "gabor.hip", "kwta.hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, "kernels.h"), os.path.join(INCLUDE, "auditory_hip.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".inc"))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> libauditory_hip.so.  Returns the path."""
    if not force and not stale():
        return LIB
    # -fno-slp-vectorize: hipcc otherwise packs neighbouring f32 adds/muls of the butterflies into
    # v_pk_* instructions, which on gfx950 issue no faster than two scalar ops and cost ~90 extra
    # v_mov per wave to pair registers up (and 20 more VGPRs) -- see DESIGN.md 4.1
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared",
           "-Xarch_host", "-ffp-contract=off", "-I" + INCLUDE, "-I" + CSRC, "-o", LIB] + sources() + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
