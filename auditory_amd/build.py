"""Builds auditory_amd/libauditory_hip.so for gfx950 with hipcc (in-tree, so the .so travels
to the GPU box with the repo snapshot)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libauditory_hip.so")
SOURCES = ["host_setup.cpp", "capi.hip", "capi_host.hip", "capi_comm.hip", "wave_tables.hip", "melspec_generic.hip", "melspec_chirp.hip", "melspec_direct.hip", "melspec_wave.hip", "melspec_w16.hip", "melspec_w20.hip", "melspec_w64.hip", "smooth_mel.hip", "mfcc.hip", "gabor.hip", "kwta.hip"]


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


def _flags():
    # -fno-slp-vectorize: hipcc otherwise packs neighbouring f32 adds/muls of the butterflies into
    # v_pk_* instructions, which on gfx950 issue no faster than two scalar ops (measured: profiles/
    # r02_valu_issue_rates.txt) and cost ~90 extra v_mov per wave to pair registers up -- see DESIGN.md 4.1
    # -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in SGPRs with the wave (wave_common.h wave_kernel_t)
    return ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=" + os.environ.get("AUD_KERNARG_PRELOAD", "16"),
            "-Xarch_host", "-ffp-contract=off", "-I" + INCLUDE, "-I" + CSRC]


def build(force=False, verbose=False, stamps=False, tag=None, defines=()):
    """hipcc --offload-arch=gfx950 -> libauditory_hip.so.  One compile job per source (only the stale ones, a few
    at a time), then one link.  Returns the path.  stamps=True builds the diagnostic variant
    libauditory_hip_stamps.so (-DAUD_STAMPS: s_memtime stamps in the wave kernels, tools/stamp_profile.py); tag + defines
    an experimental build libauditory_hip_<tag>.so for tools/ab_bench.py (never shipped: *_<tag>.so is git-ignored)."""
    if stamps:
        return _build(True, verbose, os.path.join(HERE, "obj_stamps"), LIB.replace(".so", "_stamps.so"), ["-DAUD_STAMPS=1"])
    if tag:
        return _build(True, verbose, os.path.join(HERE, "obj_" + tag), LIB.replace(".so", "_%s.so" % tag), list(defines))
    if not force and not stale():
        return LIB
    return _build(force, verbose, os.path.join(HERE, "obj"), LIB, [])


def _build(force, verbose, objdir, LIB, extra):
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(INCLUDE, "auditory_hip.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC)
                                                           if f.endswith((".h", ".hpp", ".inc"))]
    newest_header = max(os.path.getmtime(h) for h in headers)
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            jobs.append([_hipcc()] + _flags() + extra + ["-x", "hip", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
        list(pool.map(run, jobs))
    run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


UBENCH_SRC = os.path.join(os.path.dirname(HERE), "tools", "ubench", "stream_read.hip")
UBENCH_LIB = os.path.join(os.path.dirname(HERE), "tools", "ubench", "libstream_read.so")


def build_stream_read(force=False):
    """tools/ubench/libstream_read.so: the plain float32 read kernel bench.py times for `roofline.measured_read_GBps` (SURVEY 8d's
    measured HBM-read roof).  Measurement only: the product library neither contains nor loads it."""
    if force or not os.path.exists(UBENCH_LIB) or os.path.getmtime(UBENCH_LIB) < os.path.getmtime(UBENCH_SRC):
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", UBENCH_LIB, UBENCH_SRC])
    return UBENCH_LIB


if __name__ == "__main__":
    import sys
    argv = sys.argv[1:]
    if "--tag" in argv:  # python -m auditory_amd.build --tag exp1 -DAUD_EXP_FOO=1
        t = argv[argv.index("--tag") + 1]
        # -D... defines and, for compiler experiments, any other flag after a literal "--" (e.g. -- -mllvm -amdgpu-foo=1)
        extra = argv[argv.index("--") + 1:] if "--" in argv else []
        print(build(tag=t, defines=[a for a in argv if a.startswith("-D")] + extra, verbose=True))
    else:
        print(build(force=True, verbose=True))
