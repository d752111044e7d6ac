"""TEST INFRASTRUCTURE ONLY: compiles the product's HIP sources, unchanged, with g++ against the
host stand-in for the HIP runtime (tests/emul/hip/hip_runtime.h) into tests/emul/libauditory_emul*.so.
Variants: plain (-O2), asan (AddressSanitizer + UBSan), tsan (ThreadSanitizer)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from auditory_amd import build as product_build  # noqa: E402  (source list only)

# -mfma: the kernels spell their fused multiply-adds out (device_common.h mad) and switch the compiler's own contraction
# off, so with the instruction available the emulator computes what the GPU computes up to the hardware's log / rcp
FLAGS = {
    "plain": ["-O2", "-mfma"],
    "asan": ["-O1", "-g", "-mfma", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
             "-fno-sanitize-recover=undefined"],
    "tsan": ["-O1", "-g", "-mfma", "-fsanitize=thread"],
    # contraction left to the compiler where the sources do not fix it (gabor, mfcc, k-WTA)
    "fma": ["-O2", "-mfma", "-ffp-contract=fast"],
}


def lib_path(variant="plain"):
    return os.path.join(HERE, "libauditory_emul%s.so" % ("" if variant == "plain" else "_" + variant))


def build(variant="plain", force=False):
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:  # parallel test workers build once, not at once
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(variant, force)


def _build(variant, force):
    out = lib_path(variant)
    srcs = product_build.sources() + [os.path.join(HERE, "emul_runtime.cpp")]
    deps = srcs + [os.path.join(HERE, "hip", "hip_runtime.h"),
                   os.path.join(product_build.CSRC, "kernels.h"),
                   os.path.join(product_build.INCLUDE, "auditory_hip.h")]
    deps += [os.path.join(product_build.CSRC, f) for f in os.listdir(product_build.CSRC)
             if f.endswith((".h", ".hpp", ".inc"))]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    common = ["g++", "-std=c++17", "-fPIC", "-pthread"] + ([] if variant == "fma" else ["-ffp-contract=off"]) + [
              "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-function", "-Wno-attributes"] + FLAGS[variant] + [
              "-I" + HERE, "-I" + product_build.INCLUDE, "-I" + product_build.CSRC]
    # one compile job per source, a few at a time (the sanitizer variants take a minute otherwise), then one link
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory(prefix="aud_emul_") as tmpdir:
        objs = [os.path.join(tmpdir, "%d_%s.o" % (i, os.path.basename(src))) for i, src in enumerate(srcs)]
        with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as pool:
            list(pool.map(lambda so: subprocess.check_call(common + ["-c", "-x", "c++", so[0], "-o", so[1]]),
                          zip(srcs, objs)))
        tmp = out + ".tmp%d" % os.getpid()
        subprocess.check_call(common + ["-shared", "-o", tmp] + objs + ["-ldl"])
    os.replace(tmp, out)  # readers never see a half-written library
    return out


if __name__ == "__main__":
    for v in sys.argv[1:] or ["plain"]:
        print(build(v, force=True))
