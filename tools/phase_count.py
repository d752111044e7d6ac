"""Static instruction counts per phase of a wave kernel (no GPU): compiles the kernel's file with -DAUD_PHASE_MARKERS
(kernels.h puts '; AUD_PHASE n' comments at the kernels' phase boundaries) and counts the instructions between them.
Counts are static: a loop body counts once, every sample route is listed.
usage: python tools/phase_count.py [kernel-name-substring, default k_melspec_w20IdLi0ELi4ELi4E] [--ops]
(the source file is melspec_w16 / w20 / w64.hip by the name's w16 / w20 / w64)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cat(op):
    if re.match(r"v_(fma|mul|add|sub|fmac|ldexp|max|min)_f64", op): return 'f64'
    if re.match(r"v_cvt_.*f64|v_frexp.*f64", op): return 'cvt64'
    if re.match(r"v_(fma|mul|add|sub|fmac|mac|max|min|max3)_f32", op): return 'f32'
    if op.startswith('v_'): return 'v_other'
    if op.startswith('ds_'): return 'ds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    return 'other'


def main():
    key = next((a for a in sys.argv[1:] if not a.startswith("--")), "k_melspec_w20IdLi0ELi4ELi4E")
    src = "melspec_%s.hip" % next(w for w in ("w16", "w20", "w64") if w in key)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "wave.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17",
                               "-DAUD_PHASE_MARKERS", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "auditory_amd", "csrc"), "-x", "hip", "--cuda-device-only", "-S",
                               os.path.join(ROOT, "auditory_amd", "csrc", src), "-o", out],
                              stderr=subprocess.DEVNULL)
        s = open(out).read().split('\n')
    start = next(i for i, l in enumerate(s) if re.match(r"^_Z\w*%s\w*:" % key, l))
    end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
    phase, seq, cnt = 'start', 0, collections.OrderedDict()
    for l in s[start + 1:end]:
        t = l.strip()
        m = re.match(r"; AUD_PHASE (\d+)", t)
        if m:
            seq += 1
            phase = "%02d after stamp %s" % (seq, m.group(1))
            continue
        if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
            continue
        op = t.split()[0]
        d = cnt.setdefault(phase, collections.Counter())
        d[cat(op)] += 1
        d['ops:' + op] += 1
    for p, d in cnt.items():
        print(p, {k: v for k, v in d.items() if not k.startswith('ops:')})
        if "--ops" in sys.argv:
            print('    ', sorted([(k[4:], v) for k, v in d.items() if k.startswith('ops:')], key=lambda x: -x[1])[:16])


if __name__ == "__main__":
    main()
