"""The cgo mirror under go/ has never met a Go compiler (none in the image): check statically what can be checked
against include/auditory_hip.h (tests/go_lint.py says what that is and is not)."""
import collections
import os
import re

import pytest

import go_lint as GL

HDR = GL.Header()
FILES = list(GL.go_files())
REL = [os.path.relpath(f, GL.ROOT) for f in FILES]


def _code(path):
    return GL.blank_go(open(path).read())


def test_header_parsed():
    # the parser sees the whole boundary: every exported symbol the library test knows about, and the structs the mirror fills
    assert len(HDR.funcs) >= 60 and HDR.funcs["aud_version"] == 0 and HDR.funcs["aud_init"] == 2
    assert {"aud_plan_desc", "aud_item", "aud_mel_fbank", "aud_dft_params", "aud_gabor_spec", "aud_gabor_set",
            "aud_kwta_params", "aud_sound_params"} <= set(HDR.structs)
    assert {"AUD_OK", "AUD_F64", "AUD_F32", "AUD_I16", "AUD_FAST_F32", "AUD_ESHORT"} <= HDR.consts
    assert list(HDR.structs["aud_item"]) == ["sig_off", "sig_len", "start0", "sig_stride", "reserved"]


@pytest.mark.parametrize("path", FILES, ids=REL)
def test_brackets_balance(path):
    GL.check_balance(_code(path), os.path.relpath(path, GL.ROOT))


@pytest.mark.parametrize("path", FILES, ids=REL)
def test_c_calls_match_the_header(path):
    code = _code(path)
    for name, args, line in GL.c_calls(code):
        assert name in HDR.funcs, "%s:%d: C.%s is not declared in auditory_hip.h" % (path, line, name)
        assert len(args) == HDR.funcs[name], "%s:%d: C.%s called with %d arguments, the header has %d" % (
            path, line, name, len(args), HDR.funcs[name])
    for m in re.finditer(r"\bC\.(AUD_\w+)", code):
        assert m.group(1) in HDR.consts, "%s: C.%s is not in the header" % (path, m.group(1))
    for m in re.finditer(r"\bC\.(aud_\w+)\b(?!\s*\()", code):
        assert m.group(1) in HDR.structs or m.group(1) in HDR.opaque, "%s: type C.%s is not in the header" % (path, m.group(1))


@pytest.mark.parametrize("path", FILES, ids=REL)
def test_struct_fields_exist(path):
    code = _code(path)
    for typ, keys, line in GL.c_literals(code):
        for k in keys:
            assert k in HDR.structs[typ], "%s:%d: %s has no field %s" % (path, line, typ, k)
    for text in GL.func_texts(code):
        for var, typ in GL.typed_vars(text).items():
            if typ not in HDR.structs:
                continue
            for m in re.finditer(r"(?<![\w.])%s\.(\w+)" % re.escape(var), text):
                assert m.group(1) in HDR.structs[typ], "%s: %s.%s: %s has no such field" % (path, var, m.group(1), typ)


def test_nested_struct_selectors():
    # desc.mel.n_filters and the like: the second selector belongs to the nested struct
    seen = 0
    for path in FILES:
        for text in GL.func_texts(_code(path)):
            for var, typ in GL.typed_vars(text).items():
                for m in re.finditer(r"(?<![\w.])%s\.([a-z]\w*)\.([a-z]\w*)" % re.escape(var), text):
                    ftype = HDR.structs.get(typ, {}).get(m.group(1))
                    if ftype in HDR.structs:
                        seen += 1
                        assert m.group(2) in HDR.structs[ftype], "%s: %s.%s.%s" % (path, var, m.group(1), m.group(2))
    assert seen >= 1      # desc.mel.n_filters in NewPlan


def test_no_duplicate_declarations():
    by_pkg = collections.defaultdict(list)
    for path in FILES:
        for kind, recv, name, line in GL.declarations(_code(path)):
            if name in ("init", "main") and not recv and os.path.basename(path).endswith("_test.go"):
                continue
            by_pkg[(os.path.dirname(path), recv, name)].append("%s:%d" % (os.path.relpath(path, GL.ROOT), line))
    dup = {k[1:]: v for k, v in by_pkg.items() if len(v) > 1 and k[2] != "init"}
    assert not dup, "declared twice in one package: %s" % dup


@pytest.mark.parametrize("path", FILES, ids=REL)
def test_imports_used_and_present(path):
    src = open(path).read()
    code = _code(path)
    body = code[code.find("\n", max(code.rfind("import"), 0)):] if "import" in code else code
    imps = GL.imports(src)
    for local, full in imps.items():
        if local == "_":
            continue
        assert re.search(r"\b%s\." % re.escape(local), body), "%s: import %s is not used" % (path, full)
    for pkg in GL.STD_PKGS:
        local = pkg.rsplit("/", 1)[-1]
        if local in ("io", "time", "sort", "flag", "reflect", "runtime", "testing", "log", "os", "strings", "strconv", "bufio",
                     "binary", "filepath", "fmt", "math", "errors", "unsafe", "sync"):
            used = re.search(r"(?<![\w.])%s\.[A-Z]" % re.escape(local), body)
            if used and local not in imps:
                # a local variable of that name (e.g. a struct field access) would be lower-case after the dot
                raise AssertionError("%s: %s.%s... used but %s is not imported" % (path, local, used.group(0)[-1], pkg))
    if re.search(r"\bC\.", code):
        assert re.search(r'^import "C"', src, flags=re.M), "%s uses C. without import \"C\"" % path


def _funcs(rel):
    return GL.go_funcs(_code(os.path.join(GL.ROOT, rel)))


def test_reference_signatures_present():
    """the exported Go signatures a drop-in keeps -- receiver, name and parameter TYPES as in the reference
    (dft/dft.go:33-62, mel/mel.go:69-192, agabor/gabor.go:73-329, sound/sndenv.go:64-527); the lists below were read off
    those lines"""
    F64, F32 = "*etensor.Float64", "*etensor.Float32"
    want = {
        "go/dft/dft.go": {("Params", "Defaults"): [], ("Params", "Filter"): ["int", F64, "int", F64, F64, F64, F64],
                          ("Params", "FftReal"): ["[]complex128", F64],
                          ("Params", "Power"): ["int", "int", "[]complex128", F64, F64, F64, F64]},
        "go/mel/mel.go": {("Params", "Defaults"): [], ("Params", "InitFilters"): ["int", "int", F64],
                          ("Params", "FilterDft"): ["int", F64, F64, F64, F64], ("", "FreqToMel"): ["float64"],
                          ("", "MelToFreq"): ["float64"], ("", "FreqToBin"): ["float64", "float64", "float64"],
                          ("FilterBank", "Defaults"): [], ("Params", "CepstrumDct"): ["int", F64, F64, F64]},
        "go/agabor/gabor.go": {("Filter", "Defaults"): ["int"], ("", "ToTensor"): ["[]Filter", "*FilterSet"],
                               ("", "Convolve"): [F64, "FilterSet", F32, "bool"], ("", "Active"): ["[]Filter"]},
        "go/sound/sndenv.go": {("SndEnv", "ParamDefaults"): [], ("SndEnv", "Defaults"): [], ("SndEnv", "Init"): [],
                               ("SndEnv", "AdjustForSilence"): ["float64", "float64"], ("SndEnv", "ToTensor"): [],
                               ("SndEnv", "ApplyNeighInhib"): [], ("SndEnv", "ApplyKwta"): [],
                               ("SndEnv", "ProcessSegment"): ["int", "int"], ("SndEnv", "ProcessStep"): ["int", "int", "int"],
                               ("SndEnv", "SndToWindow"): ["int"], ("SndEnv", "ApplyGabor"): [], ("SndEnv", "Name"): [],
                               ("SndEnv", "Desc"): [], ("SndEnv", "Tail"): ["[]float64"], ("SndEnv", "Pad"): ["[]float64", "float64"],
                               ("", "MSecToSamples"): ["float64", "int"], ("", "SamplesToMSec"): ["int", "int"]},
    }
    for rel, sigs in want.items():
        have = _funcs(rel)
        for key, types in sigs.items():
            assert key in have, "%s: no %s" % (rel, (key,))
            assert have[key] == types, "%s: %s takes %s, the reference %s" % (rel, key, have[key], types)


def test_refdump_writes_what_the_golden_hook_reads():
    """go/cmd/refdump (the program that pins the oracle where Go exists) and tests/golden_cases.py (the reader) must agree on
    every file kind, and refdump may only touch what the REFERENCE's SndEnv exports (sound/sndenv.go:73-182, :185-497; the
    list below was read off those lines, and is checked against the file itself where /root/reference is present)."""
    import golden_cases as GC
    src = open(os.path.join(GL.ROOT, "go", "cmd", "refdump", "main.go")).read()
    code = GL.blank_go(src)
    for kind, ext in GC.REF_KINDS:
        stem, suffix = ext.rsplit(".", 1)
        assert ('_%s.%s"' % (stem, suffix)) in src or ('"%s"' % stem) in src, "refdump never writes *_%s" % ext
    assert "_kwta_params.txt" in src and "%+v" in src
    ref_fields = {"Sound", "Params", "Signal", "Mel", "DFT", "Kwta", "KwtaPool", "GaborFilters", "GaborSpecs", "GborOutPoolsX",
                  "GborOutPoolsY", "GborOutUnitsX", "GborOutUnitsY", "GborOutput", "GborKwta", "MelFBankSegment",
                  "LogPowerSegment", "PowerSegment", "Energy", "MFCCSegment", "MFCCDeltas", "MFCCDeltaDeltas", "Inhibs", "ByTime"}
    ref_methods = {"Defaults", "ToTensor", "Init", "ProcessSegment", "ApplyGabor"}
    used = set(re.findall(r"\bse\.([A-Z]\w*)", code))
    assert used <= ref_fields | ref_methods, "refdump uses SndEnv members the reference does not export: %s" % sorted(
        used - ref_fields - ref_methods)
    assert {"Energy", "MFCCDeltas", "MFCCDeltaDeltas", "GborKwta", "Kwta", "KwtaPool"} <= used   # f-1 and f-4 are dumped
    assert "se.Kwta.On = false" in src and "se.KwtaPool = pool" in src                          # both passes exist
    ref = "/root/reference/sound/sndenv.go"
    if os.path.exists(ref):
        text = open(ref).read()
        for name in used & ref_fields:
            assert re.search(r"^\t%s\s" % name, text, flags=re.M), "the reference's SndEnv has no field %s" % name
        for name in used & ref_methods:
            assert re.search(r"^func \(se \*SndEnv\) %s\(" % name, text, flags=re.M), "the reference's SndEnv has no method %s" % name


def test_calls_into_the_binding_package():
    """every auditoryhip.X the drop-in packages name is declared there, function calls pass as many arguments as the
    declaration has parameters, and so do calls of the binding's methods whose names no other package declares"""
    bind = os.path.join(GL.ROOT, "go", "auditoryhip", "auditoryhip.go")
    code = _code(bind)
    funcs = GL.go_funcs(code)
    declared = {n for (_, n) in funcs if True} | {n for k, _, n, _ in GL.declarations(code) if k == "type"}
    declared |= set(re.findall(r"^(?:var|const)\s+(\w+)", code, flags=re.M)) | set(re.findall(r"^\t(\w+)\s+(?:\*|error|sync)", code, flags=re.M))
    plain = {n: t for (r, n), t in funcs.items() if not r}
    others = collections.Counter()
    for path in FILES:
        if path != bind:
            for (r, n) in GL.go_funcs(_code(path)):
                if r:
                    others[n] += 1
    methods = collections.defaultdict(set)
    for (r, n), t in funcs.items():
        if r:
            methods[n].add(len(t))
    for path in FILES:
        if path == bind:
            continue
        c = _code(path)
        for m in re.finditer(r"\bauditoryhip\.(\w+)", c):
            assert m.group(1) in declared, "%s: auditoryhip.%s is not declared" % (path, m.group(1))
        for name, n, line in GL.calls_of(c, r"\bauditoryhip\.(\w+)\s*\("):
            if name in plain:
                assert n == len(plain[name]), "%s:%d: auditoryhip.%s called with %d arguments, declared with %d" % (
                    path, line, name, n, len(plain[name]))
        for name, n, line in GL.calls_of(c, r"(?<!auditoryhip)\.(\w+)\s*\("):
            if name in methods and not others[name] and len(methods[name]) == 1 and name not in ("Close", "Len"):
                assert n in methods[name], "%s:%d: .%s called with %d arguments, the binding's method has %s" % (
                    path, line, name, n, sorted(methods[name]))


def test_lint_catches_planted_errors():
    """the checker itself: each class of mistake, planted in a snippet, is found"""
    snippet = GL.blank_go('''package x
// C.aud_nope(1) in a comment is not code
func a(p *C.aud_plan_desc) C.aud_item {
	s := "C.aud_nope(" + `{`
	rc := C.aud_init(C.int(0))
	var it C.aud_item
	it.sig_of = 1
	p.mel.n_filter = 2
	return C.aud_item{sig_off: 1, sig_length: 2}
}
func a() {}
''')
    GL.check_balance(snippet, "snippet")
    assert [(n, len(a)) for n, a, _ in GL.c_calls(snippet)] == [("aud_init", 1)] and HDR.funcs["aud_init"] == 2
    assert [(t, k) for t, k, _ in GL.c_literals(snippet)] == [("aud_item", ["sig_off", "sig_length"])]
    text = next(GL.func_texts(snippet))
    assert GL.typed_vars(text) == {"p": "aud_plan_desc", "it": "aud_item"}
    assert "sig_of" not in HDR.structs["aud_item"] and "n_filter" not in HDR.structs["aud_mel_fbank"]
    assert [d[2] for d in GL.declarations(snippet)] == ["a", "a"]
    with pytest.raises(AssertionError):
        GL.check_balance("func a() { x := b[1) }", "snippet")
    assert GL.param_types("step, winSamples int, fftCoefs []complex128, power *etensor.Float64") == [
        "int", "int", "[]complex128", "*etensor.Float64"]
