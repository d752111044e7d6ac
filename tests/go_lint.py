"""TEST INFRASTRUCTURE ONLY: a static check of the cgo mirror under go/ against include/auditory_hip.h.

The image has no Go toolchain, so go/ has never been compiled.  This is not a Go front end;
it checks the classes of mistakes that a hand-kept binding collects and that a compiler would stop at:
  * brackets balance in every file (comments and literals blanked first);
  * every C.aud_* call names a function the header declares and passes as many arguments as it has parameters;
  * every C.AUD_* constant exists in the header;
  * every keyed C.aud_* struct literal, and every field selected on a variable declared with a C.aud_* struct type, uses
    fields the header's struct has;
  * no function, method or type is declared twice in a package;
  * every import is used and the standard packages that are used are imported."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "auditory_hip.h")
GO_ROOT = os.path.join(ROOT, "go")

STD_PKGS = ("fmt", "math", "errors", "unsafe", "sync", "os", "log", "strings", "sort", "time", "runtime", "flag", "bufio",
            "encoding/binary", "path/filepath", "io", "testing", "reflect", "strconv")


def blank_go(src):
    """comments and the contents of string / rune literals replaced by spaces (newlines kept), so that brackets and
    identifiers that remain are code"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        two = src[i:i + 2]
        if two == "//":
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i))
            i = j
        elif two == "/*":
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j]))
            i = j
        elif c == "`":
            j = src.find("`", i + 1)
            j = n if j < 0 else j + 1
            out.append("`" + "".join(ch if ch == "\n" else " " for ch in src[i + 1:j - 1]) + "`")
            i = j
        elif c in "\"'":
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(c + " " * (j - i - 1) + c)
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def check_balance(code, name):
    pairs = {")": "(", "]": "[", "}": "{"}
    stack = []
    line = 1
    for ch in code:
        if ch == "\n":
            line += 1
        elif ch in "([{":
            stack.append((ch, line))
        elif ch in pairs:
            assert stack and stack[-1][0] == pairs[ch], "%s:%d: unbalanced %r" % (name, line, ch)
            stack.pop()
    assert not stack, "%s:%d: unclosed %r" % (name, stack[-1][1], stack[-1][0])


def split_top(s):
    """split at commas that are not inside brackets"""
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    tail = "".join(cur).strip()
    if tail or parts:
        parts.append(tail)
    return [p.strip() for p in parts]


def matching(code, open_at):
    """index of the bracket that closes the one at open_at"""
    depth = 0
    for j in range(open_at, len(code)):
        if code[j] in "([{":
            depth += 1
        elif code[j] in ")]}":
            depth -= 1
            if depth == 0:
                return j
    raise AssertionError("no closing bracket")


class Header:
    def __init__(self, path=HEADER):
        src = open(path).read()
        code = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
        code = re.sub(r"//[^\n]*", " ", code)
        self.consts = set(re.findall(r"#define\s+(AUD_\w+)", code))
        for body in re.findall(r"enum\s*\w*\s*\{([^}]*)\}", code):
            self.consts.update(re.findall(r"\b(AUD_\w+)\b", body))
        self.funcs = {}
        for m in re.finditer(r"\b(aud_\w+)\s*\(([^()]*)\)\s*;", code):
            params = m.group(2).strip()
            self.funcs[m.group(1)] = 0 if params in ("", "void") else len(split_top(params))
        self.structs = {}
        self.field_type = {}
        for m in re.finditer(r"typedef\s+struct\s*\w*\s*\{([^{}]*)\}\s*(aud_\w+)\s*;", code):
            fields = {}
            for decl in m.group(1).split(";"):
                decl = decl.strip()
                if not decl:
                    continue
                mt = re.match(r"((?:const\s+|unsigned\s+|struct\s+)*\w+[\s\*]+)(.*)$", decl, flags=re.S)
                assert mt, decl
                for d in mt.group(2).split(","):
                    fname = re.match(r"[\s\*]*(\w+)", d).group(1)
                    fields[fname] = mt.group(1).strip()
            self.structs[m.group(2)] = fields
        self.opaque = set(re.findall(r"typedef\s+struct\s+\w+\s+(aud_\w+)\s*;", code))


def c_calls(code):
    """(name, [args], line) of every C.aud_* call"""
    for m in re.finditer(r"\bC\.(aud_\w+)\s*\(", code):
        close = matching(code, m.end() - 1)
        yield m.group(1), split_top(code[m.end():close]), code.count("\n", 0, m.start()) + 1


def c_literals(code):
    """(type, [keys], line) of every keyed composite literal C.aud_x{...}"""
    for m in re.finditer(r"\bC\.(aud_\w+)\s*\{", code):
        if code[:m.start()].rstrip().endswith(")"):
            continue                                   # a result type in front of a function body, not a literal
        close = matching(code, m.end() - 1)
        keys = []
        for part in split_top(code[m.end():close]):
            mk = re.match(r"(\w+)\s*:", part)
            if mk:
                keys.append(mk.group(1))
        yield m.group(1), keys, code.count("\n", 0, m.start()) + 1


def typed_vars(code):
    """{variable: C struct type} for `var x C.aud_t`, `x := C.aud_t{`, `x *C.aud_t` / `x C.aud_t` in parameter lists"""
    out = {}
    for m in re.finditer(r"\bvar\s+(\w+)\s+\*?C\.(aud_\w+)", code):
        out[m.group(1)] = m.group(2)
    for m in re.finditer(r"\b(\w+)\s*:=\s*&?C\.(aud_\w+)\s*\{", code):
        out[m.group(1)] = m.group(2)
    for m in re.finditer(r"[(,]\s*(\w+)\s+\*?C\.(aud_\w+)\s*[,)]", code):
        out[m.group(1)] = m.group(2)
    return out


def func_texts(code):
    """the text of each top-level function (signature and body): variable names are scoped to these"""
    starts = [m.start() for m in re.finditer(r"^func\b", code, flags=re.M)]
    for a, b in zip(starts, starts[1:] + [len(code)]):
        yield code[a:b]


def declarations(code):
    """top-level (kind, receiver type, name, line)"""
    for m in re.finditer(r"^func\s*(?:\(\s*\w*\s*\*?(\w+)\s*\)\s*)?(\w+)\s*\(", code, flags=re.M):
        yield "func", m.group(1) or "", m.group(2), code.count("\n", 0, m.start()) + 1
    for m in re.finditer(r"^type\s+(\w+)\s", code, flags=re.M):
        yield "type", "", m.group(1), code.count("\n", 0, m.start()) + 1


def param_types(params):
    """Go parameter list -> list of types ("a, b int, c []float64" -> [int, int, []float64]); every parameter is named in
    this tree and in the reference's signatures"""
    items = split_top(params)
    out, pending = [], 0
    for it in items:
        if not it:
            continue
        bits = it.split(None, 1)
        if len(bits) == 1:
            pending += 1
        else:
            out.extend([re.sub(r"\s+", "", bits[1])] * (pending + 1))
            pending = 0
    assert pending == 0, params
    return out


def go_funcs(code):
    """{(receiver type, name): [parameter types]} of the file's top-level functions and methods"""
    out = {}
    for m in re.finditer(r"^func\s*(?:\(\s*\w*\s*\*?(\w+)\s*\)\s*)?(\w+)\s*\(", code, flags=re.M):
        close = matching(code, m.end() - 1)
        out[(m.group(1) or "", m.group(2))] = param_types(code[m.end():close])
    return out


def calls_of(code, pattern):
    """(name, n_args, line) of calls matched by `pattern` (one group: the name; the match ends at the opening bracket)"""
    for m in re.finditer(pattern, code):
        close = matching(code, m.end() - 1)
        yield m.group(1), len(split_top(code[m.end():close])), code.count("\n", 0, m.start()) + 1


def imports(src):
    """{local name: path}; `import "C"` left out"""
    out = {}
    blocks = re.findall(r"^import\s*\(([^)]*)\)", src, flags=re.M) + re.findall(r"^import\s+((?:\w+\s+)?\"[^\"]+\")", src, flags=re.M)
    for block in blocks:
        for m in re.finditer(r"(?:(\w+)\s+)?\"([^\"]+)\"", block):
            if m.group(2) == "C":
                continue
            out[m.group(1) or m.group(2).rsplit("/", 1)[-1]] = m.group(2)
    return out


def go_files():
    for d, _, files in sorted(os.walk(GO_ROOT)):
        for f in sorted(files):
            if f.endswith(".go"):
                yield os.path.join(d, f)
