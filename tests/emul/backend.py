"""TEST INFRASTRUCTURE ONLY: run the host-buffer parity cases against the CPU thread-emulator
build of the product's kernel sources (tests/emul/build_emul.py).  The product's own loader
(auditory_amd.capi.load) never looks here; this context manager swaps the binding in-process
for the duration of a test and restores it afterwards."""
import contextlib
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_emul  # noqa: E402

from auditory_amd import capi, runtime  # noqa: E402


def bind(path):
    lib = C.CDLL(path)
    for name, (res, args) in capi.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


@contextlib.contextmanager
def emulated(variant="plain"):
    lib = bind(build_emul.build(variant))
    old_lib, old_ctx = capi._LIB, dict(runtime._CTX)
    capi._LIB = lib
    runtime._CTX.clear()
    try:
        yield lib
    finally:
        for c in runtime._CTX.values():
            c.close()
        runtime._CTX.clear()
        runtime._CTX.update(old_ctx)
        capi._LIB = old_lib
