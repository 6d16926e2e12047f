"""pansvr_amd -- MI355X (gfx950) engine for panSVR's read-realignment hot path (`panSVR aln`).

Only what the path needs lives here: csrc/ (hand-written HIP kernels + the C-ABI shared library
libpsvr_engine.so) and thin ctypes bindings used by the tests and bench.py.  There is no CPU
fallback: every entry point raises if the HIP library or a GPU is missing."""
from ._lib import lib, EngineError, LIB_PATH  # noqa: F401
