"""ctypes view of include/psvr_engine.h.  Loading fails loudly when the HIP library is absent."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpsvr_engine.so")


class EngineError(RuntimeError):
    pass


class Extz(C.Structure):  # psvr_extz_t
    _fields_ = [(n, C.c_int32) for n in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q",
                                          "score", "n_cigar", "reach_end", "reserved")] + [("cigar_off", C.c_int64)]


class KswParams(C.Structure):  # psvr_ksw_params_t
    _fields_ = [("m", C.c_int8), ("mat", C.c_int8 * 25), ("q", C.c_int8), ("e", C.c_int8), ("q2", C.c_int8), ("e2", C.c_int8),
                ("w", C.c_int32), ("zdrop", C.c_int32), ("end_bonus", C.c_int32), ("flag", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError("%s is missing: run `python -m pansvr_amd.build` (no CPU fallback exists)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.psvr_last_error.restype = C.c_char_p
        L.psvr_dp_plan_workspace_bytes.restype = C.c_int64
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise EngineError("psvr error %d: %s" % (rc, lib().psvr_last_error().decode()))
