"""ctypes access to (a) the oracle restatement (oracle/liboracle.so) and (b) the compiled
reference kswlib (oracle/_ref/libref_ksw.so, present only where /root/reference was).
Test infrastructure only."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_KSW_SO = os.path.join(ROOT, "oracle", "_ref", "libref_ksw.so")

EZ_FIELDS = ["max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "n_cigar", "reach_end"]


class OrcEz(C.Structure):
    _fields_ = [(n, C.c_int32) for n in EZ_FIELDS] + [("cigar_overflow", C.c_int32)]


class KswEz(C.Structure):  # ksw_extz_t, ksw2.h:26-35
    _fields_ = [("max", C.c_uint32, 31), ("zdropped", C.c_uint32, 1), ("max_q", C.c_int), ("max_t", C.c_int),
                ("mqe", C.c_int), ("mqe_t", C.c_int), ("mte", C.c_int), ("mte_q", C.c_int), ("score", C.c_int),
                ("m_cigar", C.c_int), ("n_cigar", C.c_int), ("reach_end", C.c_int), ("cigar", C.POINTER(C.c_uint32))]


def _u8(a):
    return (C.c_uint8 * len(a))(*a)


def _mat(c):
    from ksw_cases import mat5
    m = c.get("mat") or mat5(c["match"], c["mismatch"])
    return (C.c_int8 * len(m))(*m)


_orc = None


def oracle_lib():
    global _orc
    if _orc is None:
        _orc = C.CDLL(ORACLE_SO)
    return _orc


def run_oracle(c, kind="extd2", cap=4096):
    lib = oracle_lib()
    ez = OrcEz()
    cig = (C.c_uint32 * cap)()
    q, t = _u8(c["query"]), _u8(c["target"])
    if kind == "extd2":
        lib.orc_extd2(len(c["query"]), q, len(c["target"]), t, C.c_int8(c["m"]), _mat(c), C.c_int8(c["q"]), C.c_int8(c["e"]),
                      C.c_int8(c["q2"]), C.c_int8(c["e2"]), c["w"], c["zdrop"], c["end_bonus"], c["flag"], C.byref(ez), cig, cap)
    else:
        lib.orc_extz2(len(c["query"]), q, len(c["target"]), t, C.c_int8(c["m"]), _mat(c), C.c_int8(c["q"]), C.c_int8(c["e"]),
                      c["w"], c["zdrop"], c["end_bonus"], c["flag"], C.byref(ez), cig, cap)
    assert not ez.cigar_overflow
    out = {n: int(getattr(ez, n)) for n in EZ_FIELDS}
    out["cigar"] = [int(cig[i]) for i in range(ez.n_cigar)]
    return out


_ref = None


def ref_available():
    return os.path.exists(REF_KSW_SO)


def run_ref(c, kind="extd2"):
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_KSW_SO)
        _ref._libc = C.CDLL(None)
        _ref._libc.free.argtypes = [C.c_void_p]
    ez = KswEz()
    q, t = _u8(c["query"]), _u8(c["target"])
    if kind == "extd2":
        _ref.ksw_extd2_sse(None, len(c["query"]), q, len(c["target"]), t, C.c_int8(c["m"]), _mat(c), C.c_int8(c["q"]), C.c_int8(c["e"]),
                           C.c_int8(c["q2"]), C.c_int8(c["e2"]), c["w"], c["zdrop"], c["end_bonus"], c["flag"], C.byref(ez))
    else:
        _ref.ksw_extz2_sse(None, len(c["query"]), q, len(c["target"]), t, C.c_int8(c["m"]), _mat(c), C.c_int8(c["q"]), C.c_int8(c["e"]),
                           c["w"], c["zdrop"], c["end_bonus"], c["flag"], C.byref(ez))
    out = {n: int(getattr(ez, n)) for n in EZ_FIELDS}
    out["cigar"] = [int(ez.cigar[i]) for i in range(ez.n_cigar)]
    if ez.cigar:
        _ref._libc.free(C.cast(ez.cigar, C.c_void_p))
    return out
