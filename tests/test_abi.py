"""CPU-side checks of the drop-in boundary: the shared library loads and exports exactly the
entry points include/psvr_engine.h declares (no compute is launched here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "psvr_engine.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(psvr_[a-z0-9_]+)\s*\(", txt))
    names.discard("psvr_cigar_bound")  # static inline
    return names


def test_library_exports_every_declared_symbol():
    from pansvr_amd import lib
    L = lib()
    names = declared_symbols()
    assert len(names) >= 9
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_error_path_without_gpu_is_loud():
    from pansvr_amd import lib
    L = lib()
    if L.psvr_device_count() > 0:
        return
    from pansvr_amd import ksw, EngineError
    from ksw_cases import mat5
    p = ksw.make_params(5, mat5(2, 12), 16, 1, 32, 0, 200, 400, -1, 0)
    try:
        ksw.ext_batch([[0, 1]], [[0, 1]], p)
    except EngineError as ex:
        assert "no HIP device" in str(ex) or "failed" in str(ex)
    else:
        raise AssertionError("engine ran without a GPU: a CPU fallback must not exist")
