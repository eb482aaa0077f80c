"""CPU (-m "not gpu") checks of the drop-in boundary: the HIP library (cross-compiled for gfx950 by __graft_entry__.build())
loads, exports every entry point include/fv3lm.h declares and nothing undeclared, the ctypes mirrors of the option / dims
structs have the C layout, and without a GPU the library refuses to create a model (no CPU fallback).  No compute calls."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "fv3lm.h")
SO = os.path.join(ROOT, "fv3_jedi_linearmodel_amd", "libfv3lm_hip.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(SO):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build()
    return C.CDLL(SO)


def declared():
    txt = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(fv3lm_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    names = declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_nothing_undeclared_is_exported(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", SO], text=True)
    exported = sorted(set(re.findall(r"\b(fv3lm_[a-z0-9_]+)$", out, flags=re.M)))
    extra = [n for n in exported if n not in declared()]
    assert not extra, extra


def test_struct_layouts_match_header():
    import fv3_jedi_linearmodel_amd as fv3
    txt = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), txt, flags=re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                decl = decl.replace("const int*", "intptr")
                ty, rest = decl.split(None, 1)
                out += [(n.strip(), ty) for n in rest.split(",")]
        return out
    for struct, cls in (("fv3lm_options", fv3.Options), ("fv3lm_dims", fv3.Dims)):
        c_fields = fields(struct)
        assert [n for n, _ in c_fields] == [n for n, _ in cls._fields_], struct
        for (n, ty), (_, ct) in zip(c_fields, cls._fields_):
            assert (ty == "int" and ct is C.c_int) or (ty == "double" and ct is C.c_double) or (ty == "intptr" and ct is C.POINTER(C.c_int)), (struct, n)


def test_metric_names_and_loud_failure_without_gpu(lib):
    lib.fv3lm_metric_names.restype = C.c_char_p
    names = lib.fv3lm_metric_names().decode().split(",")
    assert len(names) == 50 and names[0] == "area" and names[-1] == "cos_sg9"
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import fv3_jedi_linearmodel_amd as fv3
    lib.fv3lm_last_error.restype = C.c_char_p
    h = C.c_void_p()
    dims = fv3.Dims(nx=8, ny=8, npz=4, ntile=1, nq=0, n_split=1, k_split=1, face=0, dt=100.0)
    opt = fv3.default_options()
    metrics = (C.POINTER(C.c_double) * 50)()
    rc = lib.fv3lm_create(C.byref(h), C.byref(dims), C.byref(opt), metrics, C.c_double(1.0), C.c_double(1.0), None, None, None)
    assert rc != 0 and b"no HIP device" in lib.fv3lm_last_error()
