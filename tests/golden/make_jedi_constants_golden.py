"""Generator of tests/golden/jedi_constants_ref.json: the JEDI physical constants read out of the reference's OWN
utils/fv3jedi_lm_const_mod.F90, compiled where it lies by oracle/ref/Makefile into oracle/_ref/libconst_ref.so (run in the
build container, where /root/reference exists).  Data only: 14 numbers."""
import ctypes as C
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = ["kappa", "cp", "zvir", "grav", "rgas", "rdry", "cpdry", "rvap", "runiv", "airmw", "h2omw", "radius", "omega", "pi"]


def read_reference():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libconst_ref.so"))
    v = (C.c_double * 14)()
    L.ref_jedi_constants(v)
    return dict(zip(NAMES, [float(x) for x in v]))


if __name__ == "__main__":
    d = read_reference()
    with open(os.path.join(ROOT, "tests", "golden", "jedi_constants_ref.json"), "w") as f:
        json.dump({k: v.hex() for k, v in d.items()}, f, indent=1)     # hex floats: bit-exact
    print(d)
