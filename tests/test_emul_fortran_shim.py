"""CPU (-m "not gpu"): the ISO_C_BINDING shim driven by a Fortran program (fortran/shim_driver.F90), linked against the
host-emulation build of the library -- struct layouts, array repacking and the error path of the shim, end to end
(shim_checks.py).  test_gpu_fortran_shim.py runs the same against libfv3lm_hip.so on the MI355X."""
import os
import shutil
import pytest
from common import Case, build_emul, ROOT
from shim_checks import build_driver, run_shim_check


@pytest.mark.skipif(shutil.which("amdflang") is None, reason="no Fortran compiler")
def test_fortran_host_through_the_shim(tmp_path):
    so = build_emul()
    drv = build_driver(os.path.dirname(so), "fv3lm_emul", os.path.join(os.path.dirname(so), "shim_driver_emul"))
    c = Case(nx=12, ny=10, npz=8, n_split=2, k_split=1, dt=900.0, backend="emul", oracle=False, nq=2)
    run_shim_check(c, drv, str(tmp_path))
