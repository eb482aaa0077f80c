"""-m gpu: a Fortran program linked against libfv3lm_hip.so through the ISO_C_BINDING shim runs step_tl / step_ad on the MI355X
and reproduces the ctypes-driven results bit for bit (shim_checks.py); built by __graft_entry__.build()."""
import os
import pytest
from shim_checks import run_shim_check

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fortran_host_through_the_shim_on_the_gpu(tmp_path):
    from common import Case
    drv = os.path.join(ROOT, "fortran", "shim_driver")
    assert os.path.exists(drv), "fortran/shim_driver missing: run __graft_entry__.build()"
    c = Case(nx=24, ny=20, npz=16, n_split=2, k_split=1, dt=900.0, backend="hip", oracle=False, nq=2)
    run_shim_check(c, drv, str(tmp_path))
