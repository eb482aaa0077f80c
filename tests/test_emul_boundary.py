"""CPU (-m "not gpu"): device-side traj_to_fv3 / pert_to_fv3 / fv3_to_pert of the host-emulation build (boundary_checks.py)."""
from common import Case, CubeCase
from boundary_checks import check_boundary_copies


def test_boundary_copies_periodic_tile():
    check_boundary_copies(Case(nx=12, ny=10, npz=8, n_split=2, k_split=1, dt=900.0, backend="emul", oracle=False, nq=2))


def test_boundary_copies_six_faces():
    check_boundary_copies(CubeCase(n=8, npz=6, n_split=2, k_split=2, backend="emul", nq=2), cube=True)


def test_boundary_copies_against_the_numpy_oracle():
    from boundary_checks import check_boundary_oracle
    check_boundary_oracle(CubeCase(n=8, npz=6, n_split=2, k_split=2, backend="emul", nq=2))
