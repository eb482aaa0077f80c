"""CPU (-m "not gpu"): the fused fv_tp_2d launch of the host-emulation build against the staged launches, bit for bit
(tp_fused_checks.py).  The same check runs on the MI355X in test_gpu_parity.py."""
from common import Case, CubeCase
from tp_fused_checks import check_fused_equals_staged


def test_fused_tp_periodic_tile_many_blocks():
    # 70 x 20 cells: 2 x 2 blocks, the second ones partial; 4 tracers through tracer_2d's fv_tp_2d as well
    check_fused_equals_staged(lambda: Case(nx=70, ny=20, npz=3, n_split=2, k_split=1, dt=900.0, backend="emul", oracle=False, nq=2))


def test_fused_tp_cube_face_many_blocks():
    # six C66 faces (edge values, copy_corners views): 2 x 5 blocks per face; default options = first-order sponge levels + hord 2 below
    check_fused_equals_staged(lambda: CubeCase(n=66, npz=10, n_split=1, k_split=1, dt=225.0, backend="emul", nq=1))


def test_fused_tp_nonhydrostatic_interfaces():
    # non-hydrostatic: the height transport runs fv_tp_2d on npz+1 interfaces with the global hord_tm
    check_fused_equals_staged(lambda: Case(nx=66, ny=18, npz=10, n_split=1, k_split=1, dt=300.0, backend="emul", oracle=False, hydrostatic=0))


