"""-m gpu: the message path of the halo exchange (csrc/exchange.h: pack kernel -> ncclGroupStart, ncclSend / ncclRecv, ncclGroupEnd -> unpack
kernel; the adjoint back the other way; two of the exchanges on the second stream) run through RCCL on the one GPU at hand.  A one-rank
communicator with the rank as its own peer: cube.split_table(loopback=True) turns every row between two resident tiles into a message, so
all halo data of the six faces (or of the 24 sub-face tiles) travel through ncclSend / ncclRecv instead of the local gather.  What this
cannot show is two processes meeting in one group call; the gloo runs of test_dist_cube.py cover that side with the same lists."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm():
    import fv3_jedi_linearmodel_amd as fv3
    from fv3_jedi_linearmodel_amd._lib import comm_init_rccl
    lib = fv3.load_hip_library()
    comm_init_rccl(lib, 0, 1, lambda data: data)
    yield lib
    assert lib.L.fv3lm_comm_destroy() == 0


def test_six_faces_against_the_oracle(comm):
    from common import CubeCase
    from oracle import TL, AD
    from groups import cube_check_fv_dynamics, cube_check_tracer, cube_dot_product_step
    c = CubeCase(n=8, npz=6, n_split=2, k_split=2, backend="hip", oracle=True, nq=2, loopback=True)
    cube_check_fv_dynamics(c, TL, 1e-10)
    cube_check_fv_dynamics(c, AD, 1e-10)
    cube_check_tracer(c, TL, 1e-11, scale=80.0)       # sub-cycling: q exchanged between the sub-steps, maxima through ncclAllReduce
    cube_check_tracer(c, AD, 1e-10, scale=80.0)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)


def _same_as_local(make, tol=1e-12):
    from layout_checks import run_steps
    a, b = run_steps(make(True)), run_steps(make(False))
    n = next(iter(b.values())).shape[-1] - 7
    I = (Ellipsis, slice(3, 3 + n), slice(3, 3 + n))
    for key in b:
        x, y = a[key][I], b[key][I]
        assert np.isfinite(x).all() and np.abs(y).max() > 0, key
        e = float(np.max(np.abs(x - y)) / np.max(np.abs(y)))
        assert e <= (0.0 if key[0] == "tl" else tol), (key, e)       # the forward exchange is a copy either way: bit for bit


def test_messages_equal_the_local_gather_c48(comm):
    from common import CubeCase
    _same_as_local(lambda lb: CubeCase(n=48, npz=12, n_split=3, k_split=2, dt=900.0, backend="hip", nq=2, loopback=lb))


def test_messages_equal_the_local_gather_24_tiles(comm):
    from common import CubeCase
    _same_as_local(lambda lb: CubeCase(n=32, npz=6, n_split=2, k_split=2, dt=600.0, backend="hip", nq=2, layout=2, loopback=lb))


def test_messages_equal_the_local_gather_nonhydrostatic(comm):
    from common import CubeCase
    _same_as_local(lambda lb: CubeCase(n=32, npz=8, n_split=2, k_split=1, dt=150.0, backend="hip", nq=1, hydrostatic=0, loopback=lb), tol=1e-11)
