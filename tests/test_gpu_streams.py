"""-m gpu: the stream placement of the face-edge strips must not change a result.  In the forward modes the strips of a stage run beside its
bulk launch on a third stream (dycore.h add_face); in the adjoint they go out in pairs with disjoint input regions (exec.h Pair).  Both are
switched off by environment variables read at create: the tangent step must agree bit for bit (the same kernels, only their placement
differs), the adjoint to rounding (the corner-alias launches of a pair follow both strips instead of each its own)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hydro", [1, 0])
def test_strip_streams_do_not_change_results(hydro, monkeypatch):
    from common import CubeCase
    from layout_checks import run_steps
    kw = dict(n=48, npz=8, n_split=3, k_split=2, dt=600.0, backend="hip", oracle=False, nq=2, hydrostatic=hydro)
    a = run_steps(CubeCase(**kw))
    monkeypatch.setenv("FV3LM_NO_SIDE_STRIPS", "1"); monkeypatch.setenv("FV3LM_NO_PAIR_STRIPS", "1")
    b = run_steps(CubeCase(**kw))
    for key in b:
        assert np.isfinite(a[key]).all() and np.abs(b[key]).max() > 0, key
        if key[0] == "tl":
            assert np.array_equal(a[key], b[key]), key
        else:
            e = float(np.max(np.abs(a[key] - b[key])) / np.max(np.abs(b[key])))
            assert e <= 1e-13, (key, e)


def test_fused_transport_forms_agree(monkeypatch):
    """fv_tp_2d in its current forms (tp2.h forward kernels, tpad.h one-launch adjoint) against round 2's (first tiled forward kernel,
    outer adjoint fused + inner adjoint staged): different kernels, same arithmetic up to the contraction of a*b+c"""
    from common import CubeCase
    from layout_checks import run_steps
    kw = dict(n=70, npz=6, n_split=2, k_split=1, dt=600.0, backend="hip", oracle=False, nq=1)
    a = run_steps(CubeCase(**kw))
    monkeypatch.setenv("FV3LM_TP2", "0"); monkeypatch.setenv("FV3LM_TP_AD_FUSED", "1")
    b = run_steps(CubeCase(**kw))
    for key in b:
        e = float(np.max(np.abs(a[key] - b[key])) / np.max(np.abs(b[key])))
        assert np.isfinite(a[key]).all() and e <= (1e-12 if key[0] == "tl" else 1e-11), (key, e)
