"""Options the path does not build are refused at create with a message, never approximated: checked on the
host-emulation build of the same create() code."""
import pytest
from common import Case


@pytest.mark.parametrize("kw, needle", [
    (dict(nord=2), "nord"),
    (dict(hord_dp=10, hord_dp_pert=10), "hord"),          # the tangent / adjoint exists for 1, 2, 333 only
    (dict(hord_dp=14), "hord"),                           # trajectory schemes built: 1 .. 13, 333
    (dict(hord_mt=0, hord_mt_pert=2), "hord"),
    (dict(hord_tm=5, hord_tm_pert=5), "hord"),            # 3 .. 7 are trajectory schemes only
    (dict(hydrostatic=0, a_imp=0.4), "a_imp"),
    (dict(kord_tm=-7), "kord"),                          # trajectory profiles built: linear (> 16) and the limited 8 .. 15 of cs_profile / scalar_profile
    (dict(kord_tm_pert=-9), "kord"),                     # the perturbation profile is the linear one
    (dict(hydrostatic=0, kord_wz=16), "kord"),
    (dict(kord_tr=16), "kord"),
])
def test_unsupported_options_are_refused(kw, needle):
    with pytest.raises(Exception) as e:
        Case(nx=10, ny=8, npz=12, backend="emul", oracle=False, **kw)      # 12 levels: some lie below the perturbation sponge
    assert needle in str(e.value), str(e.value)


def test_nonhydrostatic_needs_three_levels():
    with pytest.raises(Exception) as e:
        Case(nx=10, ny=8, npz=2, backend="emul", oracle=False, hydrostatic=0)
    assert "npz" in str(e.value)


def test_split_hord_needs_the_fused_transport(monkeypatch):
    """the values-only trajectory pass exists in the tiled fused fv_tp_2d only: with the staged form forced, create fails loudly"""
    monkeypatch.setenv("FV3LM_TP_FUSED", "0")
    with pytest.raises(Exception) as e:
        Case(nx=10, ny=8, npz=12, backend="emul", oracle=False, hord_dp=10)
    assert "split_hord" in str(e.value), str(e.value)


def test_package_refuses_a_host_emulation_library(monkeypatch):
    """FV3LM_LIB may name another HIP build of the sources, never the tests' g++ build: the package has no CPU path"""
    from common import build_emul
    from fv3_jedi_linearmodel_amd._lib import load_hip_library, Fv3LmError
    monkeypatch.setenv("FV3LM_LIB", build_emul())
    with pytest.raises(Fv3LmError, match="host-emulation"):
        load_hip_library()
