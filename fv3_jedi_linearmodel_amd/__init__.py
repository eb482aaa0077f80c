"""fv3_jedi_linearmodel_amd — MI355X-native tangent-linear / adjoint FV3 dynamical core.

Host-side mirror of the reference operator API for the dynamics hot path
(reference: src/fv3jedi_lm_mod.F90:20-38, src/dynamics/fv3jedi_lm_dynamics_mod.F90:38-63).
The compute path is the HIP library csrc/ -> libfv3lm_hip.so behind the C-ABI include/fv3lm.h;
there is no CPU fallback: importing works without a GPU, creating a model does not.
"""
from .config import Options, Dims, default_options  # noqa: F401
from .grid import synthetic_tile_metrics, METRIC_NAMES  # noqa: F401
from ._lib import Fv3LmLibrary, load_hip_library, LIB_PATH  # noqa: F401
