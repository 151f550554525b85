"""rf_inv_amd -- MI355X (gfx950) forward + likelihood engine for RF_INV.

Host-side mirror of the reference's `params`, `model`, `forward` and `likelihood`
module interfaces (reference: /root/reference/src/*.f90) over the C ABI of
librfgpu (include/rfgpu.h).  Every evaluation runs hand-written HIP kernels; there is
no CPU fallback -- importing works anywhere, evaluating needs a gfx950 device and
the built rf_inv_amd/lib/librfgpu.so.
"""
from .params import Params, get_params, read_obs, read_sac, write_params  # noqa: F401
from .model import RefModel, read_ref_model, format_model, vp_to_rho  # noqa: F401
from .engine import RFEngine, RFGPUError  # noqa: F401
from .forward import Forward  # noqa: F401
from .likelihood import Likelihood  # noqa: F401
from .make_syn import make_syn, write_sac  # noqa: F401

__version__ = "0.1.0"
