"""ctypes binding of librfgpu.so (include/rfgpu.h).  Fails loudly when the HIP
library has not been built: there is no other implementation behind it."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librfgpu.so")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)


class RFConfig(C.Structure):
    """struct rf_config of include/rfgpu.h"""
    _fields_ = [
        ("nfft", C.c_int32), ("ntrc", C.c_int32), ("nsmp", C.c_int32), ("deconv_mode", C.c_int32),
        ("delta", C.c_double), ("t_start", C.c_double), ("sdep", C.c_double),
        ("rayps", dp), ("a_gus", dp), ("ipha", ip),
        ("obs", dp), ("ldobs", C.c_int32),
        ("r_inv", dp),
        ("max_walkers", C.c_int32), ("nlay_max", C.c_int32), ("device", C.c_int32),
    ]


class RFModelConfig(C.Structure):
    """struct rf_model_config of include/rfgpu.h"""
    _fields_ = [
        ("k_max", C.c_int32), ("vp_mode", C.c_int32), ("nref", C.c_int32),
        ("z_max", C.c_double), ("h_min", C.c_double), ("z_ref_min", C.c_double), ("dz_ref", C.c_double),
        ("vp_min", C.c_double), ("vp_max", C.c_double), ("vs_min", C.c_double), ("vs_max", C.c_double),
        ("vpvs_min", C.c_double), ("vpvs_max", C.c_double),
        ("vp_ref", dp), ("vs_ref", dp),
    ]


class RFPostConfig(C.Structure):
    """struct rf_post_config of include/rfgpu.h"""
    _fields_ = [
        ("nbin_z", C.c_int32), ("nbin_vs", C.c_int32), ("nbin_vp", C.c_int32), ("nbin_vpvs", C.c_int32),
        ("nbin_sig", C.c_int32), ("nbin_amp", C.c_int32),
        ("amp_min", C.c_double), ("amp_max", C.c_double), ("z_min", C.c_double),
        ("sig_min", dp), ("sig_max", dp), ("sig_mode", ip),
        ("max_models", C.c_int64),
    ]


class RFPostResult(C.Structure):
    """struct rf_post_result of include/rfgpu.h"""
    _fields_ = [
        ("nmod", ip),
        ("nk", ip), ("nz", ip), ("nsig", ip), ("namp", ip), ("nvpz", ip), ("nvsz", ip), ("nvpvsz", ip),
        ("vp_mean", dp), ("vs_mean", dp), ("vpvs_mean", dp),
        ("vp_model", dp), ("vs_model", dp), ("all_likelihood", dp),
        ("amp_out_of_range", C.POINTER(C.c_int64)),
    ]


# every symbol include/rfgpu.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
SYMBOLS = {
    "rf_ctx_create": (C.c_int, [C.POINTER(RFConfig), C.POINTER(_vp)]),
    "rf_ctx_destroy": (C.c_int, [_vp]),
    "rf_last_error": (C.c_char_p, []),
    "rf_abi_version": (C.c_int, []),
    "rf_get_flt": (C.c_int, [_vp, dp]),
    "rf_get_is_ray_common": (C.c_int, [_vp, ip]),
    "rf_get_r_inv": (C.c_int, [_vp, dp]),
    "rf_compute_r_inv": (C.c_int, [C.c_int32, C.c_double, C.c_double, dp, ip, dp]),
    "rf_get_r_inv_info": (C.c_int, [_vp, ip, dp]),
    "rf_set_r_inv": (C.c_int, [_vp, dp]),
    "rf_calc_likelihood_of_trace": (C.c_int, [_vp, dp, dp, dp]),
    "rf_calc_rf": (C.c_int, [_vp, C.c_int32, dp, dp, dp, dp, dp]),
    "rf_calc_likelihood": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, dp, dp, dp, dp, dp, dp, dp]),
    "rf_eval_batch": (C.c_int, [_vp, C.c_int32, ip, ip, ip, C.c_int32, dp, dp, dp]),
    "rf_eval_batch_device": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp]),
    "rf_commit": (C.c_int, [_vp, C.c_int32, ip, ip]),
    "rf_commit_device": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp]),
    "rf_get_rft": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, dp]),
    "rf_get_rft_batch": (C.c_int, [_vp, C.c_int32, ip, C.c_int32, C.c_int32, dp]),
    "rf_set_model": (C.c_int, [_vp, C.POINTER(RFModelConfig)]),
    "rf_format_models_device": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp, _vp]),
    "rf_eval_models_device": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rf_eval_models": (C.c_int, [_vp, C.c_int32, ip, ip, ip, dp, C.c_int32, dp, dp, dp, dp, ip]),
    "rf_eval_models_begin": (C.c_int, [_vp, C.c_int32, ip, ip, ip, dp, C.c_int32, dp, dp, dp, C.c_int32, ip]),
    "rf_eval_wait": (C.c_int, [_vp, C.c_int32, dp, ip]),
    "rf_fft_c2r": (C.c_int, [C.c_int32, dp, dp]),
    "rf_fft_r2c": (C.c_int, [C.c_int32, dp, dp]),
    "rf_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_vp)]),
    "rf_host_free": (C.c_int, [_vp]),
    "rf_pt_swap_device": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rf_comm_device_key": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "rf_comm_probe": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "rf_comm_set_library": (C.c_int, [C.c_char_p]),
    "rf_comm_get_unique_id": (C.c_int, [C.c_char_p]),
    "rf_comm_init": (C.c_int, [_vp, C.c_char_p, C.c_int32, C.c_int32]),
    "rf_comm_destroy": (C.c_int, [_vp]),
    "rf_comm_info": (C.c_int, [_vp, ip, ip, ip]),
    "rf_comm_bcast_i32": (C.c_int, [_vp, ip, C.c_int32, C.c_int32]),
    "rf_comm_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "rf_comm_post_reduce": (C.c_int, [_vp, C.c_int32, ip]),
    "rf_comm_post_gather": (C.c_int, [_vp, C.c_int32, ip, dp, dp, dp]),
    "rf_pt_swap_exchange": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, dp, ip]),
    "rf_pt_swap_allgather_device": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "rf_pt_swap_gathered_device": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rf_post_create": (C.c_int, [_vp, C.POINTER(RFPostConfig)]),
    "rf_post_reset": (C.c_int, [_vp]),
    "rf_post_record": (C.c_int, [_vp, C.c_int32, ip, ip, dp, dp, dp, dp, dp, dp]),
    "rf_post_record_device": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rf_post_read": (C.c_int, [_vp, C.POINTER(RFPostResult)]),
    "rf_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "rf_get_launch_plan": (C.c_int, [_vp, ip]),
    "rf_profile_enable": (C.c_int, [_vp, C.c_int32]),
    "rf_profile_read": (C.c_int, [_vp, dp, C.POINTER(C.c_int64), C.c_int32]),
}

_lib = None


def load(path=None):
    """Load librfgpu.so and type every entry point.  Raises if it is not built.
    `path` (first call only; tools/ab.sh, bench.py --lib): another build of the same ABI for
    A/B timing -- an explicit argument, never an environment variable."""
    global _lib, LIB_PATH
    if _lib is not None:
        if path is not None and os.path.abspath(path) != os.path.abspath(LIB_PATH):
            raise RuntimeError(f"librfgpu already loaded from {LIB_PATH}")
        return _lib
    if path is not None:
        LIB_PATH = os.path.abspath(path)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C rf_inv_amd/csrc). "
            "rf_inv_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
    # /opt/rocm's).  Whichever is loaded first serves both; torch does not find its GPUs on
    # the system runtime, while librfgpu runs on either -- so let torch load first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def kernels_sha256(path=None):
    """sha256 of the library's `.hip_fatbin` section -- the gfx950 code objects of every kernel, nothing of the host
    side: what hardware counters of a kernel are a property of (bench.py's roofline matches committed counters on
    it, so that a host-only change of the library does not orphan them).  None if the section cannot be found."""
    import hashlib
    import struct

    path = path or LIB_PATH
    try:
        with open(path, "rb") as fh:
            b = fh.read()
        if b[:4] != b"\x7fELF" or b[4] != 2:
            return None
        shoff, = struct.unpack_from("<Q", b, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
        sec = lambda i: struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize)   # name, type, flags, addr, off, size, ...
        str_off = sec(shstrndx)[4]
        for i in range(shnum):
            name, _, _, _, off, size = sec(i)[:6]
            end = b.index(b"\0", str_off + name)
            if b[str_off + name:end] == b".hip_fatbin":
                return hashlib.sha256(b[off:off + size]).hexdigest()
    except (OSError, ValueError, struct.error):
        pass
    return None
