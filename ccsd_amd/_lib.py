"""ctypes binding of libccsd_hip.so (include/ccsd_hip.h).

The product path has exactly one backend: the hand-written HIP library built in-tree by
`__graft_entry__.build()`.  If it is missing, or no MI355X is visible, everything that needs
compute raises -- there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(HERE, "libccsd_hip.so")

ABI_VERSION = 5
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_WEIGHTS, ERR_RUNTIME, ERR_WORKSPACE = range(6)
SDE_VP, SDE_VE, SDE_SUBVP = 0, 1, 2
PRED_EULER, PRED_REVERSE, PRED_S4 = 0, 1, 2
CORR_NONE, CORR_LANGEVIN = 0, 1
TARGET_X, TARGET_ADJ, TARGET_RANK2 = 0, 1, 2

EXPORTS = [
    "ccsd_plan_create", "ccsd_plan_destroy", "ccsd_weight_count", "ccsd_rank2_dims", "ccsd_workspace_bytes",
    "ccsd_last_error", "ccsd_score", "ccsd_init_state", "ccsd_corrector_norms", "ccsd_corrector_apply",
    "ccsd_predictor", "ccsd_s4_apply", "ccsd_sampler_run", "ccsd_quantize", "ccsd_rank2_cells", "ccsd_profile_kernel", "ccsd_profile_stride", "ccsd_profile_read", "ccsd_profile_launches", "ccsd_debug_stamps",
    "ccsd_noise_draws", "ccsd_plan_query",
]
QUERIES = {"fused_r2": 0, "xa_variant": 1, "r2_lds_bytes": 2, "xa_lds_bytes": 3, "fused_loop": 4, "merged_r2": 5, "ew1": 6}
KERNEL_IDS = {"k_xa": 0, "k_gemm_p": 1, "k_hf_score": 2, "k_gemm_h": 3, "k_langevin_apply": 4, "k_r2": 5, "k_s4_apply": 6, "k_ew1": 7}


class StepCoef(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("sscale", "alpha", "pa", "pb", "pc", "m1", "s1", "d", "m2", "s2")]


class Config(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in (
            "abi_version", "N", "F", "is_cc", "d_min", "d_max",
            "x_depth", "x_nhid",
            "a_num_layers", "a_num_linears", "a_c_init", "a_c_hid", "a_c_final", "a_nhid", "a_adim", "a_num_heads",
            "a_is_cc_net", "h_num_layers", "h_num_linears", "h_nhid", "h_adim", "h_c_hid", "h_c_final", "h_num_heads",
            "f_num_layers", "f_num_linears", "f_nhid", "f_c_hid", "f_c_final", "f_cnum", "f_num_layers_mlp",
            "f_use_hodge_mask",
            "predictor", "corrector", "n_corr_steps", "probability_flow", "denoise")]
        + [("snr", C.c_float), ("scale_eps", C.c_float), ("diff_steps", C.c_int32), ("batch_hint", C.c_int32)]
        + [(n, C.c_int32) for n in ("x_gmh", "x_num_linears", "x_c_init", "x_c_hid", "x_c_final", "x_adim", "x_num_heads",
                                    "a_conv_mlp", "x_conv_mlp")]
    )


class State(C.Structure):
    _fields_ = [("x", C.c_void_p), ("adj", C.c_void_p), ("rank2", C.c_void_p)]


class Noise(C.Structure):
    _fields_ = [("zx", C.c_void_p), ("zadj", C.c_void_p), ("zrank2", C.c_void_p)]


class CcsdError(RuntimeError):
    pass


def raise_status(lib, status: int):
    """Map status codes onto the exception types the reference raises on this path."""
    if status == OK:
        return
    msg = lib.ccsd_last_error().decode("utf-8", "replace")
    if status == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if status in (ERR_INVALID, ERR_WEIGHTS, ERR_WORKSPACE):
        raise ValueError(msg)
    raise CcsdError(msg)


class Library:
    """A loaded C-ABI library with typed entry points."""

    def __init__(self, path: str, is_hip: bool):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.is_hip = is_hip
        L = self.c = C.CDLL(path)
        vp, i32, i64, u64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_size_t
        P = C.POINTER
        L.ccsd_plan_create.argtypes = [P(Config), P(C.c_float), sz, P(StepCoef), P(vp)]
        L.ccsd_plan_create.restype = C.c_int
        L.ccsd_plan_destroy.argtypes = [vp]
        L.ccsd_plan_destroy.restype = None
        L.ccsd_weight_count.argtypes = [P(Config)]
        L.ccsd_weight_count.restype = sz
        L.ccsd_rank2_dims.argtypes = [P(Config), P(i32), P(i64)]
        L.ccsd_rank2_dims.restype = None
        L.ccsd_workspace_bytes.argtypes = [vp, i32]
        L.ccsd_workspace_bytes.restype = sz
        L.ccsd_last_error.argtypes = []
        L.ccsd_last_error.restype = C.c_char_p
        L.ccsd_score.argtypes = [vp, i32, i32, P(State), vp, f32, vp, vp, sz, vp]
        L.ccsd_score.restype = C.c_int
        L.ccsd_init_state.argtypes = [vp, i32, vp, P(Noise), u64, i64, P(State), vp]
        L.ccsd_init_state.restype = C.c_int
        L.ccsd_corrector_norms.argtypes = [vp, i32, i32, i32, P(State), P(State), vp, P(Noise), u64, i64, vp, vp, sz, vp]
        L.ccsd_corrector_norms.restype = C.c_int
        L.ccsd_corrector_apply.argtypes = [vp, i32, i32, i32, P(State), vp, P(Noise), u64, i64, vp, P(State), vp, sz, vp]
        L.ccsd_corrector_apply.restype = C.c_int
        L.ccsd_predictor.argtypes = [vp, i32, i32, P(State), vp, P(Noise), u64, i64, P(State), P(State), vp, sz, vp]
        L.ccsd_predictor.restype = C.c_int
        L.ccsd_s4_apply.argtypes = [vp, i32, i32, P(State), vp, P(Noise), P(Noise), P(Noise), u64, i64, vp, P(State), P(State), vp, sz, vp]
        L.ccsd_s4_apply.restype = C.c_int
        L.ccsd_sampler_run.argtypes = [vp, i32, vp, u64, i64, i32, i32, P(State), P(State), P(State), vp, vp, sz, vp]
        L.ccsd_sampler_run.restype = C.c_int
        L.ccsd_quantize.argtypes = [vp, i64, f32, vp, vp]
        L.ccsd_quantize.restype = C.c_int
        L.ccsd_rank2_cells.argtypes = [vp, i32, i32, i64, f32, vp, vp, vp]
        L.ccsd_rank2_cells.restype = C.c_int
        L.ccsd_profile_kernel.argtypes = [vp, i32]
        L.ccsd_profile_kernel.restype = C.c_int
        L.ccsd_profile_stride.argtypes = [vp, i32]
        L.ccsd_profile_stride.restype = C.c_int
        L.ccsd_profile_read.argtypes = [vp, i32, P(i64), P(C.c_double)]
        L.ccsd_profile_read.restype = C.c_int
        L.ccsd_profile_launches.argtypes = [vp, i32, P(i64)]
        L.ccsd_profile_launches.restype = C.c_int
        L.ccsd_debug_stamps.argtypes = [vp, vp]
        L.ccsd_debug_stamps.restype = C.c_int
        L.ccsd_noise_draws.argtypes = [vp, i32, vp, u64, i64, i32, i32, P(State), vp]
        L.ccsd_noise_draws.restype = C.c_int
        L.ccsd_plan_query.argtypes = [vp, i32, P(i64)]
        L.ccsd_plan_query.restype = C.c_int

    def __getattr__(self, name):
        return getattr(self.c, name)

    def check(self, status: int):
        raise_status(self.c, status)


_hip: Optional[Library] = None


def get_library() -> Library:
    """The HIP library.  Fails loudly when it has not been built (python __graft_entry__.py).
    CCSD_LIB_PATH (developer A/B runs only, tools/dev/*.sh) names another build of the SAME HIP library -- e.g. a diagnostic build with
    cycle stamps -- instead of overwriting the product file; it is still a HIP library (is_hip: a GPU is required) and a missing
    file is an error, never a fallback."""
    global _hip
    if _hip is None:
        # torch first: its bundled HIP runtime (libamdhip64) must be the one in the process before libccsd_hip.so asks for that SONAME --
        # loaded the other way round (library, then torch) the two copies do not share their device state and the library's first
        # hipMalloc reports "no ROCm-capable device is detected" (seen with __graft_entry__.build() called ahead of any torch import)
        import torch  # noqa: F401

        path = os.environ.get("CCSD_LIB_PATH") or HIP_LIB_PATH
        if not os.path.exists(path):
            raise CcsdError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  ccsd_amd has no CPU fallback."
            )
        _hip = Library(path, is_hip=True)
    return _hip
