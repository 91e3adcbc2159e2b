"""ctypes binding of lib/libhtm_hip.so (C ABI: include/htm_hip.h).

The HIP library is the product: if it is missing or cannot be loaded this module raises -- there is no
Python/NumPy fallback for any compute entry point.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HTM_LIB") or os.path.join(_HERE, "lib", "libhtm_hip.so")   # HTM_LIB: A/B builds

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
up = C.POINTER(C.c_uint32)
vp = C.c_void_p


class HtmError(RuntimeError):
    pass


class ModelInit(C.Structure):
    _fields_ = [("x", dp), ("mu", dp), ("sigma", dp), ("step_size", dp), ("prior_type", ip)]


class ChainsInit(C.Structure):
    _fields_ = [
        ("n_chains", C.c_int), ("n_procs", C.c_int), ("rank", C.c_int),
        ("hypo", ModelInit), ("t_corr", ModelInit), ("vs", ModelInit), ("a_corr", ModelInit), ("qs", ModelInit),
        ("temp", dp),
        ("solve_vs", C.c_int), ("solve_t_corr", C.c_int), ("solve_qs", C.c_int), ("solve_a_corr", C.c_int),
        ("rng_state", C.c_uint32 * 4),
        ("n_burn", C.c_int), ("n_interval", C.c_int),
        ("lik_capacity", C.c_int), ("sample_capacity", C.c_int),
    ]


# every symbol include/htm_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "htm_last_error": (C.c_char_p, []),
    "htm_abi_version": (C.c_int, []),
    "htm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "htm_device_physical_id": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "htm_forward_create": (C.c_int, [C.c_int, C.c_int] + [dp] * 7 + [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "htm_forward_destroy": (C.c_int, [vp]),
    "htm_forward_set_precision": (C.c_int, [vp, C.c_int]),
    "htm_forward_set_stream": (C.c_int, [vp, vp]),
    "htm_forward_reset_stream": (C.c_int, [vp]),
    "htm_forward_loglik_full": (C.c_int, [vp, dp, dp, C.c_double, dp, C.c_double, dp]),
    "htm_forward_loglik_partial": (C.c_int, [vp, C.c_int, dp, C.c_double, dp, dp, C.c_double, dp, C.c_double, dp]),
    "htm_forward_travel_time": (C.c_int, [vp, dp, dp, C.c_double, dp]),
    "htm_forward_amp": (C.c_int, [vp, dp, dp, C.c_double, C.c_double, dp]),
    "htm_forward_travel_time_single": (C.c_int, [vp, C.c_int, dp, dp, C.c_double, dp]),
    "htm_forward_amp_single": (C.c_int, [vp, C.c_int, dp, dp, C.c_double, C.c_double, dp]),
    "htm_forward_loglik_full_batch": (C.c_int, [vp, C.c_int, dp, dp, dp, dp, dp, dp]),
    "htm_forward_loglik_full_batch_dev": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp]),
    "htm_forward_sync": (C.c_int, [vp]),
    "htm_forward_time_full_batch_dev": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, dp]),
    "htm_chains_create": (C.c_int, [vp, C.POINTER(ChainsInit), C.POINTER(vp)]),
    "htm_chains_destroy": (C.c_int, [vp]),
    "htm_chains_run": (C.c_int, [vp, C.c_int]),
    "htm_chains_step_begin": (C.c_int, [vp]),
    "htm_chains_swap_record": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "htm_chains_step_end": (C.c_int, [vp, vp]),
    "htm_chains_run_lockstep": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "htm_chains_sync": (C.c_int, [vp]),
    "htm_chains_drain": (C.c_int, [vp]),
    "htm_chains_iterations_done": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "htm_chains_get_state": (C.c_int, [vp, C.c_int, dp, dp, dp, dp, dp, dp, dp, ip, ip]),
    "htm_chains_get_rng": (C.c_int, [vp, up]),
    "htm_chains_get_loglik": (C.c_int, [vp, C.c_int, dp]),
    "htm_chains_share_gpu": (C.c_int, [vp, C.c_int]),
    "htm_quantiles": (C.c_int, [C.c_int, dp, C.c_long, C.c_long, C.POINTER(C.c_int), dp]),
    "htm_quantiles_dev": (C.c_int, [C.c_int, vp, C.c_long, C.c_long, C.c_long, C.POINTER(C.c_int), vp, vp]),
    "htm_chains_swap_record_host": (C.c_int, [vp, dp]),
    "htm_chains_step_end_host": (C.c_int, [vp, dp]),
    "htm_comm_unique_id": (C.c_int, [vp, C.c_size_t]),
    "htm_comm_create": (C.c_int, [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "htm_comm_destroy": (C.c_int, [vp]),
    "htm_comm_allgather": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "htm_chains_run_lockstep_comm": (C.c_int, [vp, C.c_int, vp]),
    "htm_chains_xchg_handle": (C.c_int, [vp, vp, C.c_size_t]),
    "htm_chains_xchg_connect": (C.c_int, [vp, vp, C.c_size_t]),
    "htm_chains_run_lockstep_direct": (C.c_int, [vp, C.c_int]),
    "htm_chains_xchg_probe": (C.c_int, [vp, C.c_uint, C.c_double]),
    "htm_chains_checkpoint_size": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "htm_chains_checkpoint_save": (C.c_int, [vp, vp, C.c_size_t]),
    "htm_chains_checkpoint_load": (C.c_int, [vp, vp, C.c_size_t]),
    "htm_chains_lik_count": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "htm_chains_lik_read": (C.c_int, [vp, ip, ip, dp]),
    "htm_chains_sample_count": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "htm_chains_sample_read": (C.c_int, [vp, C.c_int, ip, ip, dp, dp, dp, dp, dp]),
    "htm_chains_clear_records": (C.c_int, [vp]),
    "htm_chains_enable_steplog": (C.c_int, [vp, C.c_int]),
    "htm_chains_steplog_read": (C.c_int, [vp, C.POINTER(C.c_int), ip, dp]),
    "htm_chains_handoff_stats": (C.c_int, [vp, C.POINTER(C.c_int64)]),
    "htm_chains_master_stats": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "htm_chains_last_run_stats": (C.c_int, [vp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "htm_chains_profile": (C.c_int, [vp, C.c_int, dp, C.POINTER(C.c_int), dp, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "htm_select_regress": (C.c_int, [C.c_int, C.c_int, C.c_int, dp, dp, dp, C.c_double, dp, dp, dp, dp, dp]),
    "htm_selftest": (C.c_int, [C.c_int]),
    "htm_selftest_math": (C.c_int, [C.c_int, C.c_int, dp, dp, C.c_int]),
    "htm_rng_jump": (C.c_int, [up, C.c_ulonglong, up]),
}

_LIB = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so.  Two HIP runtimes in one
    process do not work (the second one finds no device), so when torch is installed we bind libhtm_hip.so
    to torch's copy (same soname, libamdhip64.so.7) by loading it first.  Without torch -- e.g. under the
    Fortran driver -- the library uses /opt/rocm's runtime through its RUNPATH."""
    import importlib.util
    import sys

    if os.environ.get("HTM_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libhtm_hip.so and bind every declared symbol.  Raises HtmError if it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise HtmError(
            f"{LIB_PATH} is missing: build it with `make -C hypotremormcmc_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    older_build = os.environ.get("HTM_LIB") and os.environ.get("HTM_LIB_OLDER_BUILD") == "1"   # tools/ab.sh: A/B against a build of an older header
    for name, (res, args) in SIGNATURES.items():
        if older_build and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().htm_last_error()
        raise HtmError(f"libhtm_hip error {rc}: {msg.decode() if msg else '?'}")
