"""ctypes binding of libmagma_hip.so (the C ABI declared in include/magma_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmagma_hip.so")

_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_d = C.c_double
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/magma_hip.h one to one
SIGNATURES = {
    "magma_hip_version": (_i, []),
    "magma_last_error": (C.c_char_p, []),
    "magma_pad": (_i, [_i]),
    "magma_workspace_bytes": (_sz, [_i]),
    "magma_device_count": (_i, [C.POINTER(_i)]),
    "magma_set_device": (_i, [_i]),
    "magma_dev_malloc": (_i, [C.POINTER(_p), _sz]),
    "magma_dev_free": (_i, [_p]),
    "magma_memcpy_h2d": (_i, [_p, _p, _sz, _p]),
    "magma_memcpy_d2h": (_i, [_p, _p, _sz, _p]),
    "magma_memset_zero": (_i, [_p, _sz, _p]),
    "magma_stream_sync": (_i, [_p]),
    "magma_prof_enable": (_i, [_i]),
    "magma_prof_reset": (_i, []),
    "magma_prof_get": (_i, [_i, C.POINTER(_i), C.POINTER(_d)]),
    "magma_kern_to_cov_batched": (_i, [_i, _p, _p, _p, _p, _i, _p, _p]),
    "magma_kern_to_cov_grid": (_i, [_i, _p, _d, _d, _d, _p, _p]),
    "magma_kern_to_inv_batched": (_i, [_i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "magma_estep_scatter": (_i, [_i, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p]),
    "magma_estep_gather": (_i, [_i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i64, _p, _p, _p, _i64,
                                _p, _p, _p, _p, _p, _p, _p]),
    "magma_estep_finish": (_i, [_i, _p, _d, _d, _d, _p, _p, _p, _p, _p, _i, C.POINTER(_d),
                                C.POINTER(_i), _p]),
    "magma_spd_inverse": (_i, [_i, _p, _p, _p, _p, _p]),
    "magma_spd_inverse_static": (_i, [_i, _p, _p, _p, _p]),
    "magma_update_inv": (_i, [_i, _p, _p, _p, _p, _i, _p, _p]),
    "magma_update_mean": (_i, [_i, _p, _p, _p, _p, _p, _i, _p, _p]),
    "magma_dmvnorm": (_i, [_i, _p, _p, _p, _p, C.POINTER(_d), C.POINTER(_i), _p]),
    "magma_indiv_eval": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _i, _p,
                              _p, _p, _p]),
    "magma_set_option": (_i, [C.c_char_p, C.c_long]),
    "magma_cpack_stride": (_i64, [_i]),
    "magma_indiv_gather_cov": (_i, [_i, _i, _p, _p, _p, _i, _p, _p]),
    "magma_theta0_eval": (_i, [_i, _p, _d, _d, _d, _p, _p, _p, _i, _p, C.POINTER(_d),
                               C.POINTER(_i), _p]),
    "magma_theta0_mstep": (_i, [_i, _p, C.POINTER(_d), _d, _p, _p, _p, _i, _d, _d, _i, _p,
                                C.POINTER(_d), C.POINTER(_d), C.POINTER(_d), C.POINTER(_i),
                                C.POINTER(_i), C.POINTER(_i), _p]),
    "magma_debug_poison_lds": (_i, [_p]),
    "magma_debug_resident_diag": (_i, [_p]),
    "magma_sum": (_i, [_i64, _p, _p, C.POINTER(_d), _p]),
    "magma_pred_gp": (_i, [_i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "magma_pred_gp_batched": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i,
                                   _p, _p, _p, _p]),
    "magma_gemm_nt": (_i, [_i, _i, _i, _d, _p, _i64, _p, _i64, _d, _p, _i64, _p]),
    "magma_copy_pad": (_i, [_i, _p, _i64, _p, _i, _p]),
    "magma_copy_unpad": (_i, [_i, _p, _p, _i64, _p]),
    "magma_lbfgsb_create": (_p, [_i, _i, _p, _i, _d, _d, _i]),
    "magma_lbfgsb_ask": (_i, [_p, _p, _p, C.POINTER(_i)]),
    "magma_lbfgsb_tell": (_i, [_p, _p, _p]),
    "magma_lbfgsb_ask_compact": (_i, [_p, _p, _p, C.POINTER(_i)]),
    "magma_lbfgsb_tell_compact": (_i, [_p, _i, _p, _p, _p]),
    "magma_indiv_mstep": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _d, _d, _i,
                               _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, C.POINTER(_i),
                               C.POINTER(_i64), _p]),
    "magma_indiv_mstep_resident": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _d, _d, _i,
                                        _p, _p, _p, _p, _p, _p, _p]),
    "magma_comm_unique_id": (_i, [_p]),
    "magma_ctx_create": (_i, [C.POINTER(_p), _i]),
    "magma_ctx_create_rccl": (_i, [C.POINTER(_p), _i, _i, _i, _p]),
    "magma_ctx_create_callback": (_i, [C.POINTER(_p), _i, _i, _i, _p, _p]),
    "magma_ctx_destroy": (None, [_p]),
    "magma_ctx_info": (_i, [_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "magma_ctx_stream": (_p, [_p]),
    "magma_ctx_selfcheck": (_i, [_p, _i]),
    "magma_dataset_create": (_i, [_p, C.POINTER(_p), _i64, _i64, _i64, _p, _p, _p, _i, _p]),
    "magma_dataset_destroy": (None, [_p]),
    "magma_dataset_info": (_i, [_p, C.POINTER(_i64), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "magma_dataset_grid": (_i, [_p, _p]),
    "magma_dataset_buffer": (_p, [_p, C.c_char_p]),
    "magma_get_mean": (_i, [_p, _p, _p]),
    "magma_get_cov": (_i, [_p, _p, _p, _i64]),
    "magma_set_mean_cov": (_i, [_p, _p, _p, _p, _i64]),
    "magma_set_m0": (_i, [_p, _p, _p, _i]),
    "magma_get_stats": (_i, [_p, _p, _i]),
    "magma_e_step": (_i, [_p, _p, _p, _p, _i, _d, _p, _i, C.POINTER(_i), C.POINTER(_d)]),
    "magma_m_step": (_i, [_p, _p, _p, _p, _i, _p, _p, _p]),
    "magma_logl_monitoring": (_i, [_p, _p, _p, _p, _i, C.POINTER(_d)]),
    "magma_sum_fi": (_i, [_p, _p, _p, _i, _i, _p]),
    "magma_em_iteration": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p]),
    "magma_training": (_i, [_p, _p, _p, _i, _p, _p, _i, _i, _p, _p, _p, C.POINTER(_i),
                            C.POINTER(_i), _p]),
    "magma_posterior_mu": (_i, [_p, C.POINTER(_p), _i64, _i64, _i64, _p, _p, _p, _i64, _p, _p, _p,
                                _i, _d, _p, C.POINTER(_i), C.POINTER(_i)]),
    "magma_pred_gp_resident": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "magma_train_new_gp": (_i, [_p, _i, _p, _p, _p, _p, _i64, _p, _p, _p, C.POINTER(_i)]),
    "magma_lbfgsb_result": (_i, [_p, _p, _p, _p, _p, _p]),
    "magma_lbfgsb_destroy": (None, [_p]),
}

_lib = None


class MagmaHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once). Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MagmaHipError(
                f"{LIB_PATH} not found: build it with `python -m magma_amd.build` "
                "(there is no CPU fallback for the HIP path)")
        # torch first: it brings its own HIP runtime, and the device buffers this library is
        # handed live in that runtime. Loaded before torch, the library would bind the system
        # runtime instead and see no device ("no ROCm-capable device is detected").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().magma_last_error()
        raise MagmaHipError(f"libmagma_hip error {rc}: {msg.decode() if msg else ''}")


def call(name, *args):
    """Call a status-returning entry point and raise on failure."""
    check(getattr(load(), name)(*args))
