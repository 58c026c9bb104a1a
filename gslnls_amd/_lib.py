"""ctypes loader for libgslnls_hip.so (the C ABI of include/gslnls_core.h).

There is no CPU fallback: if the HIP library is missing this module raises, and if no
GPU is present the entry points return GSLNLS_E_NODEVICE which the Python layer turns
into an exception.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSLNLS_LIB: another build of the same library (developer variants, e.g. the stamps build)
LIB_PATH = os.environ.get("GSLNLS_LIB") or os.path.join(_HERE, "libgslnls_hip.so")

DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int)

E_NODEVICE = -100
E_UNSUPPORTED = -101
E_INTERRUPTED = -102


MODEL_EXPR = 100


class Model(C.Structure):
    _fields_ = [("id", C.c_int), ("p", C.c_int), ("nx", C.c_int), ("x", C.c_void_p), ("x_on_device", C.c_int),
                ("expr", C.c_char_p), ("parnames", C.POINTER(C.c_char_p)), ("xnames", C.POINTER(C.c_char_p)), ("lowering", C.c_int)]


LOWERINGS = {"auto": 0, "vm": 1, "jit": 2}


def set_expr(m, expr, parnames, xnames, lowering="auto"):
    """fill the GSLNLS_MODEL_EXPR fields of a Model; returns the objects that own the C strings"""
    m.lowering = LOWERINGS[lowering]
    pn = (C.c_char_p * max(len(parnames), 1))(*[s.encode() for s in parnames])
    xn = (C.c_char_p * max(len(xnames), 1))(*[s.encode() for s in xnames])
    e = expr.encode()
    m.expr, m.parnames, m.xnames = e, C.cast(pn, C.POINTER(C.c_char_p)), C.cast(xn, C.POINTER(C.c_char_p))
    return e, pn, xn


class Sparse(C.Structure):
    """gslnls_sparse: dgRMatrix (format 0) / dgCMatrix (1) / dgTMatrix (2) as the Matrix package stores them"""
    _fields_ = [("format", C.c_int), ("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long),
                ("p", C.POINTER(C.c_int)), ("i", C.POINTER(C.c_int)), ("j", C.POINTER(C.c_int)),
                ("x", C.POINTER(C.c_double))]


FN_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)
JAC_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)
FVV_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)
LARGE_F_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)
LARGE_JAC_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(Sparse), C.c_void_p)


class Result(C.Structure):
    _fields_ = [("par", DP), ("covar", DP), ("resid", DP), ("grad", DP), ("niter", C.c_int), ("conv", C.c_int),
                ("ssr", C.c_double), ("ssrtol", C.c_double), ("neval", C.c_int * 3), ("info", C.c_int),
                ("chisq_init", C.c_double), ("irls_weights", DP), ("irls_psi", DP), ("irls_dpsi", DP),
                ("irls_sigma", C.c_double), ("irls_tol", C.c_double), ("irls_status", C.c_int),
                ("irls_niter", C.c_int), ("partrace", DP), ("ssrtrace", DP), ("mstart_nsp", C.c_int),
                ("mstart_nwsp", C.c_int), ("mstart_iters", C.c_int), ("mstart_stop", C.c_int),
                ("mstart_ssropt", C.c_double), ("loop_ms", C.c_float), ("n_launches", C.c_int),
                ("jtj_cond", C.c_double), ("n_steps", C.c_int), ("code_path", C.c_int)]


class LargeResult(C.Structure):
    _fields_ = [("par", DP), ("covar", DP), ("resid", DP), ("niter", C.c_int), ("conv", C.c_int), ("info", C.c_int),
                ("ssr", C.c_double), ("ssrtol", C.c_double), ("chisq_init", C.c_double), ("neval", C.c_int * 4),
                ("partrace", DP), ("ssrtrace", DP), ("n_passes", C.c_int), ("last_pass_ms", C.c_float)]


# every symbol include/gslnls_core.h declares (tests/test_abi.py checks the list against the header)
_SIGNATURES = {
    "gslnls_nls": (C.c_int, [C.POINTER(Model), C.c_void_p, C.c_int, C.c_int, C.c_int, DP, C.c_int, C.c_void_p,
                             C.c_int, DP, IP, DP, IP, C.c_int, DP, C.POINTER(Result)]),
    "gslnls_solver_served": (C.c_int, [IP, C.POINTER(Result)]),
    "gslnls_last_call_profile": (C.c_int, [DP, C.c_int]),
    "gslnls_last_matrix_path_profile": (C.c_int, [DP, C.c_int]),
    "gslnls_debug_bd_syrk_ms": (C.c_double, [C.c_int, C.c_int, C.c_int]),
    "gslnls_trace_text": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "gslnls_trace_set_order": (C.c_int, [IP, C.c_int]),
    "gslnls_format_trace": (C.c_size_t, [C.POINTER(Result), C.c_int, C.c_int, IP, C.c_int, C.c_char_p, C.c_size_t]),
    "gslnls_nls_fn": (C.c_int, [C.c_int, C.c_int, C.c_void_p, FN_CB, JAC_CB, FVV_CB, C.c_void_p, DP, C.c_void_p, DP, IP, DP,
                                C.POINTER(Result)]),
    "gslnls_nls_fn_loss": (C.c_int, [C.c_int, C.c_int, C.c_void_p, FN_CB, JAC_CB, FVV_CB, C.c_void_p, DP, C.c_void_p, DP, IP, DP,
                                     C.c_int, DP, C.POINTER(Result)]),
    "gslnls_nls_fn_mstart": (C.c_int, [C.c_int, C.c_int, C.c_void_p, FN_CB, JAC_CB, FVV_CB, C.c_void_p, DP, IP, C.c_void_p, DP,
                                       IP, DP, C.c_int, DP, C.POINTER(Result)]),
    "gslnls_dense_create": (C.c_void_p, [C.POINTER(Model), C.c_void_p, C.c_int, C.c_void_p, IP]),
    "gslnls_dense_destroy": (None, [C.c_void_p]),
    "gslnls_dense_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, DP, DP, IP, DP, C.c_int, C.POINTER(Result)]),
    "gslnls_trim_cache": (None, []),
    "gslnls_dense_time_pass": (C.c_float, [C.c_void_p, C.c_int, DP, C.c_int]),
    "gslnls_dense_loop_event_stats": (C.c_int, [C.c_void_p, DP, C.POINTER(C.c_longlong), C.c_int]),
    "gslnls_dense_set_swts": (C.c_int, [C.c_void_p, DP]),
    "gslnls_dense_diagnostics": (C.c_int, [C.c_void_p, C.c_int, DP, IP, DP, DP, DP]),
    "gslnls_lower_formula": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), IP, C.c_char_p, C.c_int]),
    "gslnls_set_comm": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong,
                                  C.c_int]),
    "gslnls_comm_get_unique_id": (C.c_int, [C.c_char_p]),
    "gslnls_comm_init_rank": (C.c_int, [C.c_char_p, C.c_int, C.c_int]),
    "gslnls_comm_init_file": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int]),
    "gslnls_comm_destroy": (None, []),
    "gslnls_comm_allgather_count": (C.c_longlong, []),
    "gslnls_comm_set_timing": (None, [C.c_int]),
    "gslnls_comm_allgather_ms": (C.c_double, [C.POINTER(C.c_longlong)]),
    "gslnls_comm_last_error": (C.c_char_p, []),
    "gslnls_dense_mstart": (C.c_int, [C.c_void_p, C.c_int, C.c_int, DP, DP, IP, DP, IP, C.POINTER(Result)]),
    "gslnls_mstart_batch": (C.c_int, [C.c_void_p, C.c_int, DP, DP, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_double, IP, DP, DP, C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "gslnls_mstart_record_size": (C.c_int, [C.c_int]),
    "gslnls_nls_large": (C.c_int, [C.POINTER(Model), C.c_void_p, C.c_int, DP, C.c_void_p, IP, DP,
                                   C.POINTER(LargeResult)]),
    "gslnls_large_create": (C.c_void_p, [C.POINTER(Model), C.c_void_p, C.c_int, C.c_void_p, IP]),
    "gslnls_large_create_sparse": (C.c_void_p, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, LARGE_F_CB, LARGE_JAC_CB,
                                                C.c_void_p, IP]),
    "gslnls_large_destroy": (None, [C.c_void_p]),
    "gslnls_large_solve": (C.c_int, [C.c_void_p, DP, IP, DP, C.POINTER(LargeResult)]),
    "gslnls_large_time_pass": (C.c_float, [C.c_void_p, C.c_int, DP, DP, C.c_int]),
    "gslnls_batch_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int, IP]),
    "gslnls_batch_destroy": (None, [C.c_void_p]),
    "gslnls_batch_irls": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, DP, DP, IP, DP, C.c_int, DP,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "gslnls_batch_last_passes": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "gslnls_batch_irls_gather": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, DP, DP, IP, DP, C.c_int, DP,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "gslnls_strerror": (C.c_char_p, [C.c_int]),
    "gslnls_algorithm_name": (C.c_char_p, [C.c_int]),
    "gslnls_device_count": (C.c_int, []),
    "gslnls_set_device": (C.c_int, [C.c_int]),
    "gslnls_version": (C.c_char_p, []),
    "gslnls_set_interrupt_hook": (None, [C.c_void_p]),
    "gslnls_expr_build": (C.c_int, [C.POINTER(Model), C.c_char_p, C.c_int]),
    "gslnls_expr_native_state": (C.c_int, [C.POINTER(Model), C.c_int]),
    "gslnls_expr_prefetch": (C.c_int, [C.POINTER(Model), C.c_int]),
    "gslnls_shutdown": (None, []),
    "gslnls_debug_wide_solve": (C.c_int, [C.c_int, DP, DP, C.c_double, DP, DP]),
    "gslnls_debug_wide_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_int, DP, DP]),
    "gslnls_debug_mchol_solve": (C.c_int, [C.c_int, DP, DP, C.c_double, DP, DP]),
    "gslnls_debug_mchol_solve_resident": (C.c_int, [C.c_int, C.c_void_p, DP, C.c_double, DP, DP]),
    "gslnls_debug_mchol_last_device_ms": (C.c_double, []),
    "gslnls_debug_mchol_timing": (None, [C.c_int]),
    "gslnls_debug_device_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "gslnls_debug_device_free": (C.c_int, [C.c_void_p]),
    "gslnls_debug_device_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "gslnls_debug_host_mchol_solve": (C.c_int, [C.c_int, DP, DP, C.c_double, DP, DP]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "gslnls_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C gslnls_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
        # the background compiler of GSLNLS_LOWER_AUTO must be stopped before the C runtime tears the compiler's own
        # static objects down (include/gslnls_core.h, gslnls_shutdown): Python's atexit runs long before that
        import atexit
        atexit.register(L.gslnls_shutdown)
    return _lib


def symbols():
    return sorted(_SIGNATURES)


def trace_text():
    """the text a verbose call of the reference prints (include/gslnls_core.h, gslnls_trace_text), of the last call"""
    L = lib()
    need = L.gslnls_trace_text(None, 0)
    buf = C.create_string_buffer(need + 1)
    L.gslnls_trace_text(buf, need + 1)
    return buf.value.decode()


def strerror(code):
    return lib().gslnls_strerror(int(code)).decode()


class GslnlsDeviceError(RuntimeError):
    pass


def check(rc):
    if rc == E_NODEVICE:
        raise GslnlsDeviceError("gslnls_amd: no usable HIP device (MI355X path has no CPU fallback)")
    if rc == E_UNSUPPORTED:
        raise NotImplementedError("gslnls_amd: configuration not lowered to the device")
    if rc == E_INTERRUPTED:
        raise KeyboardInterrupt("gslnls_amd: fit interrupted through the interrupt hook")
    return rc
