"""ctypes binding of libalq.so (include/alq.h).  No fallback: if the HIP library is missing the
import of any device function raises, so a CPU-only or stale install fails loudly."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get('ALQ_LIB', 'libalq.so'))   # ALQ_LIB: diagnostic builds
BUILD_SCRIPT = os.path.join(_HERE, 'csrc', 'build.sh')

ALQ_CONV, ALQ_CONVT, ALQ_POOL, ALQ_FC = 0, 1, 2, 3


class AlqError(RuntimeError):
    pass


class LayerT(C.Structure):
    """Mirror of alq_layer_t."""
    _fields_ = [('type', C.c_int32), ('cout', C.c_int32), ('k', C.c_int32 * 3), ('s', C.c_int32 * 3),
                ('relu', C.c_int32), ('skip_src', C.c_int32)]


_P = C.c_void_p
_SIGNATURES = {
    'alq_last_error': (C.c_char_p, []),
    'alq_version': (C.c_int, []),
    'alq_ctx_create': (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    'alq_ctx_destroy': (C.c_int, [_P]),
    'alq_ctx_use_side_stream': (C.c_int, [_P, C.c_int]),
    'alq_ctx_set_stream': (C.c_int, [_P, _P]),
    'alq_ctx_synchronize': (C.c_int, [_P]),
    'alq_model_create': (C.c_int, [_P, C.POINTER(LayerT), C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(_P)]),
    'alq_model_destroy': (C.c_int, [_P]),
    'alq_model_num_param_layers': (C.c_int, [_P]),
    'alq_model_max_batch': (C.c_int, [_P]),
    'alq_model_param_sizes': (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'alq_model_layer_out_elems': (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    'alq_model_set_weights': (C.c_int, [_P, C.c_int, _P, _P]),
    'alq_gather_normalize': (C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int, C.POINTER(C.c_int64), _P, C.c_int64,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int, C.c_int, _P]),
    'alq_forward': (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_int]),
    'alq_score_entropy': (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    'alq_topk_work_bytes': (C.c_size_t, [C.c_int64]),
    'alq_topk_uncertain': (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P]),
    'alq_topk_merge': (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P]),
    'alq_fisher': (C.c_int, [_P, _P, C.c_int, _P, C.c_double, _P, _P, _P, _P, _P, _P]),
    'alq_forward_rows': (C.c_int, [_P, _P, _P, C.c_int, _P, _P, _P, C.c_int]),
    'alq_fisher_rows': (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_double, _P, _P, _P, _P, _P, _P]),
    'alq_model_num_params': (C.c_int64, [_P]),
    'alq_forward_dropout': (C.c_int, [_P, _P, C.c_int, C.c_float, C.c_uint64, C.c_int64, C.POINTER(C.c_int32), C.c_int, _P, _P]),
    'alq_param_grads': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_float, C.c_float, C.c_uint64, C.c_int64,
                                  C.POINTER(C.c_int32), C.c_int, C.c_int, _P, _P, _P]),
    'alq_sgd_step': (C.c_int, [_P, _P, _P, C.c_int64, C.c_float]),
    'alq_adam_step': (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64]),
    'alq_sq_accum': (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    'alq_shrink_sum': (C.c_int, [_P, _P, C.c_int, C.c_int64, C.POINTER(C.c_int64), C.c_int, _P]),
    'alq_fisher_classes': (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    'alq_row_norms': (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    'alq_cosine_sims': (C.c_int, [_P, _P, C.c_int64, _P, C.c_int, C.c_int, _P, _P, _P]),
    'alq_colsum_work_bytes': (C.c_size_t, [C.c_int64, C.c_int]),
    'alq_colsum_max': (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P, _P, _P]),
    'alq_take_colmax': (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
    'alq_fold_rowmax': (C.c_int, [_P, _P, C.c_int, C.c_int64, _P]),
    'alq_comm_unique_id': (C.c_int, [_P]),
    'alq_comm_init': (C.c_int, [_P, _P, C.c_int, C.c_int]),
    'alq_comm_destroy': (C.c_int, [_P]),
    'alq_allreduce_sum': (C.c_int, [_P, _P, C.c_int64]),
    'alq_prof_enable': (C.c_int, [_P, C.c_int]),
    'alq_prof_reset': (C.c_int, [_P]),
    'alq_prof_num_classes': (C.c_int, []),
    'alq_prof_class_name': (C.c_char_p, [C.c_int]),
    'alq_prof_read': (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    'alq_model_debug_copy': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.POINTER(C.c_int64)]),
    'alq_debug_set_stamp_buffer': (C.c_int, [_P]),
    'alq_debug_set': (C.c_int, [C.c_int, C.c_int]),
    'alq_model_engine_info': (C.c_int, [_P, C.c_int]),
    'alq_ref64_scores': (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_double, C.c_int, _P, _P, _P, _P, _P, _P]),
    'alq_synth_patches': (C.c_int, [_P, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, _P]),
}

_lib = None


def build(force=False):
    """Compiles csrc/*.hip for gfx950 into libalq.so (hipcc cross-compiles without a GPU)."""
    global _lib
    if os.path.exists(LIB_PATH) and not force:
        srcs = [os.path.join(_HERE, 'csrc', f) for f in os.listdir(os.path.join(_HERE, 'csrc'))
                if f.endswith(('.hip', '.h', '.inc'))] + [os.path.join(os.path.dirname(_HERE), 'include', 'alq.h')]
        if all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
            return LIB_PATH
    subprocess.check_call(['bash', BUILD_SCRIPT])
    _lib = None
    return LIB_PATH


def lib():
    """The loaded library; raises AlqError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AlqError('%s is missing: run `python __graft_entry__.py` (or csrc/build.sh); there is no '
                           'CPU fallback for the device path' % LIB_PATH)
        # libalq.so needs libamdhip64.so.7; PyTorch-ROCm ships its own copy under that SONAME.  Load
        # torch FIRST so that the process holds ONE HIP runtime and torch's device pointers and
        # stream handles are valid inside libalq (loading libalq first would pull in /opt/rocm's
        # copy beside torch's: two runtimes, "no ROCm-capable device" on the second).
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def exported_names():
    return sorted(_SIGNATURES)


def check(rc):
    if rc != 0:
        raise AlqError('libalq error %d: %s' % (rc, lib().alq_last_error().decode()))
