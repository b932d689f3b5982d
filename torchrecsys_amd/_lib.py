# -*- coding: utf-8 -*-
"""ctypes binding of libtrs_hip.so — the C-ABI declared in include/trs.h.

There is NO CPU fallback: a missing library, a missing symbol or a non-gfx950 device raises.  PyTorch-ROCm is used
only as the owner of device memory and of the HIP stream the kernels are enqueued on.
"""
import ctypes as C
import os

TRS_MAX_META = 8
TRS_NET_LINEAR = 0
TRS_NET_FM = 1
LOSS_ID = {"hinge": 0, "bpr": 1}  # TRS_LOSS_HINGE / TRS_LOSS_BPR
ABI_VERSION = 5  # == TRS_ABI_VERSION of include/trs.h (tests/test_abi.py)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtrs_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class TrsTables(C.Structure):
    """struct trs_tables (include/trs.h)."""
    _fields_ = [
        ("user", C.c_void_p),
        ("item", C.c_void_p),
        ("user_lin", C.c_void_p),
        ("item_lin", C.c_void_p),
        ("meta", C.c_void_p * TRS_MAX_META),
        ("meta_lin", C.c_void_p * TRS_MAX_META),
        ("n_users", C.c_int64),
        ("n_items", C.c_int64),
        ("n_meta", C.c_int64 * TRS_MAX_META),
        ("D", C.c_int32),
        ("M", C.c_int32),
    ]


class TrsBatch(C.Structure):
    """struct trs_batch (include/trs.h)."""
    _fields_ = [
        ("user", C.c_void_p),
        ("pos", C.c_void_p),
        ("neg", C.c_void_p),
        ("pos_meta", C.c_void_p),
        ("neg_meta", C.c_void_p),
        ("B", C.c_int64),
        ("idx_bytes", C.c_int32),
        ("err_flag_dev", C.c_void_p),
    ]


class TrsOpt(C.Structure):
    """struct trs_opt (include/trs.h): update rule of the presorted step."""
    _fields_ = [
        ("kind", C.c_int32),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("lr_decay", C.c_float),
        ("step0", C.c_int64),
        ("user_s1", C.c_void_p), ("user_s2", C.c_void_p), ("item_s1", C.c_void_p), ("item_s2", C.c_void_p),
        ("user_lin_s1", C.c_void_p), ("user_lin_s2", C.c_void_p), ("item_lin_s1", C.c_void_p),
        ("item_lin_s2", C.c_void_p),
        ("gacc", C.c_void_p), ("gacc_lin", C.c_void_p), ("cut_rows", C.c_void_p), ("cut_count", C.c_void_p),
        ("cut_capacity", C.c_int32),
        ("meta_s1", C.c_void_p * TRS_MAX_META), ("meta_s2", C.c_void_p * TRS_MAX_META),
        ("meta_lin_s1", C.c_void_p * TRS_MAX_META), ("meta_lin_s2", C.c_void_p * TRS_MAX_META),
        ("meta_gacc", C.c_void_p * TRS_MAX_META), ("meta_gacc_lin", C.c_void_p * TRS_MAX_META),
        ("meta_cut_rows", C.c_void_p * TRS_MAX_META), ("meta_cut_count", C.c_void_p * TRS_MAX_META),
    ]


class TrsMetaStage(C.Structure):
    """struct trs_meta_stage (include/trs.h): staging of metadata scorers on the presorted step."""
    _fields_ = [("item_meta_tab", C.c_void_p), ("xstage", C.c_void_p), ("grad_rows", C.c_void_p),
                ("grad_lin", C.c_void_p), ("meta_ids", C.c_void_p), ("sorted_keys", C.c_void_p * TRS_MAX_META),
                ("sorted_vals", C.c_void_p * TRS_MAX_META), ("lin_scratch", C.c_void_p),
                ("pos_meta_ids", C.c_void_p), ("neg_meta_ids", C.c_void_p)]


class TrsSampler(C.Structure):
    """struct trs_sampler (include/trs.h): sampler options beyond the reference's."""
    _fields_ = [("k_neg", C.c_int32), ("popularity", C.c_int32), ("max_tries", C.c_int32), ("reserved", C.c_int32),
                ("seen_off", C.c_void_p), ("seen_items", C.c_void_p), ("pop_items", C.c_void_p), ("pop_n", C.c_int64),
                ("seen_users", C.c_int64)]


class TrsTrainArgs(C.Structure):
    """struct trs_train_args (include/trs.h): arguments of trs_train_steps_sgd."""
    _fields_ = [("net", C.c_int32), ("n_steps", C.c_int32), ("tables", C.POINTER(TrsTables)), ("batch", C.c_int64),
                ("lr", C.c_float), ("loss", C.c_int32), ("first_stamp", C.c_uint32),
                ("stream_ui_dev", C.c_void_p), ("neg_static_dev", C.c_void_p), ("N", C.c_int64),
                ("shuffle_key", C.c_uint64), ("sample_seed", C.c_uint64), ("first_pos", C.c_int64),
                ("user_buf_dev", C.c_void_p), ("pos_buf_dev", C.c_void_p), ("neg_buf_dev", C.c_void_p),
                ("gz_buf_dev", C.c_void_p), ("du_buf_dev", C.c_void_p), ("loss_sums_dev", C.c_void_p),
                ("err_flag_dev", C.c_void_p), ("scratch_dev", C.c_void_p),
                ("sorted_keys_dev", C.c_void_p), ("sorted_vals_dev", C.c_void_p), ("key_bytes", C.c_int32),
                ("ukey_bytes", C.c_int32), ("user_dup_flags_dev", C.c_void_p), ("item_dup_flags_dev", C.c_void_p),
                ("ustage_buf_dev", C.c_void_p), ("sorted_ukeys_dev", C.c_void_p), ("sorted_uvals_dev", C.c_void_p),
                ("slice_pos0", C.c_int64),
                ("opt", C.POINTER(TrsOpt)), ("meta", C.POINTER(TrsMetaStage)),
                ("events", C.POINTER(C.c_void_p)), ("sync_dev", C.c_void_p),
                ("sync_count_host", C.POINTER(C.c_uint32)), ("n_flagged_dev", C.c_void_p)]


_vp, _i32, _i64, _u64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
_T, _Bp = C.POINTER(TrsTables), C.POINTER(TrsBatch)

# name -> (restype, argtypes); must list every function include/trs.h declares (tests/test_abi.py checks both ways)
PROTOTYPES = {
    "trs_last_error": (C.c_char_p, []),
    "trs_abi_version": (C.c_int, []),
    "trs_check_device": (C.c_int, []),
    "trs_events_create": (C.c_int, [_i32, C.POINTER(C.c_void_p)]),
    "trs_events_destroy": (C.c_int, [_i32, C.POINTER(C.c_void_p)]),
    "trs_events_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "trs_sample_neg": (C.c_int, [_vp, C.c_int, _i64, _i64, _u64, _u64, _vp, _vp]),
    "trs_batch_prepare": (C.c_int, [_vp, _vp, _vp, _i64, _u64, _i64, _i64, _i64, _u64, _u64, _vp, _i32,
                                    _vp, _vp, _vp, _vp, _vp, C.POINTER(TrsSampler), _vp]),
    "trs_score_forward": (C.c_int, [C.c_int, _T, _Bp, _vp, _vp, _vp]),
    "trs_score_fwd_bwd": (C.c_int, [C.c_int, _T, _Bp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "trs_score_backward": (C.c_int, [C.c_int, _T, _Bp, _vp, _vp, _vp, _vp, _vp]),
    "trs_rows_scatter_add": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _i64, _i64, _f, _vp, _vp]),
    "trs_score_sgd_update": (C.c_int, [C.c_int, _T, _Bp, _vp, _vp, _f, _vp]),
    "trs_rows_apply_sparse_adam": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i64, _i32,
                                             _f, _f, _f, _f, _i64, _vp]),
    "trs_rows_apply_adagrad": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _i64, _i32, _f, _f, _vp]),
    "trs_train_scratch_bytes": (C.c_int64, [_i64, _i64, _i64, _i32]),
    "trs_train_steps_sgd": (C.c_int, [C.POINTER(TrsTrainArgs), _vp]),
    "trs_epoch_presort_meta": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i32, _i32, _i64, _vp, _vp, _vp, _i64, _vp,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _vp, _vp, _vp]),
    "trs_epoch_flags": (C.c_int, [_vp, _vp, _i64, _u64, _u64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                                  C.POINTER(TrsSampler), _vp]),
    "trs_epoch_flags_ordered": (C.c_int, [_vp, _vp, _i64, _u64, _u64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _vp, C.POINTER(TrsSampler), _vp]),
    "trs_epoch_user_flags": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "trs_epoch_user_dups_sizes": (C.c_int, [_i64, _i64, _i64, c_int64_p, c_int64_p, c_int64_p]),
    "trs_epoch_user_dups": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_void_p), c_int32_p, _vp]),
    "trs_epoch_presort_sizes": (C.c_int, [_i64, _i64, _i64, c_int64_p, c_int64_p, c_int64_p, c_int64_p]),
    "trs_epoch_presort": (C.c_int, [_vp, _vp, _i64, _u64, _u64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _i64, _vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _vp,
                                    C.POINTER(TrsSampler), _vp]),
    "trs_hinge_auc": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i32, _vp]),
    "trs_hinge_auc_batches": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _i32, _vp]),
    "trs_hinge_backward": (C.c_int, [_vp, _vp, _i64, _f, _vp, _vp, _i32, _vp]),
    "trs_hinge_auc_backward": (C.c_int, [_vp, _vp, _i64, _f, _vp, _vp, _vp, _vp, _i32, _vp]),
    "trs_score_all_items": (C.c_int, [C.c_int, _T, _i64, _i64, _i64, _vp, _vp, _vp]),
    "trs_topk_workspace_bytes": (C.c_int64, [_i64, _i32]),
    "trs_topk": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _i64, _vp]),
    "trs_tuning_set": (C.c_int, [C.c_char_p, _i64, _i32]),
    "trs_mlp_gather_concat": (C.c_int, [_T, _Bp, _i32, _vp, _vp, _i64, _vp]),
    "trs_mlp_gather_gemm1_fwd": (C.c_int, [_T, _Bp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64,
                                           _vp]),
    "trs_mlp_embed_sgd_update_supported": (C.c_int, [_T]),
    "trs_mlp_embed_sgd_update": (C.c_int, [_T, _Bp, _vp, _vp, _i64, _f, _vp, _vp, _vp]),
    "trs_gemm_f32_workspace_bytes": (C.c_int64, [_i64, _i64, _i64]),
    "trs_gemm_f32": (C.c_int, [C.c_int, C.c_int, _i64, _i64, _i64, _f, _vp, _i64, _vp, _i64, _f, _vp, _i64, _vp, _vp,
                               _vp, _i64, _vp]),
    "trs_gemm_bf16": (C.c_int, [C.c_int, C.c_int, _i64, _i64, _i64, _f, _vp, _i64, _vp, _i64, _f, _vp, _i64, _vp, _vp,
                                _vp, _i64, _vp]),
    "trs_gemm_bf16in_workspace_bytes": (C.c_int64, [_i64, _i64, _i64]),
    "trs_gemm_bf16in": (C.c_int, [_i32, _i64, _i64, _i64, _f, _vp, _i64, _vp, _i64, _f, _vp, _vp, _i64, _vp, _vp, _vp,
                                  _i64, _vp]),
    "trs_f32_to_bf16": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "trs_f32_to_bf16_multi": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "trs_bn_stats_finalize": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _f, _vp, _vp, _vp, _vp, _vp]),
    "trs_bn_workspace_floats": (C.c_int64, [_i64, _i32, _i32]),
    "trs_bn_batch_stats": (C.c_int, [_vp, _i64, _i32, _i64, _i32, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "trs_bn_relu_forward": (C.c_int, [_vp, _i32, _i64, _i32, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _f, _vp, _vp,
                                      _i64, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "trs_bn_backward_workspace_floats": (C.c_int64, [_i64, _i32, _i32]),
    "trs_bn_relu_backward": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _f,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    "trs_colsum_workspace_floats": (C.c_int64, [_i64, _i32, _i32]),
    "trs_colsum": (C.c_int, [_vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp]),
    "trs_rowdot": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "trs_outer": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _vp]),
}


class TrsError(RuntimeError):
    """A negative return code from libtrs_hip.so."""


_lib = None


def load():
    """Load libtrs_hip.so and bind every prototype.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            f"`make -C torchrecsys_amd/csrc`.  torchrecsys_amd has no CPU fallback.")
    # PyTorch first: it ships its own HIP runtime, and the process must end up with ONE — the library's libamdhip64
    # dependency then resolves to the copy torch has loaded.  (Loaded the other way round — this library before the first
    # `import torch`, e.g. build() followed by smoke() in one process — the library's runtime saw no device on the GPU box.)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.trs_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libtrs_hip.so ABI version {got} != binding version {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().trs_last_error()
        raise TrsError(f"{what or 'libtrs_hip'} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


class tuning:
    """Context manager over trs_tuning_set: `with _lib.tuning(GEMM16_TILE=256): ...` sets the library's knobs (names as
    in csrc/trs_common.h TrsTuning, without the TRS_ prefix) and restores their defaults on exit.  The library reads the
    TRS_* environment only once, at its first use, so tests and tools switch kernels through this."""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        lib = load()
        for k, v in self.knobs.items():
            check(lib.trs_tuning_set(k.encode(), int(v), 0), "trs_tuning_set")
        return self

    def __exit__(self, *exc):
        lib = load()
        for k in self.knobs:
            check(lib.trs_tuning_set(k.encode(), 0, 1), "trs_tuning_set")
        return False
