"""ctypes binding of libdyffusion_hip.so (C ABI: include/dyffusion_hip.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950).  There is deliberately NO
fallback: if the shared object is missing or a symbol cannot be resolved, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DYF_LIB") or os.path.join(_HERE, "lib", "libdyffusion_hip.so")  # DYF_LIB: experiment builds
# the fp16 build of the same sources (-DDYF_F16=1): same ABI, fp16 storage + v_mfma_*_f16 (BASELINE configs[4])
LIB_PATH_F16 = os.environ.get("DYF_LIB_F16") or os.path.join(_HERE, "lib", "libdyffusion_hip_f16.so")
DTYPES = {"bf16": 0, "bfloat16": 0, "fp16": 1, "float16": 1, "half": 1}

DYF_ABI_VERSION = 8
DYF_OK, DYF_ERR_INVALID_ARGUMENT, DYF_ERR_UNSUPPORTED, DYF_ERR_HIP, DYF_ERR_STATE = range(5)
NET_FORECASTER, NET_INTERPOLATOR = 0, 1
ARCH_UNET_SIMPLE, ARCH_UNET_RESNET = 0, 1
ARCH_SIMPLE_CONV_NET = 2
FCOND = {"none": 0, "data": 1, "data+noise": 2}


class NetConfig(C.Structure):
    _fields_ = [("arch", C.c_int32), ("in_channels", C.c_int32), ("cond_channels", C.c_int32),
                ("out_channels", C.c_int32), ("dim", C.c_int32), ("with_time_emb", C.c_int32),
                ("upsample_h", C.c_int32), ("upsample_w", C.c_int32), ("dropout", C.c_float),
                ("input_dropout", C.c_float), ("n_mults", C.c_int32), ("dim_mults", C.c_int32 * 6),
                ("block_dropout1", C.c_float), ("attn_dropout", C.c_float), ("groups", C.c_int32),
                ("init_kernel_size", C.c_int32), ("init_padding", C.c_int32), ("outer_nearest", C.c_int32),
                ("keep_spatial_dims", C.c_int32), ("single_conv_layer", C.c_int32), ("learned_sinusoidal_dim", C.c_int32)]


class EngineConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("max_batch", C.c_int32), ("use_graph", C.c_int32), ("enable_mfma", C.c_int32), ("dtype", C.c_int32),
                ("batch_invariant", C.c_int32), ("net", NetConfig * 2)]


class PlanStep(C.Structure):
    _fields_ = [("forecaster_time", C.c_float), ("tau", C.c_float), ("i_next", C.c_float), ("i_cur", C.c_float),
                ("is_last", C.c_int32), ("out_slot", C.c_int32)]


class Plan(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("steps", C.POINTER(PlanStep)), ("sampling_cold", C.c_int32),
                ("cold_for_last_step", C.c_int32), ("forward_conditioning", C.c_int32), ("n_refine", C.c_int32),
                ("refine_times", C.POINTER(C.c_float)), ("refine_slots", C.POINTER(C.c_int32)),
                ("n_out_slots", C.c_int32), ("interpolator_dropout", C.c_int32), ("forecaster_dropout", C.c_int32)]


class BcArgs(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_fields", C.c_int32), ("rows", C.c_int32), ("channels", C.c_int32),
                ("height", C.c_int32), ("width", C.c_int32), ("n_meta", C.c_int32), ("row_meta_dev", C.c_void_p),
                ("time_factor_dev", C.c_void_p), ("times_per_meta", C.c_int32), ("fixed_mask_dev", C.c_void_p),
                ("in_velocity_dev", C.c_void_p), ("vertex_y_dev", C.c_void_p), ("boundary_dev", C.c_void_p)]


BC_NAVIER_STOKES, BC_SPRING_MESH = 0, 1
COMM_ID_BYTES = 128  # DYF_COMM_ID_BYTES (ncclUniqueId)
TRAIN_BATCH_STATS, TRAIN_DROPOUT = 1, 2

# every symbol include/dyffusion_hip.h and include/dyffusion_hip_testing.h declare: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("dyf_engine_create", C.c_int, [C.POINTER(EngineConfig), C.POINTER(_P)]),
    ("dyf_engine_destroy", None, [_P]),
    ("dyf_last_error", C.c_char_p, [_P]),
    ("dyf_abi_version", C.c_int32, []),
    ("dyf_dtype", C.c_int32, []),
    ("dyf_load_weights", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(_P),
                                   C.POINTER(C.c_int32)]),
    ("dyf_train_load_weights", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(_P),
                                         C.POINTER(C.c_int32)]),
    ("dyf_train_load_weights_dev", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(_P),
                                             C.POINTER(C.c_int32)]),
    ("dyf_train_export_dev", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(_P)]),
    ("dyf_net_forward", C.c_int, [_P, C.c_int32, _P, _P, _P, _P, C.c_int32, C.c_int32, C.POINTER(_P), _P]),
    ("dyf_set_plan", C.c_int, [_P, C.POINTER(Plan)]),
    ("dyf_sample", C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(_P), _P, _P]),
    ("dyf_seed", C.c_int, [_P, C.c_uint64]),
    ("dyf_set_row_offset", C.c_int, [_P, C.c_uint32]),
    ("dyf_set_log_intermediates", C.c_int, [_P, C.c_int32]),
    ("dyf_get_log", C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    ("dyf_set_row_groups", C.c_int, [_P, C.c_int32]),
    ("dyf_row_groups", C.c_int32, [_P]),
    ("dyf_comm_unique_id", C.c_int, [_P]),
    ("dyf_comm_init", C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    ("dyf_comm_destroy", C.c_int, [_P]),
    ("dyf_comm_count", C.c_int, [_P, _P]),
    ("dyf_sample_gather", C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    ("dyf_get_sampler_state", C.c_int, [_P, C.c_int32, _P, C.c_int32, _P]),
    ("dyf_plan_forward_counts", C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("dyf_net_flops", C.c_int, [_P, C.c_int32, C.POINTER(C.c_double)]),
    ("dyf_net_flops_executed", C.c_int, [_P, C.c_int32, C.POINTER(C.c_double)]),
    ("dyf_time_conv_layer", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("dyf_ensemble_metrics", C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.POINTER(C.c_double), _P]),
    ("dyf_time_layer_in_rollout", C.c_int, [_P, C.c_int32, C.c_int32, _P, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    ("dyf_time_kernel_in_rollout", C.c_int, [_P, C.c_int32, C.c_int32, _P, C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("dyf_op_conv2d", C.c_int, [_P, _P, _P] + [C.c_int32] * 9 + [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    ("dyf_op_upconv2d", C.c_int, [_P, _P, _P] + [C.c_int32] * 5 + [_P, _P, C.c_int32, _P, _P]),
    ("dyf_op_linear_attention", C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    ("dyf_op_linear_attention_fused", C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    ("dyf_op_attention", C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    ("dyf_op_attention_dropout", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_float, _P, _P]),
    ("dyf_criterion", C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _P, _P]),
    ("dyf_train_forward", C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    ("dyf_train_backward", C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P]),
    ("dyf_train_zero_grads", C.c_int, [_P, C.c_int32]),
    ("dyf_train_set_precision", C.c_int, [_P, C.c_int32]),
    ("dyf_train_precision", C.c_int32, [_P]),
    ("dyf_train_export", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(_P)]),
    ("dyf_criterion_grad", C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_float, _P, _P]),
    ("dyf_train_conv_check", C.c_int, [_P] + [C.c_int32] * 9 + [C.c_uint32, C.POINTER(C.c_float)]),
    ("dyf_apply_boundary_conditions", C.c_int, [_P, C.POINTER(BcArgs), _P, _P]),
    ("dyf_debug_read_block_output", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("dyf_poll_errors", C.c_int, [_P, C.c_int32]),
    ("dyf_gn_fuse_state", C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("dyf_debug_gn_fuse", C.c_int, [_P, C.c_uint32, C.c_int32]),
    ("dyf_debug_form_log", None, [C.c_int32]),
    ("dyf_debug_form_log_read", C.c_int32, [C.c_char_p, C.c_int32]),
    ("dyf_time_named_kernel_in_rollout", C.c_int, [_P, C.c_char_p, C.c_int32, _P, C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                                  C.POINTER(C.c_double)]),
    ("dyf_debug_set_form", None, [C.c_char_p, C.c_char_p]),
    ("dyf_debug_forms", C.c_int32, [C.c_char_p, C.c_int32]),
]


def load_library(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the DYffusion HIP engine has not been built.  Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` from the repository root (needs hipcc).  There is no CPU fallback for the product path.")
    lib = C.CDLL(path)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.dyf_abi_version() != DYF_ABI_VERSION:
        raise ImportError(f"ABI mismatch: library {lib.dyf_abi_version()} vs binding {DYF_ABI_VERSION}")
    return lib


_LIBS = {}


def lib(dtype: str = "bf16") -> C.CDLL:
    """The C-ABI library for `dtype`: "bf16" -> libdyffusion_hip.so, "fp16" -> libdyffusion_hip_f16.so."""
    code = DTYPES[dtype]
    if code not in _LIBS:
        l = load_library(LIB_PATH_F16 if code else LIB_PATH)
        if l.dyf_dtype() != code:
            raise ImportError(f"library for dtype {dtype} reports dyf_dtype() = {l.dyf_dtype()}")
        for k, v in _FORMS.items():  # switches set before this build of the library was loaded
            l.dyf_debug_set_form(k.encode(), v.encode())
        _LIBS[code] = l
    return _LIBS[code]


_FORMS = {}


def set_form(key=None, value=None) -> None:
    """Test / tool seam (include/dyffusion_hip_testing.h dyf_debug_set_form): set kernel-form switch `key` to `value` in every loaded
    build of the library (and in builds loaded later); value None removes the key, key None removes all.  The product never calls
    this, and the library reads no such switch from the environment."""
    if key is None:
        _FORMS.clear()
    elif value is None:
        _FORMS.pop(key, None)
    else:
        _FORMS[key] = str(value)
    for l in _LIBS.values():
        l.dyf_debug_set_form(None if key is None else key.encode(), None if value is None else str(value).encode())


def forms() -> dict:
    return dict(_FORMS)
