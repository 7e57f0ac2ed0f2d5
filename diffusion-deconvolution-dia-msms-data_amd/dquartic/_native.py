"""ctypes binding of libdq_hip.so (include/dq_hip.h).  There is NO fallback: if the library is missing or a call
fails, a RuntimeError is raised -- the product path never routes through PyTorch ops or the test oracle."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DQ_HIP_LIB", os.path.join(os.path.dirname(_HERE), "libdq_hip.so"))

_lib = None
ABI_VERSION = 10  # DQ_ABI_VERSION of include/dq_hip.h this table was written against
PRED_TYPES = {"eps": 0, "x0": 1}  # DQ_PRED_EPS / DQ_PRED_X0
PRECISIONS = {"fp32": 0, "bf16x3": 1}  # DQ_PRECISION_FP32 / DQ_PRECISION_BF16X3

# name -> (restype, argtypes); this table is checked against include/dq_hip.h by tests/test_abi.py
PROTOTYPES = {
    "dq_last_error": (c_char_p, []),
    "dq_abi_version": (c_int, []),
    "dq_plan_create": (c_void_p, [c_int, c_int, POINTER(c_int), c_int, c_int]),
    "dq_plan_destroy": (None, [c_void_p]),
    "dq_plan_num_params": (c_int, [c_void_p]),
    "dq_plan_param_floats": (c_int64, [c_void_p]),
    "dq_plan_param_info": (c_int, [c_void_p, c_int, c_char_p, c_int, POINTER(c_int64), POINTER(c_int), POINTER(c_int64)]),
    "dq_unet_workspace_bytes": (c_int64, [c_void_p, c_int, c_int, c_int]),
    "dq_q_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "dq_ddim_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dq_ddim_step_x0": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dq_unet_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float,
                            c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "dq_unet_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_int64, c_int, c_int, c_void_p]),
    "dq_mse_loss_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dq_mse_loss_weighted_fwd_bwd": (c_int, [c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int, c_int64, c_void_p]),
    "dq_adamw_clip_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_float, c_void_p,
                                       c_double, c_double, c_double, c_double, c_void_p, c_void_p, c_void_p]),
    "dq_plan_set_side_stream": (c_int, [c_void_p, c_int]),
    "dq_set_option": (c_int, [c_char_p, c_int64]),
    "dq_get_option": (c_int64, [c_char_p]),
    "dq_adamw_clip_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_float, c_double,
                                   c_double, c_double, c_double, c_double, c_int, c_void_p, c_void_p]),
    "dq_train_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                              c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "dq_ms1_loss_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                    c_void_p, c_int, c_int, c_int, c_void_p]),
    "dq_ddim_sample": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_float), c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                               POINTER(c_int32), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int,
                               c_void_p]),
    "dq_pair_batch_scratch_bytes": (c_int64, [c_int]),
    "dq_pair_batch": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int64, c_float, c_float, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "dq_debug_tensor_offset": (c_int64, [c_void_p, c_char_p]),
    "dq_debug_side_tail_store": (c_int, [c_void_p, c_void_p, c_float, c_int]),
    "dq_tfm_create": (c_void_p, [c_int, c_int, c_int, c_int]),
    "dq_tfm_destroy": (None, [c_void_p]),
    "dq_tfm_num_params": (c_int, [c_void_p]),
    "dq_tfm_param_floats": (c_int64, [c_void_p]),
    "dq_tfm_param_info": (c_int, [c_void_p, c_int, c_char_p, c_int, POINTER(c_int64), POINTER(c_int), POINTER(c_int64)]),
    "dq_tfm_workspace_bytes": (c_int64, [c_void_p, c_int, c_int, c_int, c_int]),
    "dq_tfm_fwd": (c_int, [c_void_p] * 9 + [c_int, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "dq_tfm_bwd": (c_int, [c_void_p] * 8 + [c_int] + [c_void_p] * 3 + [c_int64, c_int, c_int, c_int, c_void_p]),
    "dq_gemm_scratch_floats": (c_int64, [c_int, c_int, c_int]),
    "dq_gemm": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int64,
                        c_void_p]),
    "dq_gemm_bf16x3": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int64,
                               c_void_p]),
    "dq_tfm_bwd_buckets": (c_int, [c_void_p] * 8 + [c_int] + [c_void_p] * 3 + [c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dq_tfm_num_buckets": (c_int, [c_void_p]),
    "dq_tfm_bucket_info": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "dq_tfm_set_precision": (c_int, [c_void_p, c_int]),
    "dq_linattn_fwd": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_void_p]),
    "dq_linattn_bwd": (c_int, [c_void_p] * 15 + [c_int, c_int, c_int, c_void_p]),
    "dq_linattn_prep_floats": (c_int64, []),
    "dq_linattn_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "dq_linattn_fwd_prepared": (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_void_p]),
    "dq_rmsnorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dq_time_mlp_fwd": (c_int, [c_void_p] * 8 + [c_int, c_void_p]),
    "dq_scale_shift_fwd": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "dq_prep_inputs_fwd": (c_int, [c_void_p] * 4 + [c_float, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dq_conv_fwd": (c_int, [c_void_p] * 4 + [c_int, c_void_p] + [c_int] * 7 + [c_void_p]),
    "dq_resblock_workspace_floats": (c_int64, [c_int] * 5),
    "dq_level_param_floats": (c_int64, [c_int] * 5),
    "dq_level_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p, c_int64, c_void_p]),
    "dq_resblock_dout_offset": (c_int64, [c_int] * 5),
    "dq_resblock_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p, c_int64, c_void_p]),
    "dq_resblock_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 5 + [c_int] * 4 + [c_void_p, c_int64, c_void_p]),
    "dq_rope": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_void_p]),
    "dq_attn_fwd": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "dq_attn_bwd": (c_int, [c_void_p] * 10 + [c_int, c_int, c_void_p]),
}


def lib():
    """Load (once) and return the shared library; raises RuntimeError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libdq_hip.so not found at {LIB_PATH}: build it with `make -C diffusion-deconvolution-dia-msms-data_amd` "
                "(or python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
        import torch  # noqa: F401  (PyTorch's HIP runtime first: the library's libamdhip64 dependency must resolve to that same copy)

        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.dq_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {L.dq_abi_version()}, this package binds version {ABI_VERSION}: rebuild it")
        _lib = L
    return _lib


def build_id() -> str:
    """Identity of the native sources this tree was built from (sha256 over csrc/, include/dq_hip.h and the Makefile, 16 hex
    digits).  Profiles under profiles/ record it, and bench.py only quotes a counter-derived figure whose build id matches."""
    import hashlib

    root = os.path.dirname(_HERE)
    files = sorted(os.path.join(root, "csrc", f) for f in os.listdir(os.path.join(root, "csrc")) if f.endswith((".hip", ".h", ".cpp")))
    files += [os.path.join(root, "Makefile"), os.path.join(os.path.dirname(root), "include", "dq_hip.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def set_option(key: str, value: int) -> None:
    """``dq_set_option``: a process-wide tuning option of the library (keys: include/dq_hip.h)."""
    check(lib().dq_set_option(key.encode(), int(value)), f"dq_set_option({key!r})")


def get_option(key: str) -> int:
    return int(lib().dq_get_option(key.encode()))


def check(rc, what):
    if rc != 0:
        msg = lib().dq_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device/host pointer of a contiguous torch tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_ptr():
    import torch

    return c_void_p(torch.cuda.current_stream().cuda_stream)
