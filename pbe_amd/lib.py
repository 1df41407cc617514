"""ctypes binding of libpbe_hip.so (include/pbe_hip.h).  No CPU fallback: if the shared library
is missing or a symbol is absent this module raises — the product path must fail loudly."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBE_LIB_PATH") or os.path.join(_HERE, "libpbe_hip.so")     # PBE_LIB_PATH: diagnostic builds (tools/) only
ABI_VERSION = 8

c_i32, c_i64, c_f32, c_vp, c_sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class GemmDesc(C.Structure):
    _fields_ = [("A", c_vp), ("A2", c_vp), ("W", c_vp), ("C", c_vp), ("bias", c_vp), ("rowvec", c_vp), ("resid", c_vp),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("K1", c_i32),
                ("lda", c_i64), ("lda2", c_i64), ("ldw", c_i64), ("ldc", c_i64), ("ldr", c_i64),
                ("ldv", c_i32), ("group_rows", c_i32),
                ("strideA", c_i64), ("strideW", c_i64), ("strideC", c_i64), ("strideR", c_i64),
                ("batch", c_i32), ("alpha", c_f32), ("act", c_i32), ("bias_per_row", c_i32),
                ("workspace", c_vp), ("workspace_bytes", c_sz), ("tile_cfg", c_i32),
                ("a_scale", c_vp), ("w_scale", c_vp), ("a_scale_stride", c_i64), ("w_scale_stride", c_i64), ("operand_dtype", c_i32),
                ("alpha_cols", c_i32), ("ln_stats", c_vp), ("ln_parts", c_i32), ("ln_stats_ld", c_i64), ("ln_colsum", c_vp), ("ln_eps", c_f32),
                ("row_stats_out", c_vp), ("row_stats_ld", c_i64), ("VT", c_vp), ("vt_col0", c_i32), ("vt_tokens", c_i32), ("vt_bs", c_i64), ("vt_rs", c_i64)]


class Conv3x3Desc(C.Structure):
    _fields_ = [("X", c_vp), ("X2", c_vp), ("Wp", c_vp), ("Y", c_vp), ("bias", c_vp), ("rowvec", c_vp), ("resid", c_vp),
                ("B", c_i32), ("H", c_i32), ("W", c_i32), ("C1", c_i32), ("C2", c_i32), ("Cout", c_i32),
                ("stride", c_i32), ("pad", c_i32), ("upsample", c_i32), ("ldv", c_i32), ("act", c_i32),
                ("workspace", c_vp), ("workspace_bytes", c_sz), ("tile_cfg", c_i32), ("kblock", c_i32),
                ("group_stats_out", c_vp), ("group_stats_groups", c_i32), ("group_stats_blocks", c_vp)]


class AttnDesc(C.Structure):
    _fields_ = [("Q", c_vp), ("K", c_vp), ("VT", c_vp), ("O", c_vp),
                ("B", c_i32), ("H", c_i32), ("Nq", c_i32), ("Nk", c_i32), ("D", c_i32),
                ("q_bs", c_i64), ("q_rs", c_i64), ("k_bs", c_i64), ("k_rs", c_i64),
                ("vt_bs", c_i64), ("vt_rs", c_i64), ("o_bs", c_i64), ("o_rs", c_i64), ("scale", c_f32), ("q_prescaled", c_i32)]


# name -> (restype, argtypes): every symbol include/pbe_hip.h declares
SYMBOLS = {
    "pbe_abi_version": (c_i32, []),
    "pbe_last_error": (C.c_char_p, []),
    "pbe_source_hash": (C.c_char_p, []),
    "pbe_sizeof_gemm_desc": (c_sz, []),
    "pbe_sizeof_conv3x3_desc": (c_sz, []),
    "pbe_sizeof_attn_desc": (c_sz, []),
    "pbe_gemm_f16": (c_i32, [C.POINTER(GemmDesc), c_vp]),
    "pbe_conv3x3_f16": (c_i32, [C.POINTER(Conv3x3Desc), c_vp]),
    "pbe_gemm_plan": (c_i32, [C.POINTER(GemmDesc), C.POINTER(c_i32), C.POINTER(c_sz)]),
    "pbe_conv3x3_plan": (c_i32, [C.POINTER(Conv3x3Desc), C.POINTER(c_i32), C.POINTER(c_sz)]),
    "pbe_im2col3x3_f16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "pbe_groupnorm_workspace_bytes": (c_sz, [c_i32, c_i32]),
    "pbe_groupnorm_f16": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp, c_sz, c_vp]),
    "pbe_groupnorm_apply_f16": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "pbe_layernorm_f16": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_i64, c_f32, c_vp]),
    "pbe_row_stats_f16": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i64, c_vp]),
    "pbe_layernorm_f8": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_i64, c_f32, c_vp]),
    "pbe_attention_f16": (c_i32, [C.POINTER(AttnDesc), c_vp]),
    "pbe_softmax_rows_f16": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i64, c_i64, c_f32, c_vp]),
    "pbe_geglu_f16": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    "pbe_timestep_embedding_f16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_f32, c_vp]),
    "pbe_nchw_f32_to_nhwc_f16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "pbe_nhwc_f16_to_nchw_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "pbe_plms_pack_input": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "pbe_plms_update": (c_i32, [c_vp, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, C.POINTER(c_f32), c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "pbe_axpy_f32": (c_i32, [c_vp, c_f32, c_vp, c_i64, c_vp]),
    "pbe_qsample_blend_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "pbe_posterior_sample": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp]),
    "pbe_scale_latent_f16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "pbe_clip_patchify_f16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "pbe_bcast_row_f16": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp]),
    "pbe_image_post_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "pbe_resize_bilinear_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "pbe_u8_to_planes_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, C.POINTER(c_f32), C.POINTER(c_f32), c_i32, c_vp]),
    "pbe_mul_planes_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "pbe_planes_to_u8_canvas": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, C.POINTER(c_f32), C.POINTER(c_f32), c_i32, c_vp]),
    "pbe_tune": (c_i32, [c_i32, c_i32]),
    "pbe_prof_enable": (c_i32, [c_i32]),
    "pbe_prof_reset": (c_i32, []),
    "pbe_prof_collect": (c_i32, [C.POINTER(C.c_double), c_i32]),
    "pbe_prof_class_name": (C.c_char_p, [c_i32]),
}

_lib = None
_lock = threading.Lock()

SOURCES = ["runtime.hip", "igemm.hip", "igemm_dense.hip", "igemm_conv.hip", "igemm_halo.hip", "igemm_f8.hip", "igemm_ex.hip", "igemm_ex_ln.hip", "igemm_ex_st.hip", "igemm_ex_qkv.hip", "igemm_ex_all.hip", "igemm_astat.hip", "attention.hip", "norm.hip",
           "elementwise.hip"]
HASHED = [os.path.join("csrc", f) for f in SOURCES] + [os.path.join("csrc", "common.h"), os.path.join("csrc", "igemm_kernel.h"), os.path.join("..", "include", "pbe_hip.h"), "build.py"]


def source_hash() -> str:
    """Identity of what libpbe_hip.so must have been built from: every kernel source, the shared headers and build.py
    (which holds the compiler flags).  build.py embeds it (-DPBE_SRC_HASH); load() compares."""
    import hashlib
    h = hashlib.sha256()
    for rel in HASHED:
        with open(os.path.join(_HERE, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


class PbeError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load (once) and type every entry point.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch bundles its own libamdhip64; import it FIRST so libpbe_hip.so binds to the HIP runtime
        # that owns torch's device context and streams (loading ours first leaves two runtimes in
        # the process and launches fail with "no ROCm-capable device").
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise PbeError(f"{LIB_PATH} not found: build it with `python -m pbe_amd.build` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise PbeError(f"libpbe_hip.so does not export {name}") from e
            fn.restype, fn.argtypes = res, args
        v = lib.pbe_abi_version()
        if v != ABI_VERSION:
            raise PbeError(f"libpbe_hip.so ABI version {v} != expected {ABI_VERSION}")
        for cls, fn in ((GemmDesc, lib.pbe_sizeof_gemm_desc), (Conv3x3Desc, lib.pbe_sizeof_conv3x3_desc), (AttnDesc, lib.pbe_sizeof_attn_desc)):
            if C.sizeof(cls) != fn():
                raise PbeError(f"{cls.__name__}: ctypes layout is {C.sizeof(cls)} bytes, libpbe_hip.so was compiled with {fn()}")
        built, want = lib.pbe_source_hash().decode(), source_hash()
        if built != want and not os.environ.get("PBE_LIB_PATH"):
            raise PbeError(f"libpbe_hip.so was built from other sources (binary {built}, tree {want}): run `python -m pbe_amd.build`")
        _lib = lib
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().pbe_last_error().decode("utf-8", "replace")
        raise PbeError(f"{what} failed with code {code}: {msg}")
