"""ctypes binding of libkai0hip.so — the C-ABI boundary (include/kai0hip.h).

The product path has NO fallback: if the shared library is missing, or a call is attempted without a GPU,
this module raises.  (The CPU restatement under oracle/ is test infrastructure and is never imported here.)
"""

from __future__ import annotations

import ctypes as C
import os
import pathlib

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libkai0hip.so"

c_p = C.c_void_p
c_i = C.c_int
c_i64 = C.c_int64
c_f = C.c_float


class GemmSeg(C.Structure):
    _fields_ = [("dst", c_p), ("ld", c_i64), ("n_begin", C.c_int32), ("_pad", C.c_int32)]


class GemmDesc(C.Structure):
    """Mirror of `kai0_gemm_desc` (include/kai0hip.h)."""

    _fields_ = [
        ("A", c_p), ("B", c_p), ("C", c_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("a_kc", C.c_int32), ("b_kc", C.c_int32),
        ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64),
        ("batch", C.c_int32), ("batch_inner", C.c_int32),
        ("sA1", c_i64), ("sA2", c_i64), ("sB1", c_i64), ("sB2", c_i64), ("sC1", c_i64), ("sC2", c_i64),
        ("a_rpb", C.c_int32), ("b_rpb", C.c_int32), ("c_rpb", C.c_int32), ("_pad0", C.c_int32),
        ("a_bs", c_i64), ("a_off", c_i64), ("b_bs", c_i64), ("b_off", c_i64), ("c_bs", c_i64), ("c_off", c_i64),
        ("bias", c_p), ("bias_f32", C.c_int32), ("scale", c_f),
        ("act", C.c_int32), ("out_f32", C.c_int32),
        ("pre_out", c_p), ("gate", c_p), ("gate_rpb", C.c_int32), ("accumulate", C.c_int32),
        ("gate_ld", c_i64), ("residual", c_p), ("ldr", c_i64), ("sR1", c_i64), ("sR2", c_i64),
        ("split_k", C.c_int32), ("c_nontemporal", C.c_int32), ("workspace", c_p), ("workspace_bytes", c_i64),
        ("aux1", c_p), ("aux2", c_p),
        ("nseg", C.c_int32), ("_pad2", C.c_int32), ("seg", GemmSeg * 3),
        ("rowvec", c_p), ("rv_s1", c_i64), ("rv_s2", c_i64), ("rv_ld", c_i64),
        ("B2", c_p), ("pre_out2", c_p),
        ("norm_out", c_p), ("norm_w", c_p), ("norm_b", c_p), ("norm_eps", C.c_float), ("norm_kind", C.c_int32),
        ("rope_cos", c_p), ("rope_sin", c_p), ("rope_half", C.c_int32), ("rope_n_end", C.c_int32),
        ("tile_cfg", C.c_int32), ("persist", C.c_int32), ("general_epilogue", C.c_int32), ("small_w8", C.c_int32),
    ]  # fmt: skip


class AttnDesc(C.Structure):
    """Mirror of `kai0_attn_desc` (include/kai0hip.h)."""

    _fields_ = [
        ("Q", c_p), ("K", c_p), ("V", c_p), ("O", c_p), ("P", c_p), ("qcode", c_p), ("kcode", c_p),
        ("rows", C.c_int32), ("Sk", C.c_int32), ("HD", C.c_int32), ("H", C.c_int32), ("q0", C.c_int32),
        ("batch", C.c_int32), ("batch_inner", C.c_int32), ("_pad0", C.c_int32),
        ("ldq", c_i64), ("ldk", c_i64), ("ldv", c_i64), ("ldo", c_i64), ("ldp", c_i64),
        ("sQ1", c_i64), ("sQ2", c_i64), ("sK1", c_i64), ("sK2", c_i64), ("sV1", c_i64), ("sV2", c_i64),
        ("sO1", c_i64), ("sO2", c_i64), ("sP", c_i64), ("qcode_ld", c_i64), ("kcode_ld", c_i64),
        ("scale", c_f), ("online", C.c_int32), ("lse", c_p), ("s_lse", c_i64),
    ]  # fmt: skip


class AttnBwdDesc(C.Structure):
    """Mirror of `kai0_attn_bwd_desc` (include/kai0hip.h)."""

    _fields_ = [
        ("dO", c_p), ("O", c_p), ("Q", c_p), ("K", c_p), ("V", c_p), ("lse", c_p), ("qcode", c_p), ("kcode", c_p),
        ("P", c_p), ("dS", c_p), ("dQ", c_p),
        ("batch", C.c_int32), ("rows", C.c_int32), ("Sk", C.c_int32), ("HD", C.c_int32), ("H", C.c_int32), ("_pad0", C.c_int32),
        ("ldo", c_i64), ("ldk", c_i64), ("ldv", c_i64), ("ldp", c_i64), ("sO", c_i64), ("sK", c_i64), ("sV", c_i64), ("sP", c_i64),
        ("s_lse", c_i64), ("qcode_ld", c_i64), ("kcode_ld", c_i64),
        ("scale", c_f), ("_pad1", C.c_int32),
    ]  # fmt: skip


class SkinnySeg(C.Structure):
    _fields_ = [("dst", c_p), ("ld", c_i64), ("n_begin", C.c_int32), ("n_end", C.c_int32), ("rope", C.c_int32),
                ("_pad", C.c_int32)]  # fmt: skip


class ReduceItem(C.Structure):
    """Mirror of `kai0_reduce_item` (include/kai0hip.h)."""

    _fields_ = [("partial", c_p), ("out", c_p), ("ld", c_i64), ("blocks", C.c_int32), ("ncols", C.c_int32),
                ("out_f32", C.c_int32), ("_pad", C.c_int32)]  # fmt: skip


class PackPart(C.Structure):
    """Mirror of `kai0_pack_part` (include/kai0hip.h)."""

    _fields_ = [("src", c_p), ("dst", c_p), ("src_bs", c_i64), ("src_ld", c_i64), ("dst_bs", c_i64), ("dst_ld", c_i64),
                ("B", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("mode", C.c_int32), ("pos_off", C.c_int32),
                ("_pad", C.c_int32)]  # fmt: skip


class SkinnyDesc(C.Structure):
    """Mirror of `kai0_skinny_desc` (include/kai0hip.h)."""

    _fields_ = [
        ("A", c_p), ("W", c_p), ("lda", c_i64), ("ldw", c_i64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("pair_stride", C.c_int32), ("mode", C.c_int32), ("split_k", C.c_int32),
        ("a_rpb", C.c_int32), ("c_rpb", C.c_int32),
        ("a_bs", c_i64), ("a_off", c_i64), ("c_bs", c_i64), ("c_off", c_i64),
        ("seg", SkinnySeg * 3), ("nseg", C.c_int32), ("gate_rpb", C.c_int32),
        ("gate", c_p), ("gate_ld", c_i64), ("residual", c_p), ("ldr", c_i64),
        ("rope_cos", c_p), ("rope_sin", c_p), ("rope_half", C.c_int32), ("w_packed", C.c_int32),
        ("workspace", c_p), ("workspace_bytes", c_i64),
        ("mod", c_p), ("mod_ld", c_i64), ("mod_rpb", C.c_int32), ("eps", c_f),
        ("rowsq_in", c_p), ("cvec", c_p), ("rowsq_parts", C.c_int32), ("_pad2", C.c_int32), ("rowsq_ld", c_i64),
        ("rowsq_out", c_p), ("rowsq_out_ld", c_i64),
    ]  # fmt: skip


# name -> argtypes (every function returns int except where noted)
_PROTOS: dict[str, list] = {
    "kai0_abi_version": [],
    "kai0_gemm_desc_size": [],
    "kai0_device_info": [c_i, C.POINTER(c_i), C.POINTER(c_i), C.c_char_p],
    "kai0_gemm_bf16": [C.POINTER(GemmDesc), c_p],
    "kai0_attn_fwd": [C.POINTER(AttnDesc), c_p],
    "kai0_attn_desc_size": [],
    "kai0_attn_combine": [c_p, c_p, c_p, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_p],
    "kai0_attn_bwd_dq2": [C.POINTER(AttnBwdDesc), c_p],
    "kai0_attn_bwd_desc_size": [],
    "kai0_siglip_attn_fwd": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_p],
    "kai0_siglip_attn_bwd2": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i64, c_f, c_p],
    "kai0_gemm_skinny_bf16": [C.POINTER(SkinnyDesc), c_p],
    "kai0_skinny_desc_size": [],
    "kai0_attn_decode": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i, c_i64, c_i64,
                         c_i64, c_i64, c_f, c_p, c_i64, c_p],
    "kai0_attn_bwd_dq": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                         c_f, c_p],
    "kai0_transpose_strided_bf16": [c_p, c_p, c_i, c_i, c_i64, c_i64, c_i, c_i64, c_i64, c_p],
    "kai0_rope_table": [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p],
    "kai0_prefix_codes": [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "kai0_gemm_f32": [c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i64, c_p],
    "kai0_linear_rows_f32": [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p],
    "kai0_rmsnorm_fwd": [c_p, c_p, c_p, c_p, c_i64, c_i, c_f, c_p],
    "kai0_rmsnorm_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i64, c_i, c_p],
    "kai0_adarms_fwd": [c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_p],
    "kai0_adarms_combine": [c_p, c_i, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_f, c_p],
    "kai0_adarms_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p],
    "kai0_layernorm_fwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_f, c_p],
    "kai0_layernorm_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i64, c_i, c_p],
    "kai0_reduce_partials": [c_p, c_i, c_i, c_i64, c_p, c_i, c_p],
    "kai0_colsum_bf16": [c_p, c_i64, c_i, c_i64, c_p, c_i, c_p, c_i, c_p],
    "kai0_reduce_partials_batch": [c_p, c_i, c_p],
    "kai0_pack_rows": [c_p, c_i, c_p, c_i64, c_p, c_i, c_p],
    "kai0_sampled_checksum": [c_p, c_i, c_i, c_p, c_p],
    "kai0_colsum_partials_bf16": [c_p, c_i64, c_i, c_i64, c_p, c_i, c_p, c_p],
    "kai0_rope_inplace": [c_p, c_p, c_p, c_i, c_i, c_i64, c_i64, c_i, c_i, c_i, c_p],
    "kai0_rope_inplace2": [c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_i64, c_i64, c_i, c_p],
    "kai0_rope_copy": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i64, c_i64, c_i, c_p],
    "kai0_softmax_mask_fwd": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i64, c_i64, c_i, c_i64, c_i64, c_p],
    "kai0_siglip_attn_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i64, c_i64, c_f, c_p],
    "kai0_rowdot_bf16": [c_p, c_p, c_p, c_i64, c_i, c_p],
    "kai0_softmax_bwd": [c_p, c_p, c_i, c_p, c_i64, c_i, c_i64, c_f, c_p],
    "kai0_geglu_fwd": [c_p, c_p, c_p, c_i64, c_p],
    "kai0_geglu_bwd": [c_p, c_p, c_p, c_p, c_p, c_i64, c_p],
    "kai0_gelu_bwd": [c_p, c_p, c_p, c_i64, c_p],
    "kai0_silu_fwd_f32": [c_p, c_p, c_i64, c_p],
    "kai0_silu_bwd_f32": [c_p, c_p, c_p, c_i64, c_p],
    "kai0_gated_fwd": [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p],
    "kai0_gated_bwd": [c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p],
    "kai0_embed_gather": [c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i64, c_i64, c_i64, c_p],
    "kai0_embed_grad": [c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i64, c_i64, c_i64, c_p],
    "kai0_cast_f32_to_bf16": [c_p, c_p, c_i64, c_p],
    "kai0_cast_bf16_to_f32": [c_p, c_p, c_i64, c_p],
    "kai0_add_bf16": [c_p, c_p, c_p, c_i64, c_p],
    "kai0_add_f32": [c_p, c_p, c_p, c_i64, c_p],
    "kai0_transpose_bf16": [c_p, c_p, c_i, c_i, c_p],
    "kai0_copy_rows_bf16": [c_p, c_p, c_i, c_i, c_i, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p],
    "kai0_patch_im2col": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "kai0_add_pos_cast": [c_p, c_p, c_p, c_i64, c_i, c_i, c_p],
    "kai0_time_sincos": [c_p, c_p, c_i, c_i, C.c_double, C.c_double, c_p],
    "kai0_flow_mix": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p],
    "kai0_mse_fwd": [c_p, c_p, c_p, c_i64, c_p],
    "kai0_mse_bwd": [c_p, c_p, c_p, c_p, c_i64, c_p],
    "kai0_euler_step": [c_p, c_p, c_f, c_i64, c_p],
    "kai0_denoise_glue": [c_p, c_p, c_i64, c_i, c_f, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_i64, c_i, c_i, c_p, c_p],
    "kai0_sumsq": [c_p, c_i, c_i64, c_p, c_p, c_p],
    "kai0_sum_chunks": [c_p, c_i, c_i, c_i64, c_i64, c_p, c_p],
    "kai0_clip_coef": [c_p, c_f, c_p, c_p, c_p],
    "kai0_adamw": [c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p],
    "kai0_adamw_rows": [c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i64, c_i, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p],
}  # fmt: skip

EXPORTED_SYMBOLS = ("kai0_last_error", "kai0_skinny_workspace_bytes", "kai0_attn_decode_workspace_bytes", "kai0_gemm_f32_workspace_bytes",
                    *_PROTOS.keys())

_lib = None


class Kai0HipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libkai0hip.so (built in-tree by `python -m kai0_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = pathlib.Path(os.environ.get("KAI0_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise Kai0HipError(
            f"{path} not found: the HIP extension is not built. Run `python -m kai0_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path."
        )
    # The kernels must run on the SAME HIP runtime instance as PyTorch (its streams and device pointers are
    # passed straight through the C ABI).  torch bundles its own libamdhip64; importing torch first makes the
    # dynamic loader resolve this library's libamdhip64.so.N dependency to that already-loaded copy.
    import torch  # noqa: F401

    lib = C.CDLL(str(path))
    lib.kai0_last_error.restype = C.c_char_p
    lib.kai0_last_error.argtypes = []
    for name, args in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = c_i
        fn.argtypes = args
    if lib.kai0_gemm_desc_size() != C.sizeof(GemmDesc):
        raise Kai0HipError(
            f"kai0_gemm_desc layout mismatch: C {lib.kai0_gemm_desc_size()} B vs ctypes {C.sizeof(GemmDesc)} B"
        )
    for fn, mirror in (("kai0_attn_desc_size", AttnDesc), ("kai0_attn_bwd_desc_size", AttnBwdDesc)):
        if getattr(lib, fn)() != C.sizeof(mirror):
            raise Kai0HipError(f"{fn[:-5]} layout mismatch: C {getattr(lib, fn)()} B vs ctypes {C.sizeof(mirror)} B")
    lib.kai0_attn_decode_workspace_bytes.restype = c_i64
    lib.kai0_attn_decode_workspace_bytes.argtypes = [c_i, c_i]
    lib.kai0_skinny_workspace_bytes.restype = c_i64
    lib.kai0_skinny_workspace_bytes.argtypes = [c_i, c_i, c_i]
    if lib.kai0_skinny_desc_size() != C.sizeof(SkinnyDesc):
        raise Kai0HipError(
            f"kai0_skinny_desc layout mismatch: C {lib.kai0_skinny_desc_size()} B vs ctypes {C.sizeof(SkinnyDesc)} B"
        )
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    """Invoke an entry point and raise with kai0_last_error() on a non-zero return."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise Kai0HipError(f"{name} failed ({rc}): {lib.kai0_last_error().decode()}")
