"""ctypes binding of libhamt_hip.so (include/hamt.h).

The product path has NO fallback: if the library is missing, or a call returns a negative status,
this module raises.  Build it with ``python vln_hamt_amd/csrc/build.py`` (or
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhamt_hip.so")

HAMT_F32, HAMT_BF16, HAMT_U8G, HAMT_F16 = 0, 1, 2, 3
PREC_BF16, PREC_F32 = 0, 1
EPI_BIAS, EPI_GELU, EPI_RELU, EPI_ACCUM, EPI_MUL_DGELU, EPI_MUL_DRELU, EPI_SAVE_PRE = 1, 2, 4, 8, 16, 32, 64
EPI_GELU_GRAD, EPI_MUL_AUX, EPI_ADD_AUX, EPI_DROPOUT = 128, 256, 512, 1024
SUMSQ_SPARSE = 2            # HAMT_SUMSQ_SPARSE (hamt_sumsq_table's `accumulate`, bit 1)

vp, i32, u32, f32, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_float, C.c_size_t


class GemmDesc(C.Structure):
    _fields_ = [("M", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldb", i32), ("ldc", i32), ("ldaux", i32),
                ("a_kmajor", i32), ("b_kmajor", i32), ("dtype_a", i32), ("dtype_b", i32), ("dtype_c", i32),
                ("dtype_aux", i32), ("prec", i32), ("epilogue", i32), ("alpha", f32), ("ka_rows", i32), ("kb_rows", i32),
                ("p_drop", f32), ("call_id", u32), ("rng", C.c_void_p)]


class LnReduceDesc(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dxsum", C.c_void_p), ("M", i32), ("H", i32), ("atomic", i32)]


LNRED_TABLE_ENTRY = 48


class AttnDesc(C.Structure):
    _fields_ = [("B", i32), ("heads", i32), ("Sq", i32), ("Sk", i32), ("d_head", i32), ("ldq", i32), ("ldk", i32),
                ("ldv", i32), ("ldo", i32), ("dtype_qkv", i32), ("dtype_o", i32), ("scale", f32), ("p_drop", f32),
                ("call_id", u32), ("prec", i32)]


LN_X_BF16, LN_Z_BF16, LN_X_F16, LN_Z_F16 = 1, 2, 4, 8      # hamt_ln_desc.io16


class LnDesc(C.Structure):
    _fields_ = [("M", i32), ("H", i32), ("eps", f32), ("p_pre", f32), ("p_post", f32), ("call_id", u32), ("Mpad16", i32), ("io16", i32)]


class VisEmbedDesc(C.Structure):
    _fields_ = [("M", i32), ("H", i32), ("A", i32), ("ld_ang", i32), ("eps1", f32), ("eps2", f32), ("x_bf16", i32), ("Mpad16", i32)]


# name -> argtypes (every entry point of include/hamt.h; tests/test_abi.py cross-checks against the header)
WGRAD_TABLE_ENTRY = 112     # HAMT_WGRAD_TABLE_ENTRY


class WgradDesc(C.Structure):
    _fields_ = [("dy", vp), ("x", vp), ("dw", vp), ("db", vp), ("M", i32), ("N", i32), ("K", i32), ("ldy", i32),
                ("ldx", i32), ("ldw", i32), ("accum_dw", i32), ("accum_db", i32), ("ss", vp), ("K_valid", i32),
                ("wire_scale", f32), ("dy2", vp), ("x2", vp), ("K2", i32), ("ldy2", i32), ("ldx2", i32), ("K2_valid", i32)]


SIGNATURES = {
    "hamt_version": [],
    "hamt_last_error": [C.c_char_p, sz],
    "hamt_last_kernel": [C.c_char_p, sz],
    "hamt_workspace_bytes": [i32, C.POINTER(i32), i32],
    "hamt_gemm": [C.POINTER(GemmDesc), vp, vp, vp, vp, vp, vp],
    "hamt_gemm_ksplit": [C.POINTER(GemmDesc)],
    "hamt_gemm_ws": [C.POINTER(GemmDesc), vp, vp, vp, vp, vp, vp, sz, vp],
    "hamt_cast_pad_bf16": [i32, i32, i32, i32, vp, i32, vp, i32, vp],
    "hamt_cast_pad_bf16_dropout": [i32, i32, i32, vp, i32, vp, i32, f32, u32, vp, vp],
    "hamt_cast_transpose": [i32, i32, vp, i32, i32, vp, i32, i32, vp],
    "hamt_wgrad_grouped": [i32, C.POINTER(WgradDesc), vp, sz, vp],
    "hamt_wgrad_grouped_ex": [i32, C.POINTER(WgradDesc), vp, sz, i32, vp],
    "hamt_smallk_wgrad": [i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp],
    "hamt_colsum": [i32, i32, vp, i32, i32, vp, i32, vp, vp],
    "hamt_attn_small_fwd": [C.POINTER(AttnDesc), vp, vp, vp, vp, vp, vp, vp, vp],
    "hamt_attn_small_bwd": [C.POINTER(AttnDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "hamt_attn_varlen_fwd": [C.POINTER(AttnDesc), vp, vp, vp, vp, vp, vp, vp, vp],
    "hamt_attn_varlen_bwd": [C.POINTER(AttnDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "hamt_attn_varlen_cross_fwd": [C.POINTER(AttnDesc), vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp],
    "hamt_attn_varlen_cross_bwd": [C.POINTER(AttnDesc), vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "hamt_vis_embed_fwd": [C.POINTER(VisEmbedDesc)] + [vp] * 12,
    "hamt_vis_embed_bwd": [C.POINTER(VisEmbedDesc)] + [vp] * 18,
    "hamt_ln_fwd": [C.POINTER(LnDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "hamt_ln_bwd": [C.POINTER(LnDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "hamt_ln_bwd_add": [C.POINTER(LnDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "hamt_ln_bwd_reduce": [i32, i32, vp, vp, vp, vp, vp],
    "hamt_ln_bwd_reduce_grouped": [i32, vp, vp, sz, vp],
    "hamt_gather_rows": [i32, i32, vp, i32, vp, vp, i32, vp, i32, i32, vp],
    "hamt_scatter_add_rows": [i32, i32, vp, i32, i32, vp, vp, i32, vp],
    "hamt_scatter_add_rows_small": [i32, i32, vp, i32, i32, vp, i32, vp, vp, vp],
    "hamt_scatter_add_rows_ordered": [i32, i32, vp, i32, i32, vp, vp, i32, i32, vp, vp],
    "hamt_embed_sum_fwd": [i32, i32, i32, vp, vp, vp, vp, vp, vp],
    "hamt_embed_sum_bwd": [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp],
    "hamt_mean_mid_fwd": [i32, i32, i32, vp, vp, vp],
    "hamt_mean_mid_bwd": [i32, i32, i32, vp, vp, vp],
    "hamt_mul_bcast_fwd": [i32, i32, i32, vp, vp, i32, vp, vp],
    "hamt_mul_bcast_bwd": [i32, i32, i32, vp, vp, i32, vp, vp, vp, vp],
    "hamt_sum_rows": [i32, i32, i32, vp, i32, vp, vp, vp],
    "hamt_patchify": [i32, i32, i32, i32, i32, vp, vp, i32, i32, i32, vp],
    "hamt_add3": [sz, vp, vp, vp, vp, vp],
    "hamt_dropout": [sz, vp, vp, f32, u32, vp, vp],
    "hamt_cast_f32_bf16": [sz, vp, vp, vp],
    "hamt_unpack_padded": [vp, vp, i32, i32, i32, i32, vp, vp],
    "hamt_seq_masks": [vp, i32, i32, i32, vp, vp, vp],
    "hamt_wire_pack_bf16": [C.c_size_t, vp, vp, f32, vp],
    "hamt_wire_unpack_bf16": [C.c_size_t, vp, vp, vp],
    "hamt_fill_where_zero": [sz, vp, vp, f32, vp],
    "hamt_act_bwd": [sz, vp, vp, i32, vp, vp],
    "hamt_ce_fwd": [i32, i32, vp, i32, vp, vp, vp, vp],
    "hamt_ce_bwd": [i32, i32, vp, i32, vp, vp, vp, vp, i32, vp],
    "hamt_mse_fwd": [sz, vp, vp, vp, vp],
    "hamt_mse_bwd": [sz, vp, vp, vp, vp, vp],
    "hamt_kl_fwd": [i32, i32, vp, i32, vp, i32, vp, vp, vp],
    "hamt_kl_bwd": [i32, i32, vp, i32, vp, i32, vp, vp, vp, i32, vp],
    "hamt_extend_mask": [sz, vp, vp, vp],
    "hamt_debug_fill_lds": [u32, vp],
    "hamt_debug_wgrad_timing": [C.c_int],
    "hamt_debug_wgrad_times": [vp, vp, vp, C.c_int],
    "hamt_a2c_fwd": [i32, i32, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp],
    "hamt_a2c_bwd": [i32, i32, vp, vp, vp, f32, vp, vp, vp, vp, vp],
    "hamt_sumsq": [sz, vp, vp, i32, vp, vp],
    "hamt_sumsq_table": [sz, sz, vp, vp, vp, i32, vp, i32, vp, vp],
    "hamt_sumsq_partials": [sz, vp, vp, i32, vp],
    "hamt_wire_unpack_sumsq": [sz, sz, vp, vp, vp, vp, i32, vp, i32, vp, vp],
    "hamt_adamw_flat": [sz, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, i32, vp],
    "hamt_adamw_table": [sz, vp, vp, vp, vp, vp, vp, vp, i32, vp, f32, f32, f32, f32, i32, vp],
    "hamt_adamw_table_range": [sz, sz, vp, vp, vp, vp, vp, vp, vp, i32, vp, f32, f32, f32, f32, i32, vp],
    "hamt_clip_scale": [sz, vp, vp, f32, vp],
    "hamt_rng_advance": [vp, vp],
}

_lib = None


class HamtError(RuntimeError):
    pass


def load():
    """Load libhamt_hip.so once.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HamtError(f"{LIB_PATH} is missing: build it with `python vln_hamt_amd/csrc/build.py` "
                        "(there is no CPU/PyTorch fallback for the HAMT kernels)")
    # Load order: PyTorch-ROCm ships its own libamdhip64 (torch/lib); libhamt_hip.so is linked against the soname and binds to whichever
    # runtime is already in the process.  Loaded BEFORE torch it pulls in /opt/rocm's build, torch then loads its own, and the process holds
    # two HIP runtimes -- the second never sees the device ("no ROCm-capable device is detected" at the first launch).  The host side of
    # this library is PyTorch's memory and streams anyway: make sure its runtime is the one.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = sz if name == "hamt_workspace_bytes" else i32
    if lib.hamt_version() != 2:
        raise HamtError("libhamt_hip.so ABI version mismatch")
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().hamt_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def last_kernel() -> str:
    """kernel the most recent hamt_gemm call of this thread launched (rocprofv3's name, template arguments included)"""
    buf = C.create_string_buffer(160)
    load().hamt_last_kernel(buf, 160)
    return buf.value.decode(errors="replace")


def workspace_bytes(op: int, *shape) -> int:
    arr = (i32 * max(1, len(shape)))(*shape)
    return int(load().hamt_workspace_bytes(op, arr, len(shape)))


WS_GEMM_SPLITK, WS_COLSUM, WS_SUMSQ, WS_LN_BWD, WS_WGRAD_TABLE, WS_LNRED_TABLE, WS_VIS_EMBED_BWD, WS_EMBED_BWD = range(8)


def check(rc: int, name: str):
    if rc != 0:
        raise HamtError(f"{name} failed (status {rc}): {last_error()}")
