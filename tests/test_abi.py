"""CPU: the C-ABI library loads and exports exactly what include/hamt.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_protos():
    src = open(os.path.join(ROOT, "include", "hamt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|size_t)\s+(hamt_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        protos[m.group(1)] = n
    return protos


def test_header_and_binding_agree():
    from vln_hamt_amd import _lib
    protos = _header_protos()
    assert len(protos) >= 30
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, n in protos.items():
        assert len(_lib.SIGNATURES[name]) == n, (name, n, len(_lib.SIGNATURES[name]))


def test_library_exports_every_symbol():
    from vln_hamt_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in _header_protos():
        assert hasattr(lib, name), name
    assert lib.hamt_version() == 2
    buf = ctypes.create_string_buffer(64)
    assert lib.hamt_last_error(buf, 64) >= 0
    assert lib.hamt_last_kernel(buf, 64) >= 0
    # scratch sizes the callers allocate (hamt_workspace_bytes): pure host arithmetic
    assert _lib.workspace_bytes(_lib.WS_SUMSQ) == 4096
    assert _lib.workspace_bytes(_lib.WS_COLSUM, 5120, 768) == 64 * 768 * 4
    assert _lib.workspace_bytes(_lib.WS_LN_BWD, 5120, 768) == 3 * 256 * 768 * 4
    assert _lib.workspace_bytes(_lib.WS_WGRAD_TABLE, 768, 2304, 30522) == (12 + 36 + 477) * _lib.WGRAD_TABLE_ENTRY
    assert _lib.workspace_bytes(_lib.WS_LNRED_TABLE, 54) == 54 * _lib.LNRED_TABLE_ENTRY
    assert _lib.workspace_bytes(_lib.WS_GEMM_SPLITK, 5120, 3072, 768) == 0          # a grid that fills the chip is not split
    assert _lib.workspace_bytes(_lib.WS_GEMM_SPLITK, 128, 768, 8192) == 0           # narrow + small grid: K groups inside the workgroup
    ks = _lib.workspace_bytes(_lib.WS_GEMM_SPLITK, 512, 2048, 8192)
    assert ks > 0 and ks % (512 * 2048 * 4) == 0
    assert _lib.workspace_bytes(99) == 0


def test_struct_layouts_match_header():
    """field order/count of the descriptor structs (plain ints/floats, no padding surprises)."""
    from vln_hamt_amd import _lib
    assert ctypes.sizeof(_lib.GemmDesc) == 18 * 4 + 4 + 4 + 8 and _lib.GemmDesc.p_drop.offset == 72 and _lib.GemmDesc.rng.offset == 80
    assert ctypes.sizeof(_lib.AttnDesc) == 15 * 4
    assert ctypes.sizeof(_lib.LnDesc) == 8 * 4 and _lib.LnDesc.io16.offset == 28
    assert ctypes.sizeof(_lib.WgradDesc) == 4 * 8 + 8 * 4 + 8 + 2 * 4 + 2 * 8 + 4 * 4     # 4 pointers + 8 ints + the ss pointer + K_valid + wire_scale + the second operand pair
    assert _lib.WgradDesc.M.offset == 32 and _lib.WgradDesc.accum_db.offset == 60 and _lib.WgradDesc.ss.offset == 64 and _lib.WgradDesc.K_valid.offset == 72 and _lib.WgradDesc.dy2.offset == 80 and _lib.WGRAD_TABLE_ENTRY == 112
    assert ctypes.sizeof(_lib.LnReduceDesc) == 4 * 8 + 4 * 4 and _lib.LnReduceDesc.M.offset == 32 and _lib.LnReduceDesc.atomic.offset == 40      # hamt_ln_reduce_desc (3 ints + tail padding)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under vln_hamt_amd/ may import it (or torch CPU fallbacks)."""
    pkg = os.path.join(ROOT, "vln_hamt_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)


def test_ops_fail_loudly_without_gpu():
    import torch
    from vln_hamt_amd import ops
    from vln_hamt_amd._lib import HamtError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(HamtError):
        ops.linear(torch.randn(4, 8), torch.randn(8, 8), torch.randn(8), 0, "fp32")
    with pytest.raises(HamtError):
        ops.layer_norm(torch.randn(4, 8), None, torch.nn.LayerNorm(8))
