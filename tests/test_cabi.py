"""The C-ABI library loads and exports every symbol include/mopa_hip.h declares (no compute: no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mopa_hip.h")


def _protos():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t)\s+(mopa_\w+)\s*\(([^;]*)\)\s*;", src):
        args = [a for a in m.group(3).split(",") if a.strip() and a.strip() != "void"]
        out[m.group(2)] = (m.group(1), len(args))
    return out


def test_header_lists_the_whole_abi():
    protos = _protos()
    assert len(protos) >= 49
    for required in ("mopa_voxel_hash_build", "mopa_group_split", "mopa_rulebook_subm", "mopa_spconv_fwd", "mopa_spconv_bwd_weight",
                     "mopa_bnrelu_rows_fwd", "mopa_output_layer_heads_fwd", "mopa_conv2d_igemm", "mopa_conv2d_bwd_weight",
                     "mopa_maxpool3x3s2_fwd", "mopa_dropout_rows", "mopa_dropout_rows_dseed", "mopa_softmax_kl_fwd", "mopa_wce_fwd",
                     "mopa_mask_cons_fwd", "mopa_adam_flat"):
        assert required in protos


def test_library_exports_every_declared_symbol():
    from mopa_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _protos():
        assert hasattr(lib, name), f"{name} declared in mopa_hip.h but not exported"


def test_python_binding_matches_header():
    from mopa_amd import _lib
    protos = _protos()
    assert set(_lib.SIGNATURES) == set(protos), set(_lib.SIGNATURES) ^ set(protos)
    for name, (res, args) in _lib.SIGNATURES.items():
        ctype, nargs = protos[name]
        assert len(args) == nargs, (name, len(args), nargs)
        assert (res == "z") == (ctype == "size_t"), name
    _lib.load()  # sets argtypes on every symbol


def test_workspace_queries_run_on_the_host():
    from mopa_amd import _lib
    assert _lib.query("mopa_voxel_hash_workspace_bytes", 1000) > 0
    assert _lib.query("mopa_spconv_wgrad_workspace_bytes", 27, 10000, 32, 16) >= 27 * 32 * 16 * 4
    assert _lib.query("mopa_bnrelu_rows_workspace_bytes", 5000, 64) >= 5 * 2 * 64 * 4


def test_missing_library_fails_loudly(monkeypatch):
    from mopa_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmopa_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
