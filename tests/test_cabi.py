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


def test_command_list_executor_table_and_packing():
    """csrc/exec2d.hip without a GPU: every launching entry point of the binding table has an id in the generated dispatch table,
    the recorder packs arguments as the executor unpacks them (floats as the bits of a double, host arrays copied into the list's
    own blob, the stream argument kept as a patchable slot), and a malformed list is refused."""
    import struct
    import numpy as np
    from mopa_amd import _lib
    lib = _lib.load()
    launching = [n for n, (res, args) in _lib.SIGNATURES.items() if res == "i" and args.endswith("p") and not n.startswith("mopa_exec_")]
    protos = _protos()
    launching = [n for n in launching if protos[n][0] == "int"]
    ids = {n: lib.mopa_exec_fn_id(n.encode()) for n in launching}
    launching_with_stream = [n for n in launching if n not in ("mopa_spconv_grouped_wants_packed", "mopa_spconv_run_wanted", "mopa_spconv_run_form",
                                                               "mopa_vgi_zslots", "mopa_exec_fn_id", "mopa_exec_fn_count", "mopa_exec_replay")]
    assert all(ids[n] >= 0 for n in launching_with_stream), [n for n in launching_with_stream if ids[n] < 0]
    assert len(set(ids[n] for n in launching_with_stream)) == len(launching_with_stream)
    assert lib.mopa_exec_fn_count() >= len(launching_with_stream)
    assert lib.mopa_exec_fn_id(b"mopa_no_such_entry") == -1 and lib.mopa_exec_fn_id(b"mopa_spconv_run_wanted") == -1
    # an empty list replays; a truncated one and an unknown id are refused with the failing word reported
    fail = np.full(1, -1, np.int64)
    assert lib.mopa_exec_replay(np.zeros(1, np.int64).ctypes.data, 0, fail.ctypes.data) == 0
    bad = np.asarray([0, 5, 1, 2], np.int64)
    assert lib.mopa_exec_replay(bad.ctypes.data, len(bad), fail.ctypes.data) == -1
    unknown = np.asarray([10_000, 0], np.int64)
    assert lib.mopa_exec_replay(unknown.ctypes.data, len(unknown), fail.ctypes.data) == -1 and fail[0] == 0
    # packing
    rec = _lib.CommandList(main_stream=111)
    geom = (ctypes.c_int32 * 25)(*range(25))
    rec.add("mopa_conv2d_igemm", (1000, 2000, None, 3000, ctypes.addressof(geom), 1, 111))
    rec.add("mopa_dropout_rows", (10, 4, 20, 4, 7, 64, 0.4, 99, 222))          # second stream 222 = the side stream
    rec.event_record(555, 111)
    rec.stream_wait(222, 555)
    rec.finish()
    w = rec.words_np
    assert w[0] == ids["mopa_conv2d_igemm"] and w[1] == 7 and list(w[2:6]) == [1000, 2000, 0, 3000] and w[7] == 1
    got = (ctypes.c_int32 * 25).from_address(int(w[6]))
    assert list(got) == list(range(25)) and int(w[6]) != ctypes.addressof(geom)      # the geometry was copied
    d0 = 9
    assert w[d0] == ids["mopa_dropout_rows"] and struct.unpack("<d", struct.pack("<q", int(w[d0 + 2 + 6])))[0] == 0.4
    assert list(rec.main_np) == [8, d0 + 14] and list(rec.side_np) == [d0 + 10, d0 + 17]
    assert list(w[d0 + 11:d0 + 15]) == [-1, 2, 555, 0] and list(w[d0 + 15:d0 + 19]) == [-2, 2, 0, 555]
    with pytest.raises(RuntimeError):
        _lib.CommandList(1).add("mopa_spconv_run_wanted", (27, 10, 64, 64, 0))      # not a launching entry point
    rec3 = _lib.CommandList(1)
    rec3.add("mopa_zero_rows", (10, 4, 5, 4, 1))
    rec3.add("mopa_zero_rows", (10, 4, 5, 4, 2))
    with pytest.raises(RuntimeError):
        rec3.add("mopa_zero_rows", (10, 4, 5, 4, 3))                                  # a third stream
    # host pointers: which arguments are host pointers comes from the header's `_host` suffix; the stem's fused
    # weight gradient (arg 10 = geom_host) is copied like the other geometries, an entry point with a host table of unknown size
    # is refused instead of recording a raw host address, and a pointer list copies exactly n entries
    from mopa_amd._host_args import HOST_PARAMS
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    seen = 0
    for m in re.finditer(r"\b(?:int|size_t)\s+(mopa_\w+)\s*\(([^;]*)\)\s*;", src):
        params = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        want = {j: re.split(r"[\s\*]+", a)[-1] for j, a in enumerate(params) if "*" in a and re.split(r"[\s\*]+", a)[-1].endswith("_host")}
        assert HOST_PARAMS.get(m.group(1), {}) == want, m.group(1)
        seen += bool(want)
    assert seen == len(HOST_PARAMS) >= 15
    rec4 = _lib.CommandList(1)
    rec4.add("mopa_stem_bwd_weight_bn", (1, 2, 64, 3, 128, 4, 5, 2, 1, 6, ctypes.addressof(geom), 0, 7, 1024, 1))
    ptrs = (ctypes.c_int64 * 3)(11, 22, 33)
    rec4.add("mopa_add_i64_many", (ctypes.addressof(ptrs), 3, 5, 1))
    rec4.finish()
    w4 = rec4.words_np
    assert list((ctypes.c_int32 * 25).from_address(int(w4[2 + 10]))) == list(range(25)) and int(w4[2 + 10]) != ctypes.addressof(geom)
    a0 = 2 + 15
    assert list((ctypes.c_int64 * 3).from_address(int(w4[a0 + 2]))) == [11, 22, 33] and int(w4[a0 + 2]) != ctypes.addressof(ptrs)
    desc = (ctypes.c_int64 * 8)()
    with pytest.raises(RuntimeError, match="host pointer"):
        _lib.CommandList(1).add("mopa_conv2d_weight_forms_batched", (ctypes.addressof(desc), 1, 1))
