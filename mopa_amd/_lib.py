"""ctypes binding of libmopa_hip.so (the C-ABI declared in include/mopa_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call
returns a non-zero code, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOPA_HIP_LIB") or os.path.join(_HERE, "libmopa_hip.so")  # env override: A/B tuning builds only

_P, _I, _L, _Z, _F, _D = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float, ctypes.c_double
_T = {"p": _P, "i": _I, "l": _L, "z": _Z, "f": _F, "d": _D}

# name -> (restype, argtypes) ; 'p' pointer, 'i' int32, 'l' int64, 'z' size_t, 'f' float
SIGNATURES = {
    # ---- geometry (hash3d.hip)
    "mopa_voxel_hash_workspace_bytes": ("z", "l"),
    "mopa_voxel_hash_build": ("i", "plpplpppppzp"),
    "mopa_coarsen_workspace_bytes": ("z", "l"),
    "mopa_group_split": ("i", "ppiipp"),
    "mopa_coarsen_build": ("i", "pippplppppzp"),
    "mopa_rulebook_subm": ("i", "pipplipp"),
    "mopa_rulebook_updown": ("i", "ppiippp"),
    "mopa_points_csr_workspace_bytes": ("z", "l"),
    "mopa_points_csr": ("i", "piipppzp"),
    "mopa_rotate_points_f32": ("i", "pippp"),
    "mopa_voxelize_workspace_bytes": ("z", ""),
    "mopa_voxelize": ("i", "pifipilpppzp"),
    "mopa_scan_workspace_bytes": ("z", "l"),
    "mopa_scan_exclusive_i32": ("i", "ppippzp"),
    # ---- sparse conv (spconv.hip)
    "mopa_spconv_fwd": ("i", "piipiipiipip"),
    "mopa_rulebook_groups_count": ("i", "piipp"),
    "mopa_rulebook_groups_fill": ("i", "piippppp"),
    "mopa_rulebook_groups_count_batched": ("i", "piipp"),
    "mopa_rulebook_groups_fill_batched": ("i", "piippppp"),
    "mopa_spconv_grouped_workspace_bytes": ("z", "iii"),
    "mopa_spconv_fwd_grouped": ("i", "ppppiipiipiipipzp"),
    "mopa_spconv_transpose_weight": ("i", "piiipp"),
    "mopa_spconv_grouped_wants_packed": ("i", "iiii"),
    "mopa_spconv_pack_weight": ("i", "piiiiipp"),
    "mopa_spconv_pack_weights_batched": ("i", "pip"),
    "mopa_spconv_wgrad_workspace_bytes": ("z", "iiii"),
    "mopa_spconv_bwd_weight": ("i", "piipiipiipipzp"),
    # ---- sparse conv, offset-major (sprun.hip)
    "mopa_rulebook_runs_bytes": ("z", "ii"),
    "mopa_rulebook_runs_build_batched": ("i", "pip"),
    "mopa_spconv_run_pack_weight": ("i", "piiiipp"),
    "mopa_spconv_run_wanted": ("i", "iiiii"),
    "mopa_spconv_run_form": ("i", "ii"),
    "mopa_spconv_run_workspace_bytes": ("z", "iii"),
    "mopa_spconv_fwd_run": ("i", "piipiipiipiipzp"),
    "mopa_spconv_wgrad_run_wanted": ("i", "iiiii"),
    "mopa_spconv_wgrad_run_workspace_bytes": ("z", "iiiii"),
    "mopa_spconv_bwd_weight_run": ("i", "piiiipiipiipipzp"),
    # ---- native executor of the 3D layer program (scn_exec.hip)
    "mopa_scn_workspace_bytes": ("z", "pipii"),
    "mopa_scn_forward": ("i", "pippppppzp"),
    "mopa_scn_backward": ("i", "pipippppppppzp"),
    # ---- VGI (vgi.hip)
    "mopa_vgi_zslots": ("i", ""),
    "mopa_vgi_first_points": ("i", "piifiiiiipppp"),
    "mopa_vgi_box_free_workspace_bytes": ("z", "iii"),
    "mopa_vgi_box_free": ("i", "piiiiiiiippzp"),
    "mopa_vgi_ground_cells": ("i", "ppiipp"),
    "mopa_vgi_candidates": ("i", "piiippiiiippp"),
    "mopa_vgi_compact_cells": ("i", "piiiippp"),
    "mopa_vgi_road_height": ("i", "piifppiiiiiiipp"),
    "mopa_vgi_range_keep_workspace_bytes": ("z", "iii"),
    "mopa_vgi_range_keep": ("i", "piiddiippzp"),
    "mopa_voxelize_f64_workspace_bytes": ("z", ""),
    "mopa_voxelize_f64": ("i", "pippdipilpppzp"),
    # ---- pseudo-label update (pseudo.hip)
    "mopa_pseudo_fuse": ("i", "ppiippp"),
    "mopa_refine_pseudo_labels_workspace_bytes": ("z", "i"),
    "mopa_refine_pseudo_labels": ("i", "ppiilppzp"),
    "mopa_ema_update": ("i", "pplfp"),
    # ---- row ops (rows.hip)
    "mopa_bnrelu_rows_workspace_bytes": ("z", "ii"),
    "mopa_bnrelu_rows_fwd": ("i", "pipiiippppfffippzp"),
    "mopa_bnrelu_rows_bwd_workspace_bytes": ("z", "ii"),
    "mopa_bnrelu_rows_bwd": ("i", "pipipiiipfippiipzp"),
    "mopa_rows_add": ("i", "pipipiiip"),
    "mopa_zero_rows": ("i", "pilip"),
    "mopa_copy_rows": ("i", "pipilip"),
    "mopa_add_i64_many": ("i", "pilp"),
    # ---- command-list executor (exec2d.hip)
    "mopa_exec_fn_id": ("i", "p"),
    "mopa_exec_fn_count": ("i", ""),
    "mopa_exec_replay": ("i", "plp"),
    "mopa_input_layer_fwd": ("i", "pippipip"),
    "mopa_input_layer_bwd": ("i", "pippiipp"),
    "mopa_output_layer_heads_fwd": ("i", "pipiiipppppppp"),
    "mopa_output_layer_heads_bwd_workspace_bytes": ("z", "iii"),
    "mopa_output_layer_heads_bwd": ("i", "ppppppppiiiipippppipzp"),
    "mopa_bn_act_fwd": ("i", "pipiiippppfffipiippzp"),
    "mopa_bn_act_bwd": ("i", "pipipiiipfipipiiippiipzp"),
    "mopa_bn_act_fwd_groups": ("i", "pipiiiiiippppfffipiippzp"),
    "mopa_bn_act_bwd_groups": ("i", "pipipiiiiiipfipipiiippiipzp"),
    "mopa_bn_sync_moments": ("i", "piiippzp"),
    "mopa_bn_act_fwd_sync": ("i", "pipiiippppfffipipipp"),
    "mopa_bn_sync_bwd_sums": ("i", "pipiiipfipippippzp"),
    "mopa_bn_act_bwd_sync": ("i", "pipipiiipfipipiippiipp"),
    # ---- dense 2D branch (conv2d.hip, ops2d.hip)
    "mopa_conv2d_igemm": ("i", "pppppip"),
    "mopa_conv2d_igemm_batched": ("i", "ppppilllip"),
    "mopa_wino_weight": ("i", "piiipp"),
    "mopa_wino_input": ("i", "piiiiipp"),
    "mopa_wino_output": ("i", "piiiippiip"),
    "mopa_wino_dout": ("i", "piiiiipp"),
    "mopa_wino_wgrad_workspace_bytes": ("z", "iii"),
    "mopa_wino_bwd_weight": ("i", "ppiiipipzp"),
    "mopa_wino4_weight": ("i", "piiipp"),
    "mopa_wino4_weight_t": ("i", "piiipp"),
    "mopa_wino4_weight_f": ("i", "piiipp"),
    "mopa_wino4_conv": ("i", "pipppiiiiiiipiipp"),
    "mopa_wino4_weight_q": ("i", "piiipp"),
    "mopa_wino4_conv9": ("i", "pipppiiiiiiipiip"),
    "mopa_wino4_conv_tiles32": ("i", "i"),
    "mopa_conv2d_weight_forms_batched": ("i", "pip"),
    "mopa_wino4_gemm_output": ("i", "ppppiiiiiiip"),
    "mopa_wino4_input": ("i", "piiiiipp"),
    "mopa_wino4_input_bn": ("i", "piiiiipiipp"),
    "mopa_bn_bwd_sums_groups": ("i", "pipiiiiiipfipippippzp"),
    "mopa_stem_bwd_weight_bn": ("i", "ppipippiippipzp"),
    "mopa_wino4_output": ("i", "piiiippiip"),
    "mopa_wino4_dout": ("i", "piiiiipp"),
    "mopa_wino4_wgrad_workspace_bytes": ("z", "iii"),
    "mopa_wino4_bwd_weight": ("i", "ppiiipipzp"),
    # ---- F(4x4) weight gradient in one kernel (wino4wg.hip)
    "mopa_wino4_wgrad_fused_ok": ("i", "iiiii"),
    "mopa_wino4_wgrad_fused_workspace_bytes": ("z", "iiiii"),
    "mopa_wino4_wgrad_fused": ("i", "pipiipiiiiiipipzp"),
    "mopa_conv2d_wgrad_workspace_bytes": ("z", "p"),
    "mopa_conv2d_bwd_weight": ("i", "ppppipzp"),
    "mopa_conv2d_relayout_weight": ("i", "ppiiiiiiip"),
    "mopa_conv2d_stem_relayout": ("i", "ppiiip"),
    "mopa_img_to_nhwc4": ("i", "piiiiipp"),
    "mopa_stem_dgrad_image": ("i", "piiiiiippp"),
    "mopa_maxpool3x3s2_fwd": ("i", "piiiiipipp"),
    "mopa_maxpool3x3s2_fwd_bn": ("i", "piiiiipipipp"),
    "mopa_maxpool3x3s2_bwd": ("i", "pipiiiipiip"),
    "mopa_dropout_rows": ("i", "pipiliflp"),
    "mopa_dropout_rows_dseed": ("i", "pipilifpip"),
    "mopa_pixel_head_fwd": ("i", "piiiiiiiipppp"),
    "mopa_pixel_head_bwd_workspace_bytes": ("z", "iiiii"),
    "mopa_pixel_head_bwd": ("i", "ppiiiiiiiippiippipzp"),
    "mopa_colsum_workspace_bytes": ("z", "li"),
    "mopa_colsum": ("i", "pilipipzp"),
    # ---- losses (losses.hip)
    "mopa_loss_workspace_bytes": ("z", "l"),
    "mopa_softmax_kl_fwd": ("i", "ppiippzp"),
    "mopa_softmax_kl_bwd": ("i", "ppiippp"),
    "mopa_wce_fwd": ("i", "pppiilppppzp"),
    "mopa_wce_bwd": ("i", "pppiilpppp"),
    "mopa_softmax_fwd": ("i", "plipp"),
    "mopa_softmax_bwd": ("i", "pplipp"),
    "mopa_mask_cons_workspace_bytes": ("z", "iii"),
    "mopa_mask_cons_state_floats": ("z", "ii"),
    "mopa_mask_cons_fwd": ("i", "ppiiiiipppzp"),
    "mopa_mask_cons_bwd": ("i", "ppiiiiipppp"),
    # ---- optimiser (optim.hip)
    "mopa_adam_flat": ("i", "pppplffffffffp"),
}

from ._host_args import HOST_PARAMS  # noqa: E402  (generated: which arguments of an entry point are host pointers)

_lib = None


def load():
    """Load the HIP extension; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"mopa_amd: {LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mopa_amd/csrc`). There is no CPU fallback for the hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = _T[res]
        fn.argtypes = [_T[a] for a in args]
    _lib = lib
    return lib


def ptr(t, col: int = 0):
    """Device pointer of a tensor (optionally offset by `col` elements), or NULL."""
    if t is None:
        return None
    return t.data_ptr() + col * t.element_size()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """hipStream_t of torch's current stream on the current device (fast path: no Stream object, no lazy-init check: ~0.3 us;
    a 2D forward + backward asks ~740 times)."""
    if _raw_stream is not None:
        return _raw_stream(_raw_device() if _raw_device is not None else torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


RECORDER = None   # a CommandList while a pass is being recorded (mopa_amd/dense2d.py::Graph2D), else None


def call(name: str, *args):
    if RECORDER is not None:
        RECORDER.add(name, args)
    rc = getattr(_lib or load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}")


class CommandList:
    """A recorded sequence of this library's entry points (+ event hand-overs between two streams) that mopa_exec_replay
    (csrc/exec2d.hip) walks in ONE call: `words` = [op, nargs, args...]* as int64; integers and pointers as they are, float / double
    as the bits of a double.  The stream arguments (always the last one) are kept as slots: `main` / `side` word indices that
    replay() patches with the streams of the replaying pass.  Host arrays passed by pointer are copied into `blob`, which lives
    as long as the list: WHICH arguments are host pointers comes from the header's `_host` suffix (mopa_amd/_host_args.py,
    generated by csrc/gen_header.py), how many bytes they hold from HOST_BYTES below -- an entry point with a host pointer of
    unknown size is refused (a raw host address recorded verbatim would dangle at replay; the caller then runs the pass eagerly)."""

    # bytes behind a host-pointer parameter, by parameter name: (args, position) -> size
    HOST_BYTES = {"geom_host": lambda args, j: 100,                       # ConvGeom: 25 x int32 (csrc/conv2d.hip)
                  "ptrs_host": lambda args, j: 8 * int(args[j + 1])}      # mopa_add_i64_many(ptrs_host, n, ...)
    _ids = {}

    def __init__(self, main_stream: int):
        self.words, self.main, self.side, self.fix = [], [], [], []
        self.rec_main, self.rec_side = main_stream, None
        self.blob = bytearray()
        self.n_calls = 0

    def _stream_slot(self, handle, at):
        if handle == self.rec_main:
            self.main.append(at)
        else:
            if self.rec_side is None:
                self.rec_side = handle
            if handle != self.rec_side:
                raise RuntimeError("CommandList: a third stream inside a recorded pass")
            self.side.append(at)

    def add(self, name, args):
        import struct
        fid = self._ids.get(name)
        if fid is None:
            fid = self._ids[name] = int((_lib or load()).mopa_exec_fn_id(name.encode()))
        sig = SIGNATURES[name][1]
        if fid < 0 or not sig.endswith("p") or len(sig) != len(args):
            raise RuntimeError(f"CommandList: {name} cannot be recorded")
        host = HOST_PARAMS.get(name, {})
        for j, pname in host.items():
            if pname not in self.HOST_BYTES:
                raise RuntimeError(f"CommandList: {name} takes the host pointer `{pname}` of unknown size and cannot be recorded")
        base = len(self.words)
        self.words += [fid, len(args)]
        for j, (t, a) in enumerate(zip(sig, args)):
            if j in host:
                n = self.HOST_BYTES[host[j]](args, j)
                self.fix.append((base + 2 + j, len(self.blob)))
                self.blob += ctypes.string_at(a, n)
                self.blob += b"\0" * ((-len(self.blob)) % 16)
                self.words.append(0)
            elif t in ("f", "d"):
                self.words.append(struct.unpack("<q", struct.pack("<d", float(a)))[0])
            else:
                self.words.append(0 if a is None else int(a))
        self._stream_slot(0 if args[-1] is None else int(args[-1]), base + 2 + len(args) - 1)
        self.n_calls += 1

    def event_record(self, event_handle: int, stream_handle: int):
        base = len(self.words)
        self.words += [-1, 2, int(event_handle), 0]
        self._stream_slot(int(stream_handle), base + 3)

    def stream_wait(self, stream_handle: int, event_handle: int):
        base = len(self.words)
        self.words += [-2, 2, 0, int(event_handle)]
        self._stream_slot(int(stream_handle), base + 2)

    def finish(self):
        import numpy as np
        self.blob_np = np.frombuffer(bytes(self.blob) or b"\0", dtype=np.uint8).copy()
        w = np.asarray(self.words, dtype=np.int64)
        for at, off in self.fix:
            w[at] = self.blob_np.ctypes.data + off
        self.words_np = w
        self.main_np, self.side_np = np.asarray(self.main, dtype=np.int64), np.asarray(self.side, dtype=np.int64)
        self.fail = np.zeros(1, np.int64)
        self.words = None
        return self

    def replay(self, main_stream: int, side_stream: int | None):
        w = self.words_np
        w[self.main_np] = main_stream
        if len(self.side_np):
            if side_stream is None:
                raise RuntimeError("CommandList.replay: the recorded pass uses a second stream")
            w[self.side_np] = side_stream
        rc = (_lib or load()).mopa_exec_replay(w.ctypes.data, len(w), self.fail.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"mopa_exec_replay failed with code {rc} at word {int(self.fail[0])}")


_query_cache = {}

# Bumped by every in-place weight update that autograd's version counters cannot see (FlatAdam.step runs a HIP kernel on the
# flat buffer): per-step caches of re-laid-out weights key on (this epoch, tensor._version).
WEIGHTS_EPOCH = [0]


def query(name: str, *args) -> int:
    """Pure size / plan queries (workspace bytes, kernel plans): memoised per DEVICE -- some plans follow the current device's CU
    count (csrc/common.h::mopa_cu_count: 16- or 32-tile one-kernel convolution, the run-list weight gradient's piece size and
    with it its workspace), so in a process that drives different GPU models the answer for one device is not the other's."""
    if "p" in SIGNATURES[name][1]:   # takes a pointer (e.g. a geometry block): the address says nothing about the content
        return int(getattr(_lib or load(), name)(*args))
    key = (name, args, _raw_device() if (_raw_device is not None and torch.cuda.is_initialized()) else -1)
    v = _query_cache.get(key)
    if v is None:
        v = _query_cache[key] = int(getattr(_lib or load(), name)(*args))
    return v


class _Workspace:
    """One growing scratch buffer per (device, stream): kernels on one stream run in order, so they can share it;
    the 2D and 3D branches may run on different streams and must not."""

    def __init__(self):
        self.buf = {}

    def get(self, nbytes: int, device) -> torch.Tensor:
        key = (device.type, device.index, stream() if device.type == "cuda" else 0)
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


workspace = _Workspace()


# id(parameter) -> callable(parameter): "every kernel that writes this parameter's gradient in this backward pass has been enqueued"
# (FlatAdam's gradient buckets: mopa_amd/optim.py::FlatAdam.enable_buckets).  Empty unless an optimizer asked for it.
GRAD_DONE_HOOKS = {}


class GradSink:
    """Where the parameter gradients of one backward pass go.

    A parameter whose ``.grad`` is already attached (``FlatAdam`` keeps them as views of its flat gradient buffer, the one
    RCCL all-reduces) gets its gradient ACCUMULATED there by the kernel that computes it, and autograd is handed ``None``
    for it -- no temporary, no ``AccumulateGrad`` add kernel per parameter (82 / ~220 launches per 3D / 2D backward).
    Otherwise a fresh tensor is returned to autograd as usual.  ``MOPA_DIRECT_GRADS=0`` forces the second path.
    """

    def __init__(self, params: dict, order, defer_hooks=False):
        """defer_hooks: the caller takes every gradient buffer BEFORE it enqueues the kernels (the native 3D executor builds its
        pointer tables first): "gradient enqueued" is then reported once, at returned()."""
        self.params, self.order, self.defer_hooks = params, list(order), defer_hooks
        self.ret = {k: None for k in self.order}
        self.direct = os.environ.get("MOPA_DIRECT_GRADS", "1") != "0"
        self._pending = []   # attached parameters handed out by the previous take(): their kernels are enqueued when the next one starts
        self._loose = set()  # ids of parameters whose gradient went to a fresh tensor (autograd accumulates it after the pass)

    def _attached(self, p):
        g = p.grad
        return (self.direct and g is not None and g.shape == p.shape and g.dtype == p.dtype and g.device == p.device
                and g.is_contiguous())

    def take(self, *names):
        """-> ([gradient tensors], accumulate flag) for parameters that one kernel call writes together."""
        ps = [self.params[n] for n in names]
        if GRAD_DONE_HOOKS and not self.defer_hooks:
            self.flush()
        if all(self._attached(p) for p in ps):
            if GRAD_DONE_HOOKS and not self.defer_hooks:
                self._pending = ps
            return [p.grad for p in ps], True
        # not written in place: autograd adds the returned tensor to .grad AFTER this backward pass -- such a parameter is never
        # reported "done" from here (a bucket holding it stays for all_reduce() after backward)
        self._loose.update(id(p) for p in ps)
        ts = []
        for n, p in zip(names, ps):
            if self.ret[n] is None:
                self.ret[n] = torch.empty_like(p)
            ts.append(self.ret[n])
        return ts, False

    def flush(self, final=False):
        """The kernels of everything handed out so far are enqueued: tell whoever watches these parameters."""
        for p in self._pending:
            h = GRAD_DONE_HOOKS.get(id(p))
            if h is not None:
                h(p, final)
        self._pending = []

    def returned(self):
        if GRAD_DONE_HOOKS:   # end of the backward pass: every parameter of this network is final, taken or not (a head without a loss)
            self._pending = [p for p in self.params.values() if id(p) not in self._loose]   # ... unless autograd still has to add it
            self.flush(final=True)
        return tuple(self.ret[k] for k in self.order)



def upload(t: torch.Tensor, device) -> torch.Tensor:
    """Host tensor -> device.  A plain pageable copy on purpose: `t.pin_memory()` per call costs 17-20 ms for the 2 MB coordinate
    array of a batch on this platform (hipHostMalloc + CPU writes into uncached pinned memory; a persistent pinned staging buffer
    filled with `copy_` was as slow: 27 ms), the pageable `.to(device)` 0.1 ms (measured: 3.9 vs 22-39 ms per 3D forward+backward
    with host coordinates, the reference boundary's normal case)."""
    return t.contiguous().to(device)
