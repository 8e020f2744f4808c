"""Host side of the 2D image branch: UNetResNet34 + Net2DSeg heads as ONE autograd node over the C-ABI kernels.

What it replaces: the torch.nn / cuDNN graph of ``mopa/models/resnet34_unet.py:131-191`` and the heads / point
gather of ``mopa/models/xmuda_arch.py:49-79``.  Activations are NHWC fp32 (``[B*H*W, C]`` row-major, optionally a
column slice of a wider buffer, which is how the decoder's ``torch.cat([skip, up])`` disappears); parameters keep
torch's layouts and names (``state_dict`` compatible) and are re-laid-out for the implicit-GEMM kernels on the fly.
Oracle: ``oracle/net2d.py``; golden fixture G1.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

from . import _lib, syncbn
from ._lib import GradSink, call, ptr, query, stream, workspace
from .sparse3d import View

BN_EPS = 1e-5       # torch.nn.BatchNorm2d defaults
BN_MOMENTUM = 0.1
DEBUG = None  # tests may set this to a dict to capture intermediate gradients
LAYERS = [("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 6, 2), ("layer4", 512, 3, 2)]


def _ws(n, dev):
    return workspace.get(max(int(n), 256), dev)


class Img(View):
    """NHWC activation: rows = B*H*W pixels, C channels at column offset `col` of tensor t (ld = t.shape[1])."""

    __slots__ = ("B", "H", "W")

    def __init__(self, t, B, H, W, col=0, C=None):
        super().__init__(t, col, C)
        self.B, self.H, self.W = B, H, W


class LazyImg(Img):
    """relu(batchnorm(x)) that is never written: `t` is the BatchNorm's INPUT x, `bn` = (stats (G, 4, C), G).  The one consumer --
    a Winograd F(4x4) convolution -- applies scale / shift / ReLU inside its input transform (mopa_wino4_input_bn, bit-identical
    to transforming the materialised tensor) and keeps V for its weight gradient; BatchNorm's backward recomputes the mask from x
    anyway.  One apply pass (read x, write y) and one launch less per such layer."""

    __slots__ = ("bn",)

    def __init__(self, x: Img, stats, G, c0=0):
        """c0 > 0: only channels [c0, C) of x are the BatchNorm's input (stats: (G, 4, C - c0)); the channels below are a tensor that
        is non-negative already (the skip half of a decoder join buffer) and pass through."""
        super().__init__(x.t, x.B, x.H, x.W, x.col, x.C)
        self.bn = (stats, G, c0)


def new_img(B, H, W, C, dev, ld=None, zero=False):
    alloc = torch.zeros if zero else torch.empty
    return Img(alloc(B * H * W, ld or C, dtype=torch.float32, device=dev), B, H, W, 0, C)


def _geom(**kw):
    order = ["B", "IH", "IW", "OHl", "OWl", "OHa", "OWa", "OS", "OOY", "OOX", "IS", "IY0", "IX0", "IDY", "IDX",
             "TH", "TW", "KH0", "KW0", "KS", "KWF", "Cin", "Cout", "ld_in", "ld_out"]
    d = dict(OS=1, OOY=0, OOX=0, IS=1, IY0=0, IX0=0, IDY=1, IDX=1, KH0=0, KW0=0, KS=1)
    d.update(kw)
    return (ctypes.c_int32 * 25)(*[int(d[k]) for k in order])


def igemm(x_ptr, w, bias, out_ptr, geom, accumulate=False):
    call("mopa_conv2d_igemm", x_ptr, ptr(w), ptr(bias), out_ptr, ctypes.addressof(geom), int(accumulate), stream())


def igemm_batched(x_ptr, w_ptr, out_ptr, geom, nbatch, in_stride, w_stride, out_stride):
    """nbatch independent problems of one geometry in one launch (the 16 transform points of the Winograd path)."""
    call("mopa_conv2d_igemm_batched", x_ptr, w_ptr, out_ptr, ctypes.addressof(geom), nbatch, in_stride, w_stride, out_stride, 0,
         stream())


def wgrad(x_ptr, dy_ptr, dw_ptr, geom, dev, accumulate=False, oihw=False):
    """oihw: dw_ptr is the parameter-layout gradient tensor itself (plain convolutions only), not the igemm layout."""
    wsb = query("mopa_conv2d_wgrad_workspace_bytes", ctypes.addressof(geom))
    ws = _ws(wsb, dev)
    call("mopa_conv2d_bwd_weight", x_ptr, dy_ptr, dw_ptr, ctypes.addressof(geom), int(accumulate) | (2 if oihw else 0), ptr(ws),
         ws.numel(), stream())


def relayout(src, dst, O, I, KH, KW, mode, inverse=False, accumulate=False):
    call("mopa_conv2d_relayout_weight", ptr(src), ptr(dst), O, I, KH, KW, mode, int(inverse), int(accumulate), stream())


class _CaptureMiss(RuntimeError):
    """Raised while recording a HIP graph when the pass needs something the eager first pass did not leave behind."""


_CAPTURE = None   # raw handle of the stream that will replay the graph being recorded (None: not recording)
_relayout_cache = {}
_refreshed = {}   # stream -> WEIGHTS_EPOCH of the last batched refresh
BATCHED_REFRESH = True   # one launch for all stale weight forms (tests flip the attribute to compare with the one-form kernels)


def _refresh_stale_forms(st):
    """Rebuild EVERY cached weight form of this stream whose weight changed since it was built in one launch
    (mopa_conv2d_weight_forms_batched) -- igemm layouts and Winograd transforms of all convolutions, ~95 forms per joint
    step, each of which used to be an 8-10 us kernel serialised on the main stream at the layer's first use after the
    optimizer step.  The cached destination tensors are re-used (same sizes; their readers are ahead on the same stream).
    A form that was never built on this stream still takes the single-kernel path."""
    from ._lib import WEIGHTS_EPOCH
    epoch = WEIGHTS_EPOCH[0]
    rows, hits = [], []
    for key, ent in list(_relayout_cache.items()):
        if key[2] != st or len(ent) < 4 or ent[3] is None:
            continue
        tag, t, wref, meta = ent
        w = wref()
        if w is None:
            del _relayout_cache[key]
            continue
        new_tag = (epoch, w._version, w.data_ptr())
        if tag == new_tag:
            continue
        rows.append((w.data_ptr(), t.data_ptr()) + meta)
        hits.append((key, new_tag, t, wref, meta))
    for i in range(0, len(rows), 48):
        desc = np.asarray(rows[i:i + 48], dtype=np.int64)
        call("mopa_conv2d_weight_forms_batched", desc.ctypes.data, len(desc), st)
    for key, new_tag, t, wref, meta in hits:
        _relayout_cache[key] = (new_tag, t, wref, meta)


def _cached_weight_form(w, form, build, meta=None):
    """A derived layout of a conv weight, re-used until the weight changes (an iteration runs the network on the source and
    on the target batch with the same weights).  An entry belongs to ONE live tensor object (weak reference: the allocator
    hands a freed weight's address to the next tensor), one weight version (autograd's counter + the epoch that FlatAdam /
    FlatEMA bump for their raw in-place updates) and one stream (the copy is only ordered with work of the stream that
    made it).  `meta` = (O, I, KH, KW, kind, arg) of mopa_conv2d_weight_forms_batched lets a later refresh rebuild the form
    together with all other stale ones."""
    import weakref
    from ._lib import WEIGHTS_EPOCH
    if _CAPTURE is not None:
        # launches are being recorded into a HIP graph (Graph2D below): hand out the form that the replaying stream owns and record
        # nothing -- Graph2D.replay brings every form of that stream up to date (one batched launch) before each replay
        hit = _relayout_cache.get((id(w), form, _CAPTURE))
        if hit is None or hit[2]() is not w:
            raise _CaptureMiss(f"weight form {form} of a {tuple(w.shape)} weight was never built on the replaying stream")
        return hit[1]
    st = stream()
    key = (id(w), form, st)
    tag = (WEIGHTS_EPOCH[0], w._version, w.data_ptr())
    hit = _relayout_cache.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is w:
        return hit[1]
    if hit is not None and hit[2]() is w and BATCHED_REFRESH and _refreshed.get(st) != tag[0]:
        _refreshed[st] = tag[0]      # once per epoch and stream
        _refresh_stale_forms(st)
        hit = _relayout_cache.get(key)
        if hit is not None and hit[0] == tag:
            return hit[1]
    t = build()
    if len(_relayout_cache) > 4096:   # temporaries (tests, one-off calls) must not pile up
        _relayout_cache.clear()
    _relayout_cache[key] = (tag, t, weakref.ref(w), meta)
    return t


def relayout_cached(w, shape, O, I, KH, KW, mode):
    """Forward / backward-data implicit-GEMM layout of a conv weight (cached per weight version)."""
    def build():
        t = torch.empty(*shape, dtype=torch.float32, device=w.device)
        relayout(w, t, O, I, KH, KW, mode)
        return t
    return _cached_weight_form(w, ("relayout", mode), build, (O, I, KH, KW, 0, mode))


def wino_weight_cached(w, dgrad: bool, F: int = 2, transposed: int = 0):
    """U[(F+2)^2][R][C] = G g G^T of a 3x3 OIHW weight (dgrad: rotated + transposed filter), cached per weight version.
    transposed (F = 4 only): 1 = U^T[36][C][R], the k-contiguous operand of the fused GEMM + output-transform kernel; 2 = the MFMA
    B-operand fragments of the one-kernel convolution (mopa_wino4_conv), [36][R / 16][C / 16][64 lanes][4]; 3 = those of its second
    form (mopa_wino4_conv9), [36][R / 16][C / 32][2][64 lanes][4]."""
    O, I = w.shape[0], w.shape[1]
    transposed = int(transposed)

    def build():
        R, C = (O if dgrad else I), (I if dgrad else O)
        u = torch.empty((F + 2) ** 2, (C if transposed == 1 else R), (R if transposed == 1 else C), dtype=torch.float32, device=w.device)
        name = "mopa_wino_weight" if F == 2 else ("mopa_wino4_weight", "mopa_wino4_weight_t", "mopa_wino4_weight_f", "mopa_wino4_weight_q")[transposed]
        call(name, ptr(w), O, I, int(dgrad), ptr(u), stream())
        u._mopa_wino_layout = (F, transposed)   # (layouts 0 and 2 have the same shape when R == C: wino_conv checks this tag)
        return u
    assert not (transposed and F != 4)
    return _cached_weight_form(w, ("wino", F, int(dgrad), transposed), build,
                               (O, I, 3, 3, 1 if F == 2 else 2, int(dgrad) | (transposed << 1)))


# F(4x4) layers whose 36 GEMMs and output transform run as ONE kernel (mopa_wino4_gemm_output: M is never materialised): needs
# 64-aligned input channels and ~1.4 rounds of its 64-tile x 32-channel blocks (700; round 4 said 1024 from the one bench shape,
# profiles/r5_algo_table.md has 4 / 8 / 16 images at both resolutions: 750-block grids win by 10-20 %): the 304x480 layers and
# the 128-output-channel layers at 152x240 (measured in csrc/wino2d.hip; -0.55 ms per 8-image forward + backward).  Smaller
# grids stay on the batched GEMM + output transform.  MOPA_WINO4_FUSED=0 switches it off.
WINO4_FUSED_MIN_BLOCKS = int(os.environ.get("MOPA_WINO4_FUSED_MIN_BLOCKS", "700")) if os.environ.get("MOPA_WINO4_FUSED", "1") != "0" else 1 << 62


def wino4_fused(cin, cout, B, H, W):
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    blocks = ((T + 63) // 64) * (cout // 32)
    # (64 input channels = one K chunk per point: there the batched GEMM is the most memory-bound and the fused kernel already
    #  wins on one round of blocks -- 64 -> 64 at 152x240: 189 -> 175 us)
    return cin % 64 == 0 and cout % 32 == 0 and (blocks >= WINO4_FUSED_MIN_BLOCKS or (cin == 64 and 2 * blocks >= WINO4_FUSED_MIN_BLOCKS))


# F(4x4) layers that run as ONE kernel (mopa_wino4_conv: input transform, 36 GEMMs and output transform; neither V nor M reaches
# HBM): 64-aligned channels on both sides and at least MOPA_WINO4_DIRECT_MIN_TILES tiles (twice that for 128 output channels).
# MOPA_WINO4_DIRECT=0 switches it off.
WINO4_DIRECT = os.environ.get("MOPA_WINO4_DIRECT", "1") != "0"
WINO4_DIRECT_MIN_TILES = int(os.environ.get("MOPA_WINO4_DIRECT_MIN_TILES", "4096"))   # (round 4: 16384, tuned at 16 x 302 x 480 only)
WINO4_DIRECT_MAX_CIN = 128
# Roles: "dgrad" (backward-data), "fwd_eval" (a forward pass that keeps nothing), "fwd" (the forward pass of a training step: it wants
# V again for the weight gradient, the kernel stores it as a by-product and only a READ of V is saved -- measured neutral in the joint
# step, 342 either way against 337 without the kernel, so it stays on the two-kernel path by default).
WINO4_DIRECT_ROLES = tuple(r for r in os.environ.get("MOPA_WINO4_DIRECT_ROLES", "fwd_eval,dgrad").split(",") if r)


CUS = 256   # (MI355X; the dispatch rules below were measured there and the CPU tests must agree with the committed table)


def wino4_direct(cin, cout, B, H, W, role="fwd"):
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    # tiles from which the one-kernel form wins (profiles/r6_algo_table.md: 4 / 8 / 16 images at 225 x 400 and 302 x 480): 64 output
    # channels -- the input transform is done once -- from ~4,000 tiles; 128 output channels from ~8,000 with the first form
    # (mopa_wino4_conv) and, with the second (mopa_wino4_conv9: one 32-tile x 64-channel item per workgroup, one workgroup per CU),
    # from 2,560 tiles WHEN its items fill their rounds of CUs to >= 0.7 (3,000 / 6,000 / 9,120 tiles: 0.77-0.78 x the two-kernel
    # form; 2,280 / 4,560 tiles = 144 / 286 items on 256 CUs: 0.94 / 1.03 x)
    # (more than 128 output channels repeat the input transform per 64 of them: 128 -> 256 loses to the fused form by 10 %)
    if not (WINO4_DIRECT and role in WINO4_DIRECT_ROLES and cin % 64 == 0 and cout % 64 == 0 and cin <= WINO4_DIRECT_MAX_CIN and cout <= 128):
        return False
    if cout <= 64:
        return T >= WINO4_DIRECT_MIN_TILES
    if wino4_conv9(cin, cout, role) and T >= WINO4_DIRECT_MIN_TILES * 5 // 8:
        items = (T + 31) // 32 * (cout // 64)
        if items >= 0.7 * ((items + CUS - 1) // CUS * CUS):
            return True
    return T >= 2 * WINO4_DIRECT_MIN_TILES


# The one-kernel convolution's second form (mopa_wino4_conv9, csrc/wino4c9.hip: nine transform points per wave on 32x32x2 MFMAs, raw
# patches staged by LDS-DMA): 0.75-0.94 x the first form's time on every layer shape of the table (profiles/bench_conv9.py) -- used
# wherever the pass keeps no V (it has no V by-product) and the output channels come in blocks of 64.  MOPA_WINO4_CONV9=0 = off.
WINO4_CONV9 = os.environ.get("MOPA_WINO4_CONV9", "1") != "0"


def wino4_conv9(cin, cout, role):
    return WINO4_CONV9 and role != "fwd" and cin % 16 == 0 and cout % 64 == 0


def wino4_layout(cin, cout, B, H, W, role="fwd"):
    """Weight form of an F(4x4) layer: 3 / 2 = fragments (one-kernel convolution, second / first form), 1 = transposed (fused GEMM +
    output transform), 0."""
    if wino4_direct(cin, cout, B, H, W, role):
        return 3 if wino4_conv9(cin, cout, role) else 2
    return int(wino4_fused(cin, cout, B, H, W))


def wino_tile(cin, cout, k, s, p, B, H, W, role="fwd"):
    """Which algorithm runs a stride-1 3x3 convolution (and its backward-data): 0 = direct implicit GEMM, 2 = Winograd
    F(2x2,3x3), 4 = F(4x4,3x3).  Measured against the direct MFMA kernel at batch 8 (profiles/bench_wino.py):
    F(4x4) -- 4x fewer multiplies, V / M = 2.25x the activations, ~1e-5 relative error -- 1.2x at 64^2 152x240, 1.4x at
    128->64 152x240, 2.0x at 128^2 76x120, 2.7x at 256^2 38x60, 3.5x at 512^2 19x30; F(2x2) -- 2.25x fewer multiplies, V / M =
    4x the activations, ~2e-6 -- 0.8x / 0.9x / 1.3x / 1.8x / 2.4x on the same shapes: it remains for maps below 8 pixels
    (MOPA_WINOGRAD_F4=0 forces it where it is eligible)."""
    if not (WINOGRAD and k == 3 and s == 1 and p == 1 and cin % 16 == 0 and cout % 64 == 0):
        return 0
    if (role in F4_ROLES and min(H, W) >= 8 and B * H * W < F4_MAX_PIXELS
            and (role != "fwd" or B * H * W >= F4_FWD_MIN_PIXELS)):
        return 4
    return 2 if max(cin, cout) >= 128 and B * H * W < 200000 else 0


# (pixels per launch up to which F(4x4) is chosen: 16 x 304 x 480 = 2.3 M -- source + target batch in one pass -- is inside; the
#  transformed operands V / M of one layer are 2.25 x its activations, and the fused kernel addresses one transform point with 32-bit
#  byte offsets: mopa_wino4_gemm_output refuses T * Cin * 4 >= 2^32)
F4_MAX_PIXELS = int(os.environ.get("MOPA_WINOGRAD_F4_PIXELS", "3000000"))
# module-level switches (read once: the algorithm choice is asked ~460 times per forward + backward; tests set the attributes)
WINOGRAD = os.environ.get("MOPA_WINOGRAD", "1") != "0"
WINOGRAD_WGRAD = os.environ.get("MOPA_CONV2D_MFMA", "1") != "0" and os.environ.get("MOPA_WINOGRAD_WGRAD", "1") != "0"
# Which passes use F(4x4): all three (round 2; round 1 shipped "dgrad,wgrad").  In the backward passes its rounding error
# (~1e-5 relative per layer) is a linear perturbation.  In the FORWARD pass the same error also moves a few ReLU pre-activations
# across zero, and layers that normalise over few samples amplify such a flip -- measured against the fp64 oracle
# (profiles/f4_gradient_noise.py, 8 x 302 x 480, median / 90th percentile / max error over the parameter gradients): direct
# kernels 1.0 / 2.1 / 4.3 %, F(2x2) forward 1.0 / 2.1 / 5.0 %, F(4x4) forward 1.4 / 2.7 / 6.7 % -- the fp32-vs-fp64 noise of
# this network is ~1 % whatever the algorithm and F(4x4) adds a third to it; on a tiny 2 x 160 x 224 input single layer4
# tensors (70 samples per channel) went from 1.5 % to 17 %, hence F4_FWD_MIN_PIXELS below.  The logits stay inside the
# tolerance every parity test states (rtol 1e-3 / atol 2e-4 against the reference-generated fixtures G1 / G1b and the oracle;
# 1e-4 against the exact-product forward) and the whole GPU suite passes either way.  For scale: the reference's own arithmetic
# on the platform it ships for (NGC PyTorch 21.06, Ampere; mopa/common/utils/torch_util.py:12-13 sets cudnn.deterministic
# but leaves allow_tf32 at its default) is TF32 convolutions, ~1e-3 per layer, with cuDNN free to pick Winograd.
# Worth 9 % of the joint step (256 -> 280 scans/s, same box).  MOPA_WINOGRAD_F4_ROLES=dgrad,wgrad restores the exact-product
# forward pass (F(2x2) / direct kernels: bit-identical logits to round 1).
F4_ROLES = tuple(r for r in os.environ.get("MOPA_WINOGRAD_F4_ROLES", "fwd,dgrad,wgrad").split(",") if r) \
    if os.environ.get("MOPA_WINOGRAD_F4", "1") != "0" else ()
F4_FWD_MIN_PIXELS = 4096


def wino_eligible(cin, cout, k, s, p, B, H, W):
    return wino_tile(cin, cout, k, s, p, B, H, W) != 0


def wino_conv(x_p, ld_in, B, H, W, cin, cout, U, bias, out_p, ld_out, accumulate=False, F=2, bn_in=None, role="fwd", want_v=True):
    """out = conv3x3(x) (+ bias) through the input transform -> (F+2)^2 batched GEMMs -> the output transform.
    U: wino_weight_cached(w, dgrad, F, transposed=(F == 4 and wino4_fused(cin, cout, B, H, W))).
    bn_in = (stats, G, c0) (F = 4 only): x is a BatchNorm's input and relu(batchnorm(x)) is what gets convolved (LazyImg)."""
    dev = U.device
    th, tw = (H + F - 1) // F, (W + F - 1) // F
    T, NP = B * th * tw, (F + 2) ** 2
    sfx = "" if F == 2 else "4"
    if F == 4 and wino4_direct(cin, cout, B, H, W, role) and wino4_conv9(cin, cout, role) and not want_v:
        if tuple(U.shape) != (36, cin, cout) or getattr(U, "_mopa_wino_layout", (4, 3)) != (4, 3):
            raise RuntimeError("wino_conv: the nine-point one-kernel F(4x4) path takes its own fragment form (wino_weight_cached(..., transposed=3))")
        call("mopa_wino4_conv9", x_p, ld_in, ptr(U), ptr(bias) if bias is not None else None, out_p, ld_out, B, H, W, cin, cout,
             int(accumulate), ptr(bn_in[0]) if bn_in is not None else None, bn_in[1] if bn_in is not None else 1,
             bn_in[2] if bn_in is not None else 0, stream())
        return None
    if F == 4 and wino4_direct(cin, cout, B, H, W, role):
        if tuple(U.shape) != (36, cin, cout) or getattr(U, "_mopa_wino_layout", (4, 2)) != (4, 2):
            raise RuntimeError("wino_conv: the one-kernel F(4x4) path takes the fragment weight form (wino_weight_cached(..., transposed=2))")
        V = torch.empty(NP * T * cin, dtype=torch.float32, device=dev) if want_v else None   # (a by-product, for the weight gradient)
        call("mopa_wino4_conv", x_p, ld_in, ptr(U), ptr(bias) if bias is not None else None, out_p, ld_out, B, H, W, cin, cout,
             int(accumulate), ptr(bn_in[0]) if bn_in is not None else None, bn_in[1] if bn_in is not None else 1,
             bn_in[2] if bn_in is not None else 0, ptr(V) if V is not None else None, stream())
        return V
    V = torch.empty(NP * T * cin, dtype=torch.float32, device=dev)
    if bn_in is not None:
        if F != 4:
            raise RuntimeError("wino_conv: a deferred BatchNorm needs the F(4x4) input transform")
        call("mopa_wino4_input_bn", x_p, ld_in, B, H, W, cin, ptr(bn_in[0]), bn_in[1], bn_in[2], ptr(V), stream())
    else:
        call(f"mopa_wino{sfx}_input", x_p, ld_in, B, H, W, cin, ptr(V), stream())
    if F == 4 and wino4_fused(cin, cout, B, H, W):
        if tuple(U.shape) != (36, cout, cin) or getattr(U, "_mopa_wino_layout", (4, 1)) != (4, 1):
            raise RuntimeError("wino_conv: the fused F(4x4) path takes the transposed weight transform (wino_weight_cached(..., transposed=1))")
        call("mopa_wino4_gemm_output", ptr(V), ptr(U), ptr(bias) if bias is not None else None, out_p, ld_out, B, H, W, cin, cout,
             int(accumulate), stream())
        return V
    if getattr(U, "_mopa_wino_layout", (F, 0)) != (F, 0):
        raise RuntimeError("wino_conv: the batched-GEMM path takes the plain weight transform")
    M = torch.empty(NP * T * cout, dtype=torch.float32, device=dev)
    g1 = _geom(B=1, IH=1, IW=T, OHl=1, OWl=T, OHa=1, OWa=T, TH=1, TW=1, KWF=1, Cin=cin, Cout=cout, ld_in=cin, ld_out=cout)
    igemm_batched(ptr(V), ptr(U), ptr(M), g1, NP, T * cin, cin * cout, T * cout)
    call(f"mopa_wino{sfx}_output", ptr(M), B, H, W, cout, ptr(bias) if bias is not None else None, out_p, ld_out, int(accumulate), stream())
    return V


def wino_wgrad_eligible(cin, cout, k, s, p, B, H, W):
    """The weight gradient of the same layers in the transform domain (dU[p] = V[p]^T dM[p], batched 1x1 weight gradients
    on the MFMA kernel, then G^T dU G): needs 64-aligned channels on both sides and the MFMA build."""
    F = wino_tile(cin, cout, k, s, p, B, H, W, "wgrad")
    return (F != 0 and cin % 64 == 0 and cout % 64 == 0 and min(cin, cout) >= (64 if F == 4 else 128)
            and WINOGRAD_WGRAD)


# F(4x4) layers whose weight gradient runs as ONE kernel from x and dY (mopa_wino4_wgrad_fused, csrc/wino4wg.hip: neither V nor dM in
# HBM) -- the 64 / 128-channel layers with many tiles, where the two-operand form is bound by the 2 x 2.25 x activations it moves.  The
# training forward pass of such a layer keeps no V (it may then run as the one-kernel convolution).  MOPA_WINO4_WGRAD_FUSED=0 = off.
WINO4_WGRAD_FUSED = os.environ.get("MOPA_WINO4_WGRAD_FUSED", "1") != "0"
WINO4_WGRAD_FUSED_MIN_TILES = int(os.environ.get("MOPA_WINO4_WGRAD_FUSED_MIN_TILES", "8192"))   # (64 output channels: 5/8 of it)


def wino4_wgrad_fused(cin, cout, B, H, W):
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    # (measured per shape at 2 / 4 / 8 / 16 images, profiles/bench_wgrad2d.py + profiles/r6_algo_table.md: the kernel itself passes the
    #  two-operand form at ~10,000 tiles (64 -> 64) / ~18,000 (128 -> 64), but what is chosen is the PAIR forward + weight gradient: a
    #  forward pass that keeps no V runs as mopa_wino4_conv9 -- 24-36 us faster at 6,000 tiles for 7-15 us lost here)
    return (WINO4_WGRAD_FUSED and cin % 32 == 0 and cout % 64 == 0 and cin <= 128 and cout <= 128
            and T >= (WINO4_WGRAD_FUSED_MIN_TILES if cout > 64 else WINO4_WGRAD_FUSED_MIN_TILES * 5 // 8)
            and bool(query("mopa_wino4_wgrad_fused_ok", B, H, W, cin, cout)))


def wino_wgrad(x: Img, dout: Img, cin, cout, dw, V=None, accumulate=False, F=2, fused=None):
    """dw (OIHW, [cout][cin][3][3]) (+)= weight gradient of conv3x3(x) given dout, through V = B^T x B (kept from the forward
    pass when the caller has it), dM = A dout A^T -- or, for the layers wino4_wgrad_fused names (fused=None: ask it), in one kernel
    from x and dout."""
    dev = dw.device
    B, H, W = x.B, x.H, x.W
    if F == 4 and V is None and (wino4_wgrad_fused(cin, cout, B, H, W) if fused is None else fused):
        bn = getattr(x, "bn", None)
        ws = _ws(query("mopa_wino4_wgrad_fused_workspace_bytes", B, H, W, cin, cout), dev)
        call("mopa_wino4_wgrad_fused", x.p, x.ld, ptr(bn[0]) if bn is not None else None, bn[1] if bn is not None else 1,
             bn[2] if bn is not None else 0, dout.p, dout.ld, B, H, W, cin, cout, ptr(dw), int(accumulate) | 2, ptr(ws), ws.numel(), stream())
        return
    T, NP = B * ((H + F - 1) // F) * ((W + F - 1) // F), (F + 2) ** 2
    sfx = "" if F == 2 else "4"
    dM = torch.empty(NP * T * cout, dtype=torch.float32, device=dev)
    if V is None:
        V = torch.empty(NP * T * cin, dtype=torch.float32, device=dev)
        if hasattr(x, "bn"):   # a deferred BatchNorm (LazyImg): applied on the way in, as the forward pass did
            call("mopa_wino4_input_bn", x.p, x.ld, B, H, W, cin, ptr(x.bn[0]), x.bn[1], x.bn[2], ptr(V), stream())
        else:
            call(f"mopa_wino{sfx}_input", x.p, x.ld, B, H, W, cin, ptr(V), stream())
    call(f"mopa_wino{sfx}_dout", dout.p, dout.ld, B, H, W, cout, ptr(dM), stream())
    ws = _ws(query(f"mopa_wino{sfx}_wgrad_workspace_bytes", T, cin, cout), dev)
    call(f"mopa_wino{sfx}_bwd_weight", ptr(V), ptr(dM), T, cin, cout, ptr(dw), int(accumulate) | 2, ptr(ws), ws.numel(), stream())


_wgrad_streams = {}
# MOPA_WGRAD_STREAM: "1" on, "0" off, unset = on -- except under a torch.distributed process group whose backend is not RCCL.
# Every extra stream is another hardware queue.  Two ranks SHARING one device (only possible over gloo: the multi-process
# plumbing test on a 1-GPU box) with three streams each fall off a cliff (3.4 s per step instead of 0.19); two per rank behave.
# RCCL ("nccl") refuses two ranks on one device, so under it a rank owns its GPU: measured with one rank and the real
# collectives (MOPA_FORCE_COLLECTIVES=1), three compute streams + RCCL's own run the joint step at 242.9
# scans/s against 227.2 with two -- the third stream stays on there.
# (The same stream for the sparse-conv weight gradients of 3D-only training was measured too: 1347 -> 1290 scans/s -- that step
# is host-paced and the stream switches cost more than the overlap returns: not kept.)
WGRAD_STREAM = os.environ.get("MOPA_WGRAD_STREAM", "1") != "0"   # (a caller may switch it per pass)


def _shared_device_group():
    """True under a process group that may put several ranks on one device (any backend but RCCL)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    if dist.get_world_size() == 1 and os.environ.get("MOPA_FORCE_COLLECTIVES") != "1":
        return False
    return dist.get_backend() != "nccl"


def wgrad_stream(dev):
    """The stream the weight gradients of the 2D convolutions run on (None: the current stream)."""
    if not WGRAD_STREAM or (os.environ.get("MOPA_WGRAD_STREAM") != "1" and _shared_device_group()):
        return None
    key = (torch.device(dev).index, stream())   # one per (device, consumer stream)
    st = _wgrad_streams.get(key)
    if st is None:
        st = _wgrad_streams[key] = torch.cuda.Stream(device=dev)
    return st


_on_events = {}   # side stream -> the event that orders it behind the consumer stream (re-recorded at every hand-over)
_REC_EVENTS = (0, 0)   # while a command list is recorded: raw handles of its (fork, join) events (Graph2D._record)


class _on:
    """`with _on(side, *tensors)`: run the body on `side`, ordered behind everything queued on the current stream so far; the
    tensors (allocated from the current stream's pool, touched by `side`, and FREED before the streams are joined again) are
    recorded for the caching allocator.  Tensors that outlive the join need no record: activations and transformed inputs on the
    tape, parameter gradients -- they are released after Net2DFunction.backward has queued join_wgrad_stream, and whatever
    re-uses their memory is queued behind that.  side = None: no-op."""

    def __init__(self, side, *tensors):
        self.side, self.tensors = side, tensors

    def __enter__(self):
        if self.side is None:
            return
        if _lib.RECORDER is not None:   # a pass is being recorded as a command list: the hand-over is a command too
            _lib.RECORDER.event_record(_REC_EVENTS[0], stream())
            _lib.RECORDER.stream_wait(self.side.cuda_stream, _REC_EVENTS[0])
        ev = _on_events.get(self.side)
        if ev is None:
            ev = _on_events[self.side] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.side.wait_event(ev)
        for t in self.tensors:
            if torch.is_tensor(t):
                t.record_stream(self.side)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.side is not None:
            self.ctx.__exit__(*exc)
        return False


def join_wgrad_stream(dev):
    """Order the current stream behind the weight-gradient stream (end of a backward pass)."""
    ws = wgrad_stream(dev)
    if ws is not None:
        if _lib.RECORDER is not None:
            _lib.RECORDER.event_record(_REC_EVENTS[1], ws.cuda_stream)
            _lib.RECORDER.stream_wait(stream(), _REC_EVENTS[1])
        torch.cuda.current_stream().wait_stream(ws)


# ------------------------------------------------------------------------------------------------ conv wrappers
def forward_role(cin, cout, k, s, p, B, H, W, keep_v):
    """(does the weight gradient of this forward pass want V again?, the role the forward convolution is dispatched under).  V is not
    wanted when the weight gradient runs in one kernel from x and dY (wino4_wgrad_fused): the training forward pass then keeps nothing
    and is dispatched like a forward pass without gradients ("fwd_eval")."""
    a = (cin, cout, k, s, p, B, H, W)
    F = wino_tile(*a, "fwd")
    v_wanted = bool(keep_v and F == 4 and wino_tile(*a, "wgrad") == 4 and wino_wgrad_eligible(*a) and not wino4_wgrad_fused(cin, cout, B, H, W))
    return v_wanted, ("fwd" if v_wanted or (keep_v and F != 4) else "fwd_eval")


class ConvOp:
    """Conv2d(k, stride s, padding p) in NHWC through the implicit-GEMM kernels (fwd, dgrad, wgrad)."""

    def __init__(self, weight, bias, k, s, p):
        self.w, self.b, self.k, self.s, self.p = weight, bias, k, s, p
        self.O, self.I = weight.shape[0], weight.shape[1]

    def out_hw(self, H, W):
        return (H + 2 * self.p - self.k) // self.s + 1, (W + 2 * self.p - self.k) // self.s + 1

    def _fwd_geom(self, x: Img, out: Img):
        return _geom(B=x.B, IH=x.H, IW=x.W, OHl=out.H, OWl=out.W, OHa=out.H, OWa=out.W, IS=self.s, IY0=-self.p,
                     IX0=-self.p, TH=self.k, TW=self.k, KWF=self.k, Cin=self.I, Cout=self.O, ld_in=x.ld, ld_out=out.ld)

    def takes_lazy(self, B, H, W, training):
        """May the input be a LazyImg?  The forward pass must be F(4x4) (the input transform applies the BatchNorm) and, in
        training, the weight gradient must run on the V kept from it -- nothing else of this layer reads the input."""
        a = (self.I, self.O, self.k, self.s, self.p, B, H, W)
        return (wino_tile(*a, "fwd") == 4
                and (not training or (wino_tile(*a, "wgrad") == 4 and wino_wgrad_eligible(*a))))

    def forward(self, x: Img, out: Img, keep_v: bool = False):
        """-> the transformed input V when the Winograd path ran and the weight gradient will want it again (training)."""
        F = wino_tile(self.I, self.O, self.k, self.s, self.p, x.B, x.H, x.W, "fwd")
        a = (self.I, self.O, self.k, self.s, self.p, x.B, x.H, x.W)
        # does the weight gradient of this pass want V again?  Not when it runs in one kernel from x and dY (wino4_wgrad_fused): the
        # training forward pass then keeps nothing, like a forward pass without gradients
        v_wanted, drole = forward_role(*a, keep_v)
        lazy = getattr(x, "bn", None)
        if lazy is not None and not self.takes_lazy(x.B, x.H, x.W, keep_v):
            raise RuntimeError("ConvOp.forward: this layer cannot consume a deferred BatchNorm (ask takes_lazy first)")
        if F:
            V = wino_conv(x.p, x.ld, x.B, x.H, x.W, self.I, self.O,
                          wino_weight_cached(self.w, False, F, wino4_layout(self.I, self.O, x.B, x.H, x.W, drole) if F == 4 else 0), self.b, out.p,
                          out.ld, F=F, bn_in=lazy, role=drole, want_v=v_wanted)
            if F == 4 and not v_wanted:
                return None
            same = F == wino_tile(*a, "wgrad")   # V serves the weight gradient
            return V if V is not None and keep_v and same and wino_wgrad_eligible(*a) else None
        wl = relayout_cached(self.w, (self.k, self.k, self.I, self.O), self.O, self.I, self.k, self.k, 0)
        igemm(x.p, wl, self.b, out.p, self._fwd_geom(x, out))

    def backward(self, x: Img, dout: Img, dx: Img | None, dw: torch.Tensor, db, acc_dx: bool, acc_params: bool = False, V=None,
                 wgrad_side: bool = False):
        """wgrad_side: the caller joins the weight-gradient stream itself (join_wgrad_stream) after the whole backward pass."""
        dev = self.w.device
        k, s, p = self.k, self.s, self.p
        if hasattr(x, "bn") and V is None and not (wino_wgrad_eligible(self.I, self.O, k, s, p, x.B, x.H, x.W)
                                                   and wino_tile(self.I, self.O, k, s, p, x.B, x.H, x.W, "wgrad") == 4):
            raise RuntimeError("ConvOp.backward: the input was a deferred BatchNorm and this layer's weight gradient cannot apply it")
        # weight gradient: the split-K reduction writes (or accumulates into) the OIHW gradient tensor directly.  It depends on x
        # and dout only and nothing downstream of this layer waits for it: it runs on the weight-gradient stream, beside the
        # backward-data chain that the rest of the backward pass is waiting for (Net2DFunction.backward joins the stream).
        ws = wgrad_stream(dev) if wgrad_side else None
        with _on(ws, dout.t):   # (dout is dropped by the caller before the join; x, V, dw, db outlive it)
            if wino_wgrad_eligible(self.I, self.O, k, s, p, x.B, x.H, x.W):
                wino_wgrad(x, dout, self.I, self.O, dw, V, accumulate=acc_params, F=wino_tile(self.I, self.O, k, s, p, x.B, x.H, x.W, "wgrad"))
            else:
                wgrad(x.p, dout.p, ptr(dw), self._fwd_geom(x, dout), dev, accumulate=acc_params, oihw=True)
            if db is not None:
                colsum(dout, db, accumulate=acc_params)
        if dx is None:
            return
        F = wino_tile(self.O, self.I, k, s, p, x.B, x.H, x.W, "dgrad")   # backward-data of a stride-1 3x3 conv is one, too
        if F:
            wino_conv(dout.p, dout.ld, x.B, x.H, x.W, self.O, self.I,
                      wino_weight_cached(self.w, True, F, wino4_layout(self.O, self.I, x.B, x.H, x.W, "dgrad") if F == 4 else 0), None, dx.p,
                      dx.ld, acc_dx, F=F, role="dgrad", want_v=False)
            return
        wt = relayout_cached(self.w, (k, k, self.O, self.I), self.O, self.I, k, k, 1)
        if s == 1:
            g = _geom(B=x.B, IH=dout.H, IW=dout.W, OHl=x.H, OWl=x.W, OHa=x.H, OWa=x.W, IY0=p, IX0=p, IDY=-1, IDX=-1,
                      TH=k, TW=k, KWF=k, Cin=self.O, Cout=self.I, ld_in=dout.ld, ld_out=dx.ld)
            igemm(dout.p, wt, None, dx.p, g, acc_dx)
            return
        assert s == 2
        if not acc_dx:
            call("mopa_zero_rows", dx.p, dx.ld, dx.rows, dx.C, stream())   # (not a torch kernel: the pass may be a recorded command list)
        for ph in range(2):
            for pw in range(2):
                khs = [kh for kh in range(k) if (ph + p - kh) % 2 == 0]
                kws = [kw for kw in range(k) if (pw + p - kw) % 2 == 0]
                ohl, owl = (x.H - ph + 1) // 2, (x.W - pw + 1) // 2
                if not khs or not kws or ohl <= 0 or owl <= 0:
                    continue
                g = _geom(B=x.B, IH=dout.H, IW=dout.W, OHl=ohl, OWl=owl, OHa=x.H, OWa=x.W, OS=2, OOY=ph, OOX=pw,
                          IY0=(ph + p - khs[0]) // 2, IX0=(pw + p - kws[0]) // 2, IDY=-1, IDX=-1, TH=len(khs),
                          TW=len(kws), KH0=khs[0], KW0=kws[0], KS=2, KWF=k, Cin=self.O, Cout=self.I, ld_in=dout.ld,
                          ld_out=dx.ld)
                igemm(dout.p, wt, None, dx.p, g, True)


class ConvTOp:
    """ConvTranspose2d(k=2, s=2) + bias: 4 output-parity classes of a 1x1 conv."""

    def __init__(self, weight, bias):
        self.w, self.b = weight, bias
        self.I, self.O = weight.shape[0], weight.shape[1]

    def _geom(self, x: Img, out: Img, ky, kx):
        return _geom(B=x.B, IH=x.H, IW=x.W, OHl=x.H, OWl=x.W, OHa=out.H, OWa=out.W, OS=2, OOY=ky, OOX=kx, TH=1, TW=1,
                     KH0=ky, KW0=kx, KWF=2, Cin=self.I, Cout=self.O, ld_in=x.ld, ld_out=out.ld)

    def forward(self, x: Img, out: Img):
        wl = relayout_cached(self.w, (2, 2, self.I, self.O), self.O, self.I, 2, 2, 2)
        for ky in range(2):
            for kx in range(2):
                igemm(x.p, wl, self.b, out.p, self._geom(x, out, ky, kx))

    def backward(self, x: Img, dout: Img, dx: Img, dw, db, acc_params: bool = False, wgrad_side: bool = False):
        dev = self.w.device
        with _on(wgrad_stream(dev) if wgrad_side else None, dout.t):   # see ConvOp.backward
            dwl = torch.empty(2, 2, self.I, self.O, dtype=torch.float32, device=dev)
            for ky in range(2):
                for kx in range(2):
                    g = self._geom(x, dout, ky, kx)
                    g[17], g[18] = 0, 0  # KH0/KW0: the per-class launch writes a single-tap slab
                    wgrad(x.p, dout.p, ptr(dwl, (ky * 2 + kx) * self.I * self.O), g, dev)
            relayout(dwl, dw, self.O, self.I, 2, 2, 2, inverse=True, accumulate=acc_params)
            colsum(dout, db, accumulate=acc_params)
        wt = relayout_cached(self.w, (2, 2, self.O, self.I), self.O, self.I, 2, 2, 3)
        g = _geom(B=x.B, IH=dout.H, IW=dout.W, OHl=x.H, OWl=x.W, OHa=x.H, OWa=x.W, IS=2, TH=2, TW=2, KWF=2,
                  Cin=self.O, Cout=self.I, ld_in=dout.ld, ld_out=dx.ld)
        igemm(dout.p, wt, None, dx.p, g, False)


def colsum(x: View, out: torch.Tensor, accumulate=False):
    wsb = query("mopa_colsum_workspace_bytes", x.rows, x.C)
    ws = _ws(wsb, out.device)
    call("mopa_colsum", x.p, x.ld, x.rows, x.C, ptr(out), int(accumulate), ptr(ws), ws.numel(), stream())


# A/B switch: bn1 of a ResNet block is applied inside conv2's input transform instead of being written out (LazyImg)
DEFER_BN = os.environ.get("MOPA_DEFER_BN", "1") != "0"
DEFER_UP_BN = os.environ.get("MOPA_DEFER_UP_BN", "1") != "0"   # ... and the decoder's up-convolution BatchNorms inside the join's consumer
DEFER_STEM_BN = os.environ.get("MOPA_DEFER_STEM_BN", "1") != "0"   # ... and the stem's inside its two readers (max-pool, full-resolution join)
# A/B switch: the stem BatchNorm's backward apply inside the stem's weight gradient (mopa_stem_bwd_weight_bn: dx is never written)
# (the stem's weight gradient with the BatchNorm backward formed in its loader exists as an MFMA kernel only: off with MOPA_CONV2D_MFMA=0)
STEM_BN_FUSED_BWD = os.environ.get("MOPA_STEM_BN_FUSED_BWD", "1") != "0" and os.environ.get("MOPA_CONV2D_MFMA", "1") != "0"


def bn_fwd(x: View, y: View, P, name, act, res, training, stats):
    """Returns None, or -- synchronised BatchNorm (mopa_amd.syncbn) in training mode -- the gathered moments for bn_bwd."""
    if training and syncbn.active():
        return syncbn.fwd(x, y, P[name + ".weight"], P[name + ".bias"], P[name + ".running_mean"], P[name + ".running_var"],
                          BN_MOMENTUM, BN_EPS, 0.0, act, res, stats)
    wsb = query("mopa_bnrelu_rows_workspace_bytes", x.rows, x.C)
    ws = _ws(wsb, x.t.device)
    call("mopa_bn_act_fwd", x.p, x.ld, y.p, y.ld, x.rows, x.C, ptr(P[name + ".weight"]), ptr(P[name + ".bias"]),
         ptr(P[name + ".running_mean"]), ptr(P[name + ".running_var"]), BN_MOMENTUM, BN_EPS, 0.0, int(act),
         res.p if res is not None else None, res.ld if res is not None else 0, int(training), ptr(stats), ptr(ws),
         ws.numel(), stream())


def bn_fwd_groups(x: View, y: View | None, P, name, act, res, training, stats, G):
    """bn_fwd for G consecutive, equally sized row groups of one tensor in ONE set of launches (3 instead of 3 G): statistics, running
    updates (group 0 first) and the apply per group, bit-identical to G calls of bn_fwd on the row ranges.  stats: (G, 4, C).
    y = None: no apply pass (the consumer applies stats while it reads x: LazyImg)."""
    n = x.rows // G
    wsb = query("mopa_bnrelu_rows_workspace_bytes", x.rows, x.C)
    ws = _ws(wsb, x.t.device)
    call("mopa_bn_act_fwd_groups", x.p, x.ld, y.p if y is not None else None, y.ld if y is not None else 0, x.rows, x.C, G, n, 2 * n,
         ptr(P[name + ".weight"]), ptr(P[name + ".bias"]),
         ptr(P[name + ".running_mean"]), ptr(P[name + ".running_var"]), BN_MOMENTUM, BN_EPS, 0.0, int(act),
         res.p if res is not None else None, res.ld if res is not None else 0, int(training), ptr(stats), ptr(ws),
         ws.numel(), stream())


def bn_bwd_groups(dy: View, x: View, dx: View, stats, act, ymask, dres, acc_dres, training, dgamma, dbeta, G, acc_params=False):
    n = x.rows // G
    wsb = query("mopa_bnrelu_rows_bwd_workspace_bytes", x.rows, x.C)
    ws = _ws(wsb, x.t.device)
    call("mopa_bn_act_bwd_groups", dy.p, dy.ld, x.p, x.ld, dx.p, dx.ld, x.rows, x.C, G, n, 2 * n, ptr(stats), 0.0, int(act),
         ymask.p if ymask is not None else None, ymask.ld if ymask is not None else 0,
         dres.p if dres is not None else None, dres.ld if dres is not None else 0, int(acc_dres), int(training),
         ptr(dgamma), ptr(dbeta), int(acc_params), 0, ptr(ws), ws.numel(), stream())


def bn_bwd(dy: View, x: View, dx: View, stats, act, ymask, dres, acc_dres, training, dgamma, dbeta, acc_dx=False,
           acc_params=False, gathered=None):
    if gathered is not None:   # the forward pass of this layer ran with global statistics
        return syncbn.bwd(dy, x, dx, stats, 0.0, act, ymask, dres, acc_dres, dgamma, dbeta, acc_params, acc_dx, gathered)
    wsb = query("mopa_bnrelu_rows_bwd_workspace_bytes", x.rows, x.C)
    ws = _ws(wsb, x.t.device)
    call("mopa_bn_act_bwd", dy.p, dy.ld, x.p, x.ld, dx.p, dx.ld, x.rows, x.C, ptr(stats), 0.0, int(act),
         ymask.p if ymask is not None else None, ymask.ld if ymask is not None else 0,
         dres.p if dres is not None else None, dres.ld if dres is not None else 0, int(acc_dres), int(training),
         ptr(dgamma), ptr(dbeta), int(acc_params), int(acc_dx), ptr(ws), ws.numel(), stream())


# ------------------------------------------------------------------------------------------------ the backbone passes
def _group(v: Img, g, G):
    """The g-th of G equal image groups of an NHWC activation (a row range of the same buffer)."""
    if G == 1:
        return v
    n = v.rows // G
    return Img(v.t[g * n:(g + 1) * n], v.B // G, v.H, v.W, v.col, v.C)


def _backbone_forward(P, imgc, training, drop_p, drop_seed, seed_t, dev, groups=1):
    """UNetResNet34 on a contiguous fp32 (B,3,H,W) image -> (feat, tape, J).  feat: (B, Hp, Wp, 64) NHWC, the /16-padded decoder
    output; tape: what the backward pass walks; J: the join buffers.  Every launch goes to the current stream, nothing is read
    back: the pass can be recorded into a HIP graph (Graph2D).  seed_t: int64 device scalar holding the dropout seed (graph
    replays), else the seed is passed by value.

    groups = G > 1: the batch is G consecutive, equally sized image groups that the reference would have sent through the network
    in G separate calls (the source and the target batch of one xMUDA iteration, train_xmuda_mopa.py:342,426 -- no weight update
    in between).  Convolutions and pooling do not see the difference; everything that does is done per group in call order:
    BatchNorm batch statistics and the running-statistics update (group 0 first), num_batches_tracked (+G), the dropout masks
    (drop_seed = one seed per group).  The result equals the G calls; the kernels see G times the rows per launch."""
    pre = "net_2d."
    B, _, H, W = imgc.shape
    G = groups
    seeds = tuple(drop_seed) if isinstance(drop_seed, (tuple, list)) else (drop_seed,) * G
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    tape = []
    nbt = []   # BatchNorm2d.num_batches_tracked of every layer that ran: bumped together at the end (one launch, not 43)

    def bn(name, x, act=1, res=None, out=None, defer=False):
        stats = torch.empty(G, 4, x.C, dtype=torch.float32, device=dev)
        if defer:   # statistics only; the next convolution's input transform applies them
            assert act == 1 and res is None and out is None
            bn_fwd_groups(x, None, P, name, act, None, training, stats, G)
            y = LazyImg(x, stats, G)
            if training:
                nbt.append(P[name + ".num_batches_tracked"])
            tape.append(("bn", name, x, y, stats, act, None, [None] * G))
            return y
        y = out if out is not None else new_img(x.B, x.H, x.W, x.C, dev)
        if G > 1 and not (training and syncbn.active()):
            bn_fwd_groups(x, y, P, name, act, res, training, stats, G)     # one set of launches for all groups
            gathered = [None] * G
        else:
            gathered = [bn_fwd(_group(x, g, G), _group(y, g, G), P, name, act, None if res is None else _group(res, g, G), training,
                               stats[g]) for g in range(G)]
        if training:
            nbt.append(P[name + ".num_batches_tracked"])
        tape.append(("bn", name, x, y, stats, act, res, gathered))
        return y

    def conv(name, x, k, s, p, bias=False, out=None):
        op = ConvOp(P[name + ".weight"], P[name + ".bias"] if bias else None, k, s, p)
        oh, ow = op.out_hw(x.H, x.W)
        out = out if out is not None else new_img(x.B, oh, ow, op.O, dev)
        V = op.forward(x, out, keep_v=training)   # Winograd layers: the transformed input serves the weight gradient again
        tape.append(("conv", name, op, x, out, V))
        return out

    def convT(name, x, out=None):
        op = ConvTOp(P[name + ".weight"], P[name + ".bias"])
        out = out if out is not None else new_img(x.B, 2 * x.H, 2 * x.W, op.O, dev)
        op.forward(x, out)
        tape.append(("convT", name, op, x, out))
        return out

    def dropout(x, site, out=None):
        y = out if out is not None else new_img(x.B, x.H, x.W, x.C, dev)
        p = drop_p if training else 0.0
        for g in range(G):
            dropout_rows(_group(x, g, G), _group(y, g, G), p, seeds[g], None if seed_t is None else seed_t[g:], site)
        tape.append(("dropout", site, x, y, p))
        return y

    # ---- stem (resnet34_unet.py:144-148): conv1 7x7 s1 p3 on the /16-padded image, bn1, relu, maxpool
    x4 = torch.empty(B, Hp + 6, Wp + 8, 4, dtype=torch.float32, device=dev)
    call("mopa_img_to_nhwc4", ptr(imgc), B, H, W, Hp, Wp, ptr(x4), stream())
    w1 = torch.empty(7, 2, 16, 64, dtype=torch.float32, device=dev)
    call("mopa_conv2d_stem_relayout", ptr(P[pre + "conv1.weight"]), ptr(w1), 64, 0, 0, stream())
    J = {}  # join buffers: [skip | upsampled]
    J[0] = torch.empty(B * Hp * Wp, 128, dtype=torch.float32, device=dev)
    # The stem's BatchNorm + ReLU has two readers, the max-pool and the full-resolution decoder convolution (through the join buffer).
    # When that convolution normalises on the way in (see the decoder below) the stem convolution writes its RAW output into the left
    # half of the join buffer, the BatchNorm computes statistics only and the max-pool applies them to its windows: no apply pass over
    # the largest activation of the network (598 MB read + written at 16 x 304 x 480), no second copy of it.
    lazy_stem = (DEFER_BN and DEFER_UP_BN and DEFER_STEM_BN and not (training and syncbn.active())
                 and ConvOp(P[pre + "dec_conv_stage1.weight"], None, 3, 1, 1).takes_lazy(B, Hp, Wp, training))
    c1 = Img(J[0], B, Hp, Wp, 0, 64) if lazy_stem else new_img(B, Hp, Wp, 64, dev)
    stem_g = _geom(B=B, IH=Hp + 6, IW=Wp + 8, OHl=Hp, OWl=Wp, OHa=Hp, OWa=Wp, IDX=4, TH=7, TW=2, KWF=2, Cin=16,
                   Cout=64, ld_in=4, ld_out=c1.ld)
    igemm(ptr(x4), w1, None, c1.p, stem_g)
    tape.append(("stem", x4, c1, stem_g))
    H2, W2 = Hp // 2, Wp // 2
    x = new_img(B, H2, W2, 64, dev)
    amax = torch.empty(B * H2 * W2 * 64, dtype=torch.uint8, device=dev)
    if lazy_stem:
        skip0 = bn(pre + "bn1", c1, defer=True)
        call("mopa_maxpool3x3s2_fwd_bn", c1.p, c1.ld, B, Hp, Wp, 64, ptr(skip0.bn[0]), G, x.p, x.ld, ptr(amax), stream())
    else:
        skip0 = bn(pre + "bn1", c1, out=Img(J[0], B, Hp, Wp, 0, 64))
        call("mopa_maxpool3x3s2_fwd", skip0.p, skip0.ld, B, Hp, Wp, 64, x.p, x.ld, ptr(amax), stream())
    tape.append(("maxpool", skip0, x, amax))
    # ---- encoder stages
    for li, (lname, c, nblocks, stride) in enumerate(LAYERS):
        for b in range(nblocks):
            q = f"{pre}{lname}.{b}."
            s = stride if b == 0 else 1
            has_ds = (q + "downsample.0.weight") in P
            z1 = conv(q + "conv1", x, 3, s, 1)
            defer = (DEFER_BN and not (training and syncbn.active())
                     and ConvOp(P[q + "conv2.weight"], None, 3, 1, 1).takes_lazy(z1.B, z1.H, z1.W, training))
            y1 = bn(q + "bn1", z1, defer=defer)
            z = conv(q + "conv2", y1, 3, 1, 1)
            idt = bn(q + "downsample.1", conv(q + "downsample.0", x, 1, s, 0), act=0) if has_ds else x
            last = b == nblocks - 1
            out = None
            if last and lname in ("layer1", "layer2"):
                lvl = li + 1
                J[lvl] = torch.empty(z.rows, 2 * c, dtype=torch.float32, device=dev)
                out = Img(J[lvl], z.B, z.H, z.W, 0, c)
            tape.append(("block_in", x))
            x = bn(q + "bn2", z, act=1, res=idt, out=out)
        if lname == "layer3":   # dropout, then the result is skip3 AND layer4's input (:153-155)
            J[3] = torch.empty(x.rows, 2 * c, dtype=torch.float32, device=dev)
            x = dropout(x, 0, out=Img(J[3], x.B, x.H, x.W, 0, c))
        if lname == "layer4":
            x = dropout(x, 1)
    # ---- decoder (:165-182): ConvT+BN+ReLU into the right half of the join buffer, conv3x3 on [skip | up]
    for stage, lvl in (("5", 3), ("4", 2), ("3", 1), ("2", 0)):
        tname = f"{pre}dec_t_conv_stage{stage}."
        cj = J[lvl].shape[1] // 2
        cname = pre + "dec_conv_stage1" if lvl == 0 else f"{pre}dec_conv_stage{int(stage) - 1}.0"
        # The up-convolution's BatchNorm + ReLU on the way INTO the decoder convolution (DEFER_BN): the transposed convolution writes
        # its raw output into the right half of the join buffer, the BatchNorm computes statistics only, and the convolution's F(4x4)
        # input transform normalises channels [cj, 2 cj) while it reads the buffer (the skip half is non-negative already: it passes
        # through relu(1 x + 0) unchanged) -- no apply pass over the up-sampled tensor (598 MB read + written at 16 x 304 x 480).
        lazy_up = (DEFER_BN and DEFER_UP_BN and not (training and syncbn.active())
                   and ConvOp(P[cname + ".weight"], None, 3, 1, 1).takes_lazy(x.B, 2 * x.H, 2 * x.W, training))
        if lazy_up:
            right = Img(J[lvl], x.B, 2 * x.H, 2 * x.W, cj, cj)
            up_raw = convT(tname + "0", x, out=right)
            ylazy = bn(tname + "1", up_raw, defer=True)
            if lvl == 0 and lazy_stem:   # both halves are raw: [stem conv | up-convolution], one statistics tensor over the 128 channels
                st2 = torch.empty(G, 4, 2 * cj, dtype=torch.float32, device=dev)   # [stem | up-convolution] statistics side by side
                for half, src in enumerate((skip0.bn[0], ylazy.bn[0])):
                    call("mopa_copy_rows", ptr(src), cj, ptr(st2, half * cj), 2 * cj, G * 4, cj, stream())
                joined = LazyImg(Img(J[lvl], up_raw.B, up_raw.H, up_raw.W, 0, 2 * cj), st2, G)
            else:
                joined = LazyImg(Img(J[lvl], up_raw.B, up_raw.H, up_raw.W, 0, 2 * cj), ylazy.bn[0], G, c0=cj)
        else:
            up_raw = convT(tname + "0", x)
            bn(tname + "1", up_raw, out=Img(J[lvl], up_raw.B, up_raw.H, up_raw.W, cj, cj))
            joined = Img(J[lvl], up_raw.B, up_raw.H, up_raw.W, 0, 2 * cj)
        tape.append(("join", lvl, cj, lazy_up, lvl == 0 and lazy_stem))
        if lvl == 0:
            x = conv(pre + "dec_conv_stage1", joined, 3, 1, 1, bias=True)
        else:
            x = bn(cname[:-1] + "1", conv(cname, joined, 3, 1, 1, bias=True))
    for i in range(0, len(nbt), 64):   # one launch per 64 counters (43 BatchNorm layers: one)
        tab = np.zeros(64, np.int64)
        tab[:len(nbt[i:i + 64])] = [t.data_ptr() for t in nbt[i:i + 64]]
        call("mopa_add_i64_many", tab.ctypes.data, len(nbt[i:i + 64]), G, stream())
    return x, tape, J   # x: (B, Hp, Wp, 64); the crop to (H, W) is implicit in the heads' indexing (:185-186)


def dropout_rows(x: View, y: View, p, seed, seed_t, site):
    if seed_t is not None:
        call("mopa_dropout_rows_dseed", x.p, x.ld, y.p, y.ld, x.rows, x.C, float(p), ptr(seed_t), site, stream())
    else:
        call("mopa_dropout_rows", x.p, x.ld, y.p, y.ld, x.rows, x.C, float(p), seed * 2 + site, stream())


def _backbone_backward(P, sink, tape, J, feat, dfeat, training, drop_seed, seed_t, want_dimg, H, W, groups=1):
    """Walk the tape backwards from dfeat = d(loss)/d(feat); parameter gradients go through `sink`.  -> d(loss)/d(img) (fp32,
    (B,3,H,W)) if want_dimg else None.  Ends with the weight-gradient stream joined.  Recordable like _backbone_forward.
    groups: as in _backbone_forward (BatchNorm backward and dropout per group; the parameter gradients of the groups add up, as
    the G backward passes of the reference accumulate into .grad)."""
    dev = feat.t.device
    G = groups
    seeds = tuple(drop_seed) if isinstance(drop_seed, (tuple, list)) else (drop_seed,) * G
    pre = "net_2d."
    B, Hp, Wp = feat.B, feat.H, feat.W
    gmap = {}
    dimg = None

    def key(v):   # (a LazyImg shares its BatchNorm input's buffer: its gradient is another tensor)
        return (v.t.data_ptr(), v.col, v.C) + ((1,) if hasattr(v, "bn") else ())

    def like(v: Img, zero=False):
        return new_img(v.B, v.H, v.W, v.C, dev, zero=zero)

    gmap[key(feat)] = dfeat
    dJ = {}  # gradient buffers of the join tensors (full width)
    for rec in reversed(tape):
        kind = rec[0]
        if kind == "bn":
            _, name, x, y, stats, act, res, gathered = rec
            dy = gmap.pop(key(y))
            dres = None
            acc_dres = False
            if res is not None:
                k = key(res)
                if k in gmap:
                    dres, acc_dres = gmap[k], True
                else:
                    dres = like(res)
                    gmap[k] = dres
            (dg, db), pacc = sink.take(name + ".weight", name + ".bias")
            if (STEM_BN_FUSED_BWD and name == pre + "bn1" and not want_dimg and res is None and act == 1
                    and all(gt is None for gt in gathered) and x.C == 64):
                # the stem's BatchNorm: its input gradient has ONE reader, the stem's weight gradient, which forms it from (dy, x)
                # itself -- sums + parameter gradients here, no apply pass, no dx tensor
                n = x.rows // G
                coef = torch.empty(G, 2, x.C, dtype=torch.float32, device=dev)
                ws = _ws(query("mopa_bnrelu_rows_workspace_bytes", x.rows, x.C), dev)
                call("mopa_bn_bwd_sums_groups", dy.p, dy.ld, x.p, x.ld, x.rows, x.C, G, n, 2 * n, ptr(stats), 0.0, int(act), None, 0,
                     ptr(dg), ptr(db), int(pacc), ptr(coef), ptr(ws), ws.numel(), stream())
                gmap[key(x)] = ("bn", dy, x, stats, coef)
                continue
            dx = like(x)
            gmap[key(x)] = dx
            if G > 1 and all(gt is None for gt in gathered):
                bn_bwd_groups(dy, x, dx, stats, act, y if res is not None else None, dres, acc_dres, training, dg, db, G, acc_params=pacc)
            else:
                for g in range(G):
                    bn_bwd(_group(dy, g, G), _group(x, g, G), _group(dx, g, G), stats[g], act,
                           _group(y, g, G) if res is not None else None, None if dres is None else _group(dres, g, G), acc_dres, training,
                           dg, db, acc_params=pacc or g > 0, gathered=gathered[g])
        elif kind == "conv":
            _, name, op, x, out, V = rec
            dout = gmap.pop(key(out))
            k = key(x)
            acc = k in gmap
            dx = gmap[k] if acc else like(x)
            gmap[k] = dx
            pg, pacc = sink.take(*([name + ".weight"] + ([name + ".bias"] if op.b is not None else [])))
            op.backward(x, dout, dx, pg[0], pg[1] if op.b is not None else None, acc, acc_params=pacc, V=V, wgrad_side=True)
        elif kind == "convT":
            _, name, op, x, out = rec
            dout = gmap.pop(key(out))
            dx = like(x)
            gmap[key(x)] = dx
            (dw, db), pacc = sink.take(name + ".weight", name + ".bias")
            op.backward(x, dout, dx, dw, db, acc_params=pacc, wgrad_side=True)
        elif kind == "join":
            _, lvl, cj, lazy_up, lazy_left = rec
            lz = (1,) if lazy_up else ()   # (the join was consumed as a LazyImg: see key())
            full = gmap.pop((J[lvl].data_ptr(), 0, 2 * cj) + lz)
            dJ[lvl] = full
            if DEBUG is not None:
                DEBUG[f"dJ{lvl}"] = full.t.clone()
            gmap[(J[lvl].data_ptr(), 0, cj) + ((1,) if lazy_left else ())] = Img(full.t, full.B, full.H, full.W, 0, cj)
            gmap[(J[lvl].data_ptr(), cj, cj) + lz] = Img(full.t, full.B, full.H, full.W, cj, cj)
        elif kind == "dropout":
            _, site, x, y, p = rec
            dy = gmap.pop(key(y))
            dx = like(x)
            gmap[key(x)] = dx
            for g in range(G):
                dropout_rows(_group(dy, g, G), _group(dx, g, G), p, seeds[g], None if seed_t is None else seed_t[g:], site)
        elif kind == "maxpool":
            _, x, y, amax = rec
            dy = gmap.pop(key(y))
            k = key(x)
            acc = k in gmap
            dx = gmap[k] if acc else like(x)
            gmap[k] = dx
            call("mopa_maxpool3x3s2_bwd", dy.p, dy.ld, ptr(amax), x.B, x.H, x.W, x.C, dx.p, dx.ld, int(acc), stream())
        elif kind == "stem":
            _, x4, c1, g = rec
            dout = gmap.pop(key(c1))
            (dw,), pacc = sink.take(pre + "conv1.weight")
            lazy = dout if isinstance(dout, tuple) else None   # ("bn", dy, x, stats, coef): the BatchNorm above left its apply to us
            # (the weight-gradient stream reads dy and, fused, the small coef tensor: both are dropped here before the streams are joined)
            with _on(wgrad_stream(dev), (lazy[1] if lazy else dout).t, *((lazy[4],) if lazy else ())):
                dwl = torch.empty(7, 2, 16, 64, dtype=torch.float32, device=dev)
                if lazy:
                    _, bdy, bx, bstats, bcoef = lazy
                    ws = _ws(query("mopa_conv2d_wgrad_workspace_bytes", ctypes.addressof(g)), dev)
                    call("mopa_stem_bwd_weight_bn", ptr(x4), bdy.p, bdy.ld, bx.p, bx.ld, ptr(bstats), ptr(bcoef), G, int(training), ptr(dwl),
                         ctypes.addressof(g), 0, ptr(ws), ws.numel(), stream())
                else:
                    if g[24] != dout.ld:   # (forward wrote into a wider buffer; the gradient tensor has its own row stride)
                        g = (ctypes.c_int32 * 25)(*g)
                        g[24] = dout.ld
                    wgrad(ptr(x4), dout.p, ptr(dwl), g, dev)
                call("mopa_conv2d_stem_relayout", ptr(dwl), ptr(dw), 64, 1, int(pacc), stream())
            if want_dimg:   # gradient w.r.t. the image itself (not asked for by MoPA's training)
                dimg = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
                call("mopa_stem_dgrad_image", dout.p, dout.ld, B, Hp, Wp, H, W, ptr(P[pre + "conv1.weight"]), ptr(dimg), stream())
        elif kind == "block_in":
            pass
    join_wgrad_stream(dev)   # the weight gradients are complete for whatever the caller queues next
    return dimg


# ------------------------------------------------------------------------------------------------ HIP-graph replay of the backbone
# The backbone is ~200 launches forward and ~260 backward through the C ABI, every one of them with arguments that depend only on
# (B, H, W), the parameter addresses and the mode flags: for a training loop on fixed-size crops the whole launch sequence is a
# constant.  Enqueuing it from Python costs ~10 ms per forward + backward (6.6 us per launch + the interpreter around it,
# profiles/host_profile_2d.py) -- as long as the GPU needs for a 2-image batch -- and leaves gaps between the short kernels of the
# deep layers.  Graph2D records the forward launches once into a HIP graph and the backward launches into a second one (the fork to
# the weight-gradient stream and its join included) and replays them: same kernels, same order, same addresses, bit-identical
# results (tests/test_gpu_2d.py).  What stays outside the graphs is whatever depends on the number of points: the two heads, their
# backward passes and the losses.
#
# Rules that keep a replay equal to the eager pass:
#   * recorded on the SECOND call of a (shape, mode, stream) key -- the first runs eagerly and leaves every kernel attribute, weight
#     form and workspace behind; while recording, a weight form is the tensor the replaying stream's cache owns (_CAPTURE above)
#     and Graph2D.forward refreshes the stale ones in one launch before each replay, exactly as the eager pass does;
#   * inputs are copied into fixed buffers (image, dropout seed, d(feat)); parameters, BatchNorm buffers and attached gradient
#     buffers are addressed in place, and their addresses are checked before every replay (changed -> graphs dropped, eager);
#   * the activations a forward replay leaves are the ones the next backward reads: a second forward before that backward (two
#     batches through the network, then one loss) runs eagerly instead;
#   * the backward graph accumulates into the attached .grad buffers (FlatAdam's flat buffer); a parameter without an attached
#     gradient -> the backward walks the recorded tape eagerly (autograd gets its gradient tensors as usual).
# OFF by default (MOPA_GRAPH_2D=1 switches it on), because it does not pay on this platform -- measured on an MI355X, ROCm 7.2
# (profiles/graph_probe.py; gpurun_out/g2d_*): the 2D branch ALONE, 8 x 302 x 480, forward + backward: 26.3 ms eager -> 23.9 ms
# replayed (the launch gaps of the deep layers close), but hipGraphLaunch itself keeps the host for 4.6 ms (2 images) to 11 ms (8)
# per replay -- the runtime walks the ~460 nodes and enqueues one packet each -- and while it does, the host cannot feed the 3D
# branch's stream: inside the joint step the replayed backbone LOSES, 319 -> 300 scans/s (nuScenes shape), 227 -> 190 (MoPA
# iteration), 131-152 -> 135-142 (SemanticKITTI shape, within box noise); with source + target images in one pass (bn_groups = 2,
# where the host has twice the slack): 334 -> 311, 240 -> 210, 188 -> 175.  Kept as an option for 2D-only loops and for a runtime
# whose graph launch is cheap.  Training mode with gradients enabled only; not under synchronised BatchNorm (its collectives are
# issued from Python between the kernels).
GRAPH_2D = os.environ.get("MOPA_GRAPH_2D", "0") == "1"
# Round 5: the same recording replayed as a COMMAND LIST instead of a hipGraph -- mopa_exec_replay (csrc/exec2d.hip) walks the
# recorded entry points in one native call: the HIP runtime's launch cost per kernel (~3.5 us) instead of the interpreter's
# (~16 us), no graph launch.  The capture is still made (it pins every address of the pass in a private pool and applies the
# allocator's cross-stream rules), its hipGraph is simply never launched.  Default ON; MOPA_NATIVE_2D=0 = the Python walk.
NATIVE_2D = os.environ.get("MOPA_NATIVE_2D", "1") != "0"   # (MOPA_GRAPH_2D=1 takes precedence: hipGraph replay)
GRAPH_2D_MAX_KEYS = 4        # distinct (shape, mode, stream) keys recorded per network; further ones run eagerly
GRAPH_STATS = {"recorded": 0, "forward_replays": 0, "backward_replays": 0, "eager_busy": 0, "eager_backward": 0, "dropped": 0}


class _Token:
    __slots__ = ("__weakref__",)


class Graph2D:
    """The recorded forward and backward pass of the backbone for one (B, H, W, training, dropout p, stream)."""

    def __init__(self, B, H, W, dev, groups=1):
        self.B, self.H, self.W, self.dev, self.groups = B, H, W, dev, groups
        self.calls = 0            # eligible passes seen with this key
        self.eager_backward_seen = False   # recording starts once a whole eager forward + backward has run with this key
        self.fwd = self.bwd = None
        self.failed = False
        self.pending = None       # weak reference to the token of the forward whose backward has not run yet
        self.generation = 0       # bumped by every forward replay: a backward of an older forward must not read the activations

    def busy(self):
        return self.pending is not None and self.pending() is not None

    # -- recording
    def _record(self, fn, pool):
        global _CAPTURE, _REC_EVENTS
        g = torch.cuda.CUDAGraph()
        main = stream()
        _CAPTURE = main
        rec = None
        if NATIVE_2D and not GRAPH_2D:
            if getattr(self, "events", None) is None:   # the command list's fork / join events between the two streams (created and
                self.events = (torch.cuda.Event(), torch.cuda.Event())   # recorded once OUTSIDE the capture: that makes the handles)
                for e in self.events:
                    e.record()
            rec = _lib.CommandList(self.side.cuda_stream)
            _REC_EVENTS = tuple(e.cuda_event for e in self.events)
        try:
            # thread-local error mode: other threads (RCCL's watchdog polls events) must not invalidate the recording
            if rec is None:
                with torch.cuda.graph(g, pool=pool, stream=self.side, capture_error_mode="thread_local"):
                    out = fn()
            else:
                # the same capture WITHOUT torch.cuda.graph's gc.collect() + empty_cache(): emptying the caching allocator here hands
                # the eager pool's 20 GB back to the driver, and the next eager pass (a bracketed bench step, a second forward while
                # the recorded activations are busy) hipMallocs them again -- 95 segments in the middle of a training loop; on a box
                # with slow allocations that was 280 instead of 355 scans/s for the whole first process
                cur = torch.cuda.current_stream()
                self.side.wait_stream(cur)
                with torch.cuda.stream(self.side):
                    g.capture_begin(pool=pool, capture_error_mode="thread_local")
                    _lib.RECORDER = rec
                    try:
                        out = fn()
                    finally:
                        _lib.RECORDER = None
                        g.capture_end()
                cur.wait_stream(self.side)
        finally:
            _CAPTURE = None
        if rec is not None:
            rec.finish()
            GRAPH_STATS["native_lists"] = GRAPH_STATS.get("native_lists", 0) + 1
        return (g, rec), out

    def _replay(self, rec_pair):
        g, rec = rec_pair
        if rec is None:
            g.replay()
            return
        ws = wgrad_stream(self.dev)
        rec.replay(stream(), None if ws is None else ws.cuda_stream)

    def record_forward(self, P, flat, training, drop_p):
        self.side = torch.cuda.Stream(device=self.dev)   # the recording stream (replays run on the caller's)
        self.img = torch.empty(self.B, 3, self.H, self.W, dtype=torch.float32, device=self.dev)
        self.seed_t = torch.zeros(self.groups, dtype=torch.int64, device=self.dev)   # one dropout seed per image group
        self.training, self.drop_p = training, drop_p
        self.fwd, (self.feat, self.tape, self.J) = self._record(
            lambda: _backbone_forward(P, self.img, training, drop_p, 0, self.seed_t, self.dev, self.groups), None)
        self.dfeat = new_img(self.feat.B, self.feat.H, self.feat.W, 64, self.dev)
        self.ptrs = tuple(t.data_ptr() for t in flat)
        GRAPH_STATS["recorded"] += 1

    def params_moved(self, flat):
        return len(flat) != len(self.ptrs) or any(t.data_ptr() != a for t, a in zip(flat, self.ptrs))

    # -- replays
    def forward(self, imgc, seed):
        self.img.copy_(imgc)
        if self.training and self.drop_p > 0.0:
            if self.groups == 1:
                self.seed_t.fill_(seed)
            else:
                for g, sd in enumerate(seed):   # (fill kernels: a host -> device copy here would wait for the stream)
                    self.seed_t[g:g + 1].fill_(sd)
        _refresh_stale_forms(stream())   # every cached weight form of this stream that an update made stale: one launch
        self._replay(self.fwd)
        self.generation += 1
        GRAPH_STATS["forward_replays"] += 1

    def grads_attached(self, P, order):
        sink = GradSink(P, order)
        ptrs = []
        for n in order:
            p = P[n]
            if not p.requires_grad:
                continue
            if not sink._attached(p):
                return None
            ptrs.append(p.grad.data_ptr())
        return tuple(ptrs)

    def backward(self, P, order):
        """Replay (recording first if need be) the backward pass from self.dfeat.  False: cannot (gradients not attached, or moved
        since the recording) -- the caller walks self.tape eagerly."""
        gp = self.grads_attached(P, order)
        if gp is None or self.failed:
            return False
        if self.bwd is not None and gp != self.grad_ptrs:
            self.bwd = None      # the optimizer re-attached its gradient buffers elsewhere: record again
        if self.bwd is None:
            try:   # (recorded with a hook-free sink: FlatAdam's bucket hooks must not issue collectives inside the stream capture --
                   #  the outer pass's sink reports the gradients once the replay is enqueued: ADVICE r4)
                self.bwd, _ = self._record(
                    lambda: _backbone_backward(P, GradSink(P, order, defer_hooks=True), self.tape, self.J, self.feat, self.dfeat, self.training, 0,
                                               self.seed_t, False, self.H, self.W, self.groups), self.fwd[0].pool())
            except RuntimeError:   # (_CaptureMiss is one) -- this key stays eager from now on
                self.failed = True
                return False
            self.grad_ptrs = gp
        self._replay(self.bwd)
        GRAPH_STATS["backward_replays"] += 1
        return True


def _graph_for(spec, imgc, training, drop_p, flat, want_dimg):
    """-> (the Graph2D record of this pass's key or None, replay it?)."""
    holder = getattr(spec, "graphs", None)
    if (not (GRAPH_2D or NATIVE_2D) or holder is None or not training or not getattr(spec, "grad_enabled", False) or syncbn.active()
            or DEBUG is not None or want_dimg):
        return None, False
    graphs = holder.__dict__.setdefault("graphs2d", {})
    B, _, H, W = imgc.shape
    groups = getattr(spec, "groups", 1)
    key = (B, H, W, groups, float(drop_p), spec.num_classes, stream(), F4_ROLES, WGRAD_STREAM, NATIVE_2D and not GRAPH_2D)
    g = graphs.get(key)
    if g is None:
        if len(graphs) >= GRAPH_2D_MAX_KEYS:
            return None, False
        g = graphs[key] = Graph2D(B, H, W, imgc.device, groups)
    if g.failed:
        return None, False
    if g.fwd is not None and g.params_moved(flat):   # .data was re-pointed under the same parameter objects
        del graphs[key]
        GRAPH_STATS["dropped"] += 1
        return None, False
    g.calls += 1
    if not g.eager_backward_seen:   # the first forward + backward of a key run eagerly
        return g, False
    if g.busy():      # the activations of the previous forward replay still wait for their backward pass
        GRAPH_STATS["eager_busy"] += 1
        return g, False
    return g, True


# ------------------------------------------------------------------------------------------------ the network
class Net2DFunction(torch.autograd.Function):
    """img (B,3,H,W) -> feats (N,64), seg_logit, seg_logit2, seg_logit_all (B,H,W,C) as one autograd node."""

    @staticmethod
    def forward(ctx, spec, img, point_pix, training, drop_p, drop_seed, *flat):
        ctx.set_materialize_grads(False)   # an output that no loss uses arrives as None in backward, not as a zero tensor
        dev = img.device
        P = dict(zip(spec.order, flat))
        B, _, H, W = img.shape
        Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
        imgc = img.contiguous().float()
        ctx.graph_key, replay = _graph_for(spec, imgc, training, drop_p, flat, ctx.needs_input_grad[1])
        graph = ctx.graph_key if replay else None
        if graph is not None and graph.fwd is None:
            try:
                graph.record_forward(P, flat, training, drop_p)
            except RuntimeError:   # _CaptureMiss, an entry point the recorder refuses (host pointer of unknown size), or
                # torch.OutOfMemoryError (a RuntimeError): the recording pins a second set of this key's activations in a private
                # pool -- when that does not fit, the key stays eager from now on and the half-built pool is released with the record
                graph.failed, graph = True, None
                ctx.graph_key.fwd = ctx.graph_key.bwd = None
        if graph is not None:
            graph.forward(imgc, drop_seed)
            feat, tape, J = graph.feat, graph.tape, graph.J
            ctx.token = _Token()
            import weakref
            graph.pending = weakref.ref(ctx.token)
            ctx.generation = graph.generation
        else:
            feat, tape, J = _backbone_forward(P, imgc, training, drop_p, drop_seed, None, dev, getattr(spec, "groups", 1))
        ctx.graph = graph
        # ---- heads (xmuda_arch.py:58-77)
        C = spec.num_classes
        pred_all = torch.empty(B, H, W, C, dtype=torch.float32, device=dev)
        call("mopa_pixel_head_fwd", feat.p, feat.ld, B, Hp, Wp, H, W, 64, C, ptr(P["linear.weight"]),
             ptr(P["linear.bias"]), ptr(pred_all), stream())
        N = point_pix.numel()
        feats = torch.empty(N, 64, dtype=torch.float32, device=dev)
        l1 = torch.empty(N, C, dtype=torch.float32, device=dev)
        l2 = torch.empty(N, C if spec.dual_head else 0, dtype=torch.float32, device=dev)
        if N > 0:
            call("mopa_output_layer_heads_fwd", feat.p, feat.ld, ptr(point_pix), N, 64, C, ptr(P["linear.weight"]),
                 ptr(P["linear.bias"]), ptr(P["linear2.weight"]) if spec.dual_head else None,
                 ptr(P["linear2.bias"]) if spec.dual_head else None, ptr(feats), ptr(l1),
                 ptr(l2) if spec.dual_head else None, stream())
        ctx.spec, ctx.P, ctx.tape, ctx.J, ctx.training = spec, P, tape, J, training
        if DEBUG is not None:
            DEBUG.update({f"J{k}": v.clone() for k, v in J.items()})
        # feats.detach(): an alias without grad_fn -- the output object itself on ctx would be a reference cycle through this
        # node, freed only by the cyclic GC (the activations of every step stayed allocated until then)
        ctx.feat, ctx.feats, ctx.point_pix, ctx.dims = feat, feats.detach(), point_pix, (B, H, W, Hp, Wp, N)
        ctx.drop_seed, ctx.img_dtype = drop_seed, img.dtype
        return feats, l1, l2, pred_all

    @staticmethod
    def backward(ctx, dfeats, dl1, dl2, dpred):
        if dfeats is None and dl1 is None and dl2 is None and dpred is None:   # nothing flows back into this network
            return (None,) * (6 + len(ctx.spec.order))
        spec, P, tape, J = ctx.spec, ctx.P, ctx.tape, ctx.J
        B, H, W, Hp, Wp, N = ctx.dims
        feat = ctx.feat
        dev = feat.t.device
        C = spec.num_classes
        graph = ctx.graph
        if graph is not None and ctx.generation != graph.generation:
            raise RuntimeError("Net2DSeg backward: the activations of this forward pass were overwritten by a later graph replay "
                               "(a second backward through the same pass after another forward?); set MOPA_NATIVE_2D=0 (and leave "
                               "MOPA_GRAPH_2D unset) to run every pass eagerly")
        sink = GradSink(P, spec.order)   # gradients go straight into attached .grad buffers (accumulating)

        def cont(t):
            return None if t is None else t.contiguous().float()

        dfeats, dl1, dpred = cont(dfeats), cont(dl1), cont(dpred)
        dl2 = cont(dl2) if (spec.dual_head and dl2 is not None and dl2.numel()) else None
        # ---- heads: d(feat) = point-head part (dense over all pixels, zeros where no point) + full-image part
        dfeat = graph.dfeat if graph is not None else new_img(B, Hp, Wp, 64, dev)
        head_w_acc = False
        if N > 0 and (dfeats is not None or dl1 is not None or dl2 is not None):
            rows = B * Hp * Wp
            row_start = torch.empty(rows + 1, dtype=torch.int32, device=dev)
            row_points = torch.empty(N, dtype=torch.int32, device=dev)
            ws = _ws(query("mopa_points_csr_workspace_bytes", rows), dev)
            call("mopa_points_csr", ptr(ctx.point_pix), N, rows, ptr(row_start), ptr(row_points), ptr(ws), ws.numel(),
                 stream())
            ws = _ws(query("mopa_output_layer_heads_bwd_workspace_bytes", N, 64, C), dev)
            hnames = (["linear.weight", "linear.bias"] if dl1 is not None else []) + \
                     (["linear2.weight", "linear2.bias"] if dl2 is not None else [])
            hg, hacc = sink.take(*hnames)
            hg = dict(zip(hnames, hg))
            call("mopa_output_layer_heads_bwd", ptr(dfeats), ptr(dl1), ptr(dl2), ptr(ctx.feats), ptr(P["linear.weight"]),
                 ptr(P["linear2.weight"]) if spec.dual_head else None, ptr(row_start), ptr(row_points), rows, N, 64, C,
                 dfeat.p, dfeat.ld, ptr(hg.get("linear.weight")), ptr(hg.get("linear.bias")), ptr(hg.get("linear2.weight")),
                 ptr(hg.get("linear2.bias")), int(hacc), ptr(ws), ws.numel(), stream())
            head_w_acc = dl1 is not None
        else:
            dfeat.t.zero_()
        if dpred is not None:
            (pw, pb), pacc = sink.take("linear.weight", "linear.bias")   # same tensors as the point head's, if it ran
            ws = _ws(query("mopa_pixel_head_bwd_workspace_bytes", B, H, W, 64, C), dev)
            call("mopa_pixel_head_bwd", ptr(dpred), feat.p, feat.ld, B, Hp, Wp, H, W, 64, C, ptr(P["linear.weight"]),
                 dfeat.p, dfeat.ld, 1, ptr(pw), ptr(pb), int(head_w_acc or pacc), ptr(ws), ws.numel(), stream())
        dimg = None
        if graph is not None and graph.backward(P, spec.order):
            pass
        else:
            if graph is not None:
                GRAPH_STATS["eager_backward"] += 1
            elif ctx.graph_key is not None:
                ctx.graph_key.eager_backward_seen = True
            dimg = _backbone_backward(P, sink, tape, J, feat, dfeat, ctx.training, ctx.drop_seed,
                                      graph.seed_t if graph is not None else None, ctx.needs_input_grad[1], H, W,
                                      getattr(spec, "groups", 1))
        if graph is not None:
            graph.pending = None   # the activations are free for the next forward replay
        return (None, dimg if dimg is None else dimg.to(ctx.img_dtype), None, None, None, None) + sink.returned()
