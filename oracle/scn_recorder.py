"""A *recording* stand-in for the ``sparseconvnet`` package (TEST INFRASTRUCTURE ONLY).

``sparseconvnet`` is un-vendored and absent here, so the reference's
``mopa/models/scn_unet.py`` cannot execute.  This module gives it just enough of the
package's *container protocol* -- ``Sequential().add(...)``, ``ConcatTable``, ``JoinTable``,
``AddTable``, ``Identity`` and leaf layers that hold SCN-shaped parameters -- to let the
reference's own constructors and ``forward`` run on a symbolic tensor, while every leaf
logs what it was built with and what flows through it.  ``oracle/gen_golden.py::gen_g6``
imports the reference under it and commits the log as ``tests/golden/g6_scn_structure.json``.

What the log pins, and what it cannot:
* pinned by the reference's own code: the constructor arguments of every layer
  ``UNetSCN`` (scn_unet.py:25-30) and ``UNetSCN_ED`` (scn_unet.py:49-94) build, ED's
  execution order, channel counts and JoinTable operand order (scn_unet.py:96-135).
* ``scn.UNet`` itself lives inside the third-party package.  ``UNet`` below restates the
  published facebookresearch/SparseConvNet ``sparseconvnet/networkArchitectures.py``
  function (VGG and ResNet style blocks) over these generic containers; gen_g6 checks
  that its execution trace equals the reference's unrolled ``UNetSCN_ED`` layer for
  layer.  Parameter *names* under ``scn.UNet`` follow from that restatement plus
  nn.Module's index naming -- consistent with the reference's wiring, but not verified
  against a released checkpoint.
* arithmetic: nothing (no numbers flow; see tests/test_oracle_scn3d.py for that).
"""
from __future__ import annotations

import types

import torch
import torch.nn as nn

LOG: list = []      # constructor log: one dict per leaf layer, in construction order
TRACE: list = []    # execution log: one dict per leaf / table op, in forward order


class Sym:
    """Symbolic sparse tensor: channel count, UNet level (0 = finest) and the op that produced it."""
    _n = 0

    def __init__(self, channels, level, src):
        Sym._n += 1
        self.id, self.C, self.level, self.src = Sym._n, channels, level, src


def _qual(root: nn.Module, target: nn.Module) -> str:
    for name, m in root.named_modules():
        if m is target:
            return name
    return "?"


_ROOT = [None]


def set_root(module):
    """Names in the trace are relative to this module (call before running forward)."""
    _ROOT[0] = module
    TRACE.clear()


class _Leaf(nn.Module):
    kind = "leaf"

    def _log(self, **kw):
        self.ctor = dict(type=type(self).__name__, **kw)
        LOG.append(self.ctor)

    def _emit(self, x, cout, level):
        out = Sym(cout, level, type(self).__name__)
        TRACE.append(dict(op=type(self).__name__, name=_qual(_ROOT[0], self), cin=x.C if isinstance(x, Sym) else None,
                          cout=cout, level_in=x.level if isinstance(x, Sym) else None, level_out=level,
                          src=x.id if isinstance(x, Sym) else None, dst=out.id))
        return out


class InputLayer(_Leaf):
    def __init__(self, dimension, spatial_size, mode=3):
        super().__init__()
        self._log(dimension=dimension, spatial_size=int(spatial_size), mode=mode)

    def forward(self, x):   # x = [coords, features]; features carry the channel count
        out = Sym(x[1].shape[1], 0, "InputLayer")
        TRACE.append(dict(op="InputLayer", name=_qual(_ROOT[0], self), cin=out.C, cout=out.C, level_in=None, level_out=0,
                          src=None, dst=out.id))
        return out


class OutputLayer(_Leaf):
    def __init__(self, dimension):
        super().__init__()
        self._log(dimension=dimension)

    def forward(self, x):
        return self._emit(x, x.C, x.level)


class _Conv(_Leaf):
    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, volume, dlevel):
        super().__init__()
        self.nIn, self.nOut, self.dlevel = nIn, nOut, dlevel
        # SparseConvNet stores (filter_volume, nIn, nOut); releases with grouped convolutions (filter_volume, 1, nIn, nOut)
        self.weight = nn.Parameter(torch.zeros(volume, nIn, nOut))
        if bias:
            self.bias = nn.Parameter(torch.zeros(nOut))
        self._log(dimension=dimension, nIn=nIn, nOut=nOut, filter_size=filter_size, filter_stride=filter_stride,
                  bias=bool(bias), filter_volume=volume)

    def forward(self, x):
        assert x.C == self.nIn, (type(self).__name__, x.C, self.nIn)
        return self._emit(x, self.nOut, x.level + self.dlevel)


class SubmanifoldConvolution(_Conv):
    def __init__(self, dimension, nIn, nOut, filter_size, bias, groups=1):
        super().__init__(dimension, nIn, nOut, filter_size, 1, bias, filter_size ** dimension, 0)


class Convolution(_Conv):
    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, groups=1):
        super().__init__(dimension, nIn, nOut, filter_size, filter_stride, bias, filter_size ** dimension, +1)


class Deconvolution(_Conv):
    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, groups=1):
        super().__init__(dimension, nIn, nOut, filter_size, filter_stride, bias, filter_size ** dimension, -1)


class NetworkInNetwork(_Leaf):
    def __init__(self, nIn, nOut, bias):
        super().__init__()
        self.nIn, self.nOut = nIn, nOut
        self.weight = nn.Parameter(torch.zeros(nIn, nOut))
        if bias:
            self.bias = nn.Parameter(torch.zeros(nOut))
        self._log(nIn=nIn, nOut=nOut, bias=bool(bias))

    def forward(self, x):
        assert x.C == self.nIn
        return self._emit(x, self.nOut, x.level)


class BatchNormalization(_Leaf):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, affine=True, leakiness=1):
        super().__init__()
        self.nPlanes = nPlanes
        if affine:
            self.weight = nn.Parameter(torch.ones(nPlanes))
            self.bias = nn.Parameter(torch.zeros(nPlanes))
        self.register_buffer("running_mean", torch.zeros(nPlanes))
        self.register_buffer("running_var", torch.ones(nPlanes))
        self._log(nPlanes=nPlanes, eps=eps, momentum=momentum, leakiness=leakiness)

    def forward(self, x):
        assert x.C == self.nPlanes, (x.C, self.nPlanes)
        return self._emit(x, x.C, x.level)


class BatchNormReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9):
        super().__init__(nPlanes, eps, momentum, True, 0)


class BatchNormLeakyReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, leakiness=0.333):
        super().__init__(nPlanes, eps, momentum, True, leakiness)


class Identity(nn.Module):
    def forward(self, x):
        return x


class Sequential(nn.Sequential):
    def add(self, module):
        self.add_module(str(len(self._modules)), module)
        return self

    def forward(self, x):
        for m in self._modules.values():
            x = m(x)
        return x


class ConcatTable(nn.Sequential):
    def add(self, module):
        self.add_module(str(len(self._modules)), module)
        return self

    def forward(self, x):
        return [m(x) for m in self._modules.values()]


class JoinTable(nn.Module):
    def forward(self, xs):
        assert len({t.level for t in xs}) == 1
        out = Sym(sum(t.C for t in xs), xs[0].level, "JoinTable")
        TRACE.append(dict(op="JoinTable", name=_qual(_ROOT[0], self), parts=[t.C for t in xs], srcs=[t.id for t in xs],
                          part_ops=[t.src for t in xs], cout=out.C, level_out=out.level, dst=out.id))
        return out


class AddTable(nn.Module):
    def forward(self, xs):
        assert len({(t.level, t.C) for t in xs}) == 1
        out = Sym(xs[0].C, xs[0].level, "AddTable")
        TRACE.append(dict(op="AddTable", name=_qual(_ROOT[0], self), srcs=[t.id for t in xs], part_ops=[t.src for t in xs],
                          cout=out.C, level_out=out.level, dst=out.id))
        return out


def UNet(dimension, reps, nPlanes, residual_blocks=False, downsample=[2, 2], leakiness=0, n_input_planes=-1):
    """Restatement of the published ``sparseconvnet.networkArchitectures.UNet`` (the one piece of this file that is
    not driven by the reference's own source).  Logged as a call so the fixture shows what scn_unet.py:28 passes."""
    LOG.append(dict(type="UNet", dimension=dimension, reps=reps, nPlanes=list(nPlanes), residual_blocks=bool(residual_blocks),
                    downsample=list(downsample), leakiness=leakiness))

    def block(m, a, b):
        if residual_blocks:   # ResNet style
            m.add(ConcatTable()
                  .add(Identity() if a == b else NetworkInNetwork(a, b, False))
                  .add(Sequential()
                       .add(BatchNormLeakyReLU(a, leakiness=leakiness))
                       .add(SubmanifoldConvolution(dimension, a, b, 3, False))
                       .add(BatchNormLeakyReLU(b, leakiness=leakiness))
                       .add(SubmanifoldConvolution(dimension, b, b, 3, False)))
                  ).add(AddTable())
        else:                 # VGG style
            m.add(Sequential()
                  .add(BatchNormLeakyReLU(a, leakiness=leakiness))
                  .add(SubmanifoldConvolution(dimension, a, b, 3, False)))

    def U(nPlanes, n_input_planes=-1):
        m = Sequential()
        for i in range(reps):
            block(m, n_input_planes if n_input_planes != -1 else nPlanes[0], nPlanes[0])
            n_input_planes = -1
        if len(nPlanes) > 1:
            m.add(ConcatTable()
                  .add(Identity())
                  .add(Sequential()
                       .add(BatchNormLeakyReLU(nPlanes[0], leakiness=leakiness))
                       .add(Convolution(dimension, nPlanes[0], nPlanes[1], downsample[0], downsample[1], False))
                       .add(U(nPlanes[1:]))
                       .add(BatchNormLeakyReLU(nPlanes[1], leakiness=leakiness))
                       .add(Deconvolution(dimension, nPlanes[1], nPlanes[0], downsample[0], downsample[1], False))))
            m.add(JoinTable())
            for i in range(reps):
                block(m, nPlanes[0] * (2 if i == 0 else 1), nPlanes[0])
        return m

    return U(nPlanes, n_input_planes)


def as_module() -> types.ModuleType:
    mod = types.ModuleType("sparseconvnet")
    for k, v in globals().items():
        if isinstance(v, type) and issubclass(v, nn.Module) or k == "UNet":
            setattr(mod, k, v)
    return mod
