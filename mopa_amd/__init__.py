"""mopa_amd: MI355X-native (gfx950) hot path of MoPA behind the reference's model-factory API."""
__version__ = "0.2.0"


def invalidate_weight_caches():
    """Forget every cached re-layout of a conv weight (Winograd / implicit-GEMM / packed sparse-conv forms).

    The caches are validated by autograd's version counter plus an epoch that this package's own raw updates bump
    (``FlatAdam.step``, ``FlatEMA``), and every forward under ``torch.no_grad()`` bumps it on entry and exit (the
    ``torch_ema`` teacher pass of ``mopa/train/train_xmuda_mopa.py:264-280``).  Call this after any OTHER write that goes
    through ``param.data`` while gradients are enabled -- ``dist.broadcast(p.data, 0)`` after a first forward, a manual
    ``p.data.copy_(...)`` -- which changes neither the version counter nor the address."""
    from ._lib import WEIGHTS_EPOCH
    WEIGHTS_EPOCH[0] += 1
