"""Model factory -- the operator API this package is a drop-in for.

Mirrors ``mopa/models/build.py:5-22``: ``build_model_2d(cfg)`` and
``build_model_3d(cfg)`` each return ``(nn.Module, SegIoU)``.  ``cfg`` may be a
yacs node or any object/dict with the same attribute layout
(``cfg.MODEL_3D.TYPE``, ``cfg.MODEL_3D[cfg.MODEL_3D.TYPE]`` ...).
"""
from .metric import SegIoU


def _sub(node, key):
    """cfg.MODEL_3D[TYPE] for yacs nodes, dicts and plain attribute objects."""
    try:
        return node[key]
    except (TypeError, KeyError):
        return getattr(node, key)


def _kwargs(node):
    return dict(node) if hasattr(node, "keys") else dict(vars(node))


def build_model_2d(cfg):
    from .xmuda_arch import Net2DSeg
    c = cfg.MODEL_2D
    model = Net2DSeg(num_classes=c.NUM_CLASSES, backbone_2d=c.TYPE, backbone_2d_kwargs=_kwargs(_sub(c, c.TYPE)),
                     dual_head=c.DUAL_HEAD, output_all=True)
    return model, SegIoU(c.NUM_CLASSES, name="iou_2d")


def build_model_3d(cfg):
    from .xmuda_arch import Net3DSeg
    c = cfg.MODEL_3D
    model = Net3DSeg(num_classes=c.NUM_CLASSES, backbone_3d=c.TYPE, backbone_3d_kwargs=_kwargs(_sub(c, c.TYPE)),
                     dual_head=c.DUAL_HEAD)
    return model, SegIoU(c.NUM_CLASSES, name="iou_3d")
