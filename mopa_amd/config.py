"""Minimal attribute-style config with the reference's defaults for the model sub-trees
(``mopa/config/xmuda.py:188-224``); yacs is not required."""


class Node(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def default_cfg(num_classes=5, dual_head=True, pretrained=False):
    return Node(
        MODEL_2D=Node(TYPE="UNetResNet34", NUM_CLASSES=num_classes, DUAL_HEAD=dual_head,
                      UNetResNet34=Node(pretrained=pretrained)),
        MODEL_3D=Node(TYPE="SCN", NUM_CLASSES=num_classes, DUAL_HEAD=dual_head,
                      SCN=Node(in_channels=1, m=16, block_reps=1, residual_blocks=False, full_scale=4096,
                               num_planes=7, pretrained=False)),
    )
