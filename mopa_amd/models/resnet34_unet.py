"""UNetResNet34: the 2D image backbone, MI355X-native.

Drop-in for ``mopa/models/resnet34_unet.py:83-191``: same constructor (``pretrained``), same parameter / buffer
names as the reference's ``state_dict`` (torchvision ResNet34 encoder names + ``dec_*`` decoder names), same
``dropout`` attribute (``nn.Dropout(p=0.4)`` whose ``p`` the caller may change).  The torch modules below are
parameter HOLDERS only -- they are never called; the arithmetic runs in libmopa_hip.so through
``mopa_amd.dense2d.Net2DFunction`` (no CPU fallback).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn


class _BasicBlockParams(nn.Module):
    def __init__(self, cin, c, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, c, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(c)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c)
        if stride != 1 or cin != c:
            self.downsample = nn.Sequential(nn.Conv2d(cin, c, 1, stride, bias=False), nn.BatchNorm2d(c))


def _stage(cin, c, n, stride):
    return nn.Sequential(_BasicBlockParams(cin, c, stride), *[_BasicBlockParams(c, c, 1) for _ in range(n - 1)])


def _dec_stage(c_in_enc, c_out_enc, num_concat):
    conv = nn.Sequential(nn.Conv2d(num_concat * c_out_enc, c_out_enc, 3, padding=1), nn.BatchNorm2d(c_out_enc),
                         nn.ReLU(inplace=True))
    t_conv = nn.Sequential(nn.ConvTranspose2d(c_out_enc, c_in_enc, 2, 2), nn.BatchNorm2d(c_in_enc), nn.ReLU(inplace=True))
    return conv, t_conv


class UNetResNet34(nn.Module):
    def __init__(self, pretrained=True):
        super().__init__()
        # encoder: torchvision resnet34 layout with a stride-1 conv1 (resnet34_unet.py:93-94)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=1, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = _stage(64, 64, 3, 1)
        self.layer2 = _stage(64, 128, 4, 2)
        self.layer3 = _stage(128, 256, 6, 2)
        self.layer4 = _stage(256, 512, 3, 2)
        # decoder (resnet34_unet.py:104-110)
        _, self.dec_t_conv_stage5 = _dec_stage(256, 512, 1)
        self.dec_conv_stage4, self.dec_t_conv_stage4 = _dec_stage(128, 256, 2)
        self.dec_conv_stage3, self.dec_t_conv_stage3 = _dec_stage(64, 128, 2)
        self.dec_conv_stage2, self.dec_t_conv_stage2 = _dec_stage(64, 64, 2)
        self.dec_conv_stage1 = nn.Conv2d(2 * 64, 64, kernel_size=3, padding=1)
        self.dropout = nn.Dropout(p=0.4)
        # torchvision's ResNet init for the encoder: kaiming_normal(fan_out) convs, BN weight 1 / bias 0
        for name, m in self.named_modules():
            if isinstance(m, nn.Conv2d) and not name.startswith("dec_"):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if pretrained:
            self._load_imagenet()

    def _load_imagenet(self):
        """ImageNet weights for the encoder (resnet34_unet.py:90-94 uses torchvision's download; offline here)."""
        path = os.environ.get("MOPA_RESNET34_WEIGHTS")
        if path:
            sd = torch.load(path, map_location="cpu")
        else:
            try:
                from torchvision.models.resnet import resnet34  # type: ignore
                sd = resnet34(pretrained=True).state_dict()
            except Exception as e:  # no torchvision / no network
                raise RuntimeError(
                    "UNetResNet34(pretrained=True) needs torchvision's ImageNet ResNet34 weights; set "
                    "MOPA_RESNET34_WEIGHTS=/path/to/resnet34.pth or build with pretrained=False") from e
        own = self.state_dict()
        self.load_state_dict({k: v for k, v in sd.items() if k in own and v.shape == own[k].shape}, strict=False)
