"""SegIoU, returned next to each model by the factory (mopa/models/metric.py:26-77).

Integer confusion-matrix accumulation: mat[label, pred] over label != ignore_index.
"""
import torch


class SegIoU(object):
    def __init__(self, num_classes, ignore_index=-100, name="seg_iou"):
        self.num_classes, self.ignore_index, self.name = num_classes, ignore_index, name
        self.mat = None

    def update_dict(self, preds, labels):
        logit, label = preds["seg_logit"], labels["seg_label"]
        n = self.num_classes
        with torch.no_grad():
            label = label.to(logit.device)
            keep = label != self.ignore_index
            inds = n * label[keep] + logit.argmax(1)[keep]
            add = torch.bincount(inds, minlength=n * n).reshape(n, n)
            self.mat = add if self.mat is None else self.mat + add

    def reset(self):
        self.mat = None

    @property
    def iou(self):
        h = self.mat.float()
        d = torch.diag(h)
        return d / (h.sum(1) + h.sum(0) - d)

    @property
    def global_avg(self):
        return self.iou.mean().item()

    @property
    def avg(self):
        return self.global_avg

    def __str__(self):
        return "{:.4f}".format(self.global_avg)

    @property
    def summary_str(self):
        return str(self)
