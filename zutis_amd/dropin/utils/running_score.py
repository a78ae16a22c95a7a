"""Drop-in replacement for the reference's utils/running_score.py (RunningScore) with the confusion-matrix
histogram on the GPU (zh_confusion_hist).  Same API and return values (utils/running_score.py:5-49).  Labels may be
NumPy arrays (as trainer.py:347 passes them) or torch tensors already on the GPU (no D2H/H2D round trip)."""
import numpy as np
import torch

from zutis_amd import ops as _ops


class RunningScore(object):
    def __init__(self, n_classes, device: torch.device = torch.device("cuda:0")):
        self.n_classes = n_classes
        self.device = device
        self._hist = torch.zeros((n_classes * n_classes,), dtype=torch.int64, device=device)

    @property
    def confusion_matrix(self):
        return self._hist.cpu().numpy().reshape(self.n_classes, self.n_classes).astype(np.float64)

    def _dev(self, a):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=self.device, dtype=torch.int64).contiguous()

    def update(self, label_trues, label_preds):
        for lt, lp in zip(label_trues, label_preds):
            _ops.confusion_hist(self._dev(lt).reshape(-1), self._dev(lp).reshape(-1), self._hist, self.n_classes)

    def get_scores(self):
        hist = self.confusion_matrix
        with np.errstate(divide="ignore", invalid="ignore"):
            acc = np.diag(hist).sum() / hist.sum()
            acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
            iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
            mean_iu = np.nanmean(iu)
            freq = hist.sum(axis=1) / hist.sum()
            fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
        cls_iu = dict(zip(range(self.n_classes), iu))
        return {"Pixel Acc": acc, "Mean Acc": acc_cls, "FreqW Acc": fwavacc, "Mean IoU": mean_iu}, cls_iu

    def reset(self):
        self._hist.zero_()
