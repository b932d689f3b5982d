# -*- coding: utf-8 -*-
"""Evaluation metrics (reference evaluate/metrics.py:6-31)."""
import numpy as np
import torch

from .. import ops


class Metrics:

    def hit_rate(self, y_hat, y_pred):
        """Fraction of rows of `y_pred` (n, k) that contain at least one of the targets `y_hat` (t,)
        (evaluate/metrics.py:6-20).  Host-side: not on the training path."""
        targets = np.asarray(y_hat.numpy() if hasattr(y_hat, "numpy") else y_hat).reshape(-1)
        pred = np.asarray(y_pred.numpy() if hasattr(y_pred, "numpy") else y_pred)
        hits = np.isin(pred, targets).any(axis=1)
        return hits.sum() / pred.shape[0]

    def auc_score(self, positive, negative):
        """(positive > negative).sum() / len(positive)  (evaluate/metrics.py:23-31); counted on the GPU."""
        p = positive.detach().reshape(-1).contiguous().float()
        n = negative.detach().reshape(-1).contiguous().float()
        cnt = torch.zeros(1, dtype=torch.int32, device=p.device)
        ops.hinge_auc(p, n, None, cnt)
        return cnt[0] / len(positive)
