"""``ComposerModel`` / ``Metric`` stand-ins.

The reference derives its model from ``composer.models.ComposerModel`` (diffusion/models/stable_diffusion.py:10,15)
and uses ``torchmetrics.MeanSquaredError`` (:11,100-103).  Neither package exists in this image; when they are
importable the real classes are used, otherwise these minimal equivalents with the same protocol."""
from __future__ import annotations

import torch
import torch.nn as nn

try:  # pragma: no cover - not installed in the build image
    from composer.models import ComposerModel  # type: ignore
except Exception:  # noqa: BLE001

    class ComposerModel(nn.Module):
        """Protocol: forward(batch) -> outputs ; loss(outputs, batch) ; eval_forward ; get_metrics ; update_metric."""

        def forward(self, batch):
            raise NotImplementedError

        def loss(self, outputs, batch):
            raise NotImplementedError

        def eval_forward(self, batch, outputs=None):
            return outputs if outputs is not None else self.forward(batch)

        def get_metrics(self, is_train: bool = False):
            return {}

        def update_metric(self, batch, outputs, metric):
            pass


try:  # pragma: no cover
    from torchmetrics import MeanSquaredError, Metric  # type: ignore
except Exception:  # noqa: BLE001

    class Metric(nn.Module):

        def update(self, *a, **kw):
            raise NotImplementedError

        def compute(self):
            raise NotImplementedError

        def reset(self):
            pass

    class MeanSquaredError(Metric):
        """sum of squared error / count, accumulated on the device of the inputs (torchmetrics semantics)."""

        def __init__(self, **kw):
            super().__init__()
            self._kw = kw
            self.reset()

        def reset(self):
            self.sum_squared_error = None
            self.total = 0

        def update(self, preds, target):
            d = (preds.float() - target.float())
            sse = (d * d).sum()
            self.sum_squared_error = sse if self.sum_squared_error is None else self.sum_squared_error + sse
            self.total += target.numel()

        def compute(self):
            if self.sum_squared_error is None:
                return torch.tensor(float('nan'))
            return self.sum_squared_error / self.total
