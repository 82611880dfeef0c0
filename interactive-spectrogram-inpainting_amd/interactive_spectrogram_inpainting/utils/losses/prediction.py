"""`LabelSmoothingLoss` of the reference (utils/losses/prediction.py:5-20), the
criterion of the prior (train_autoregressive_model.py:666-668): class scores along
`dim`, smoothed one-hot targets, mean over all positions.  On the GPU the loss and
its gradient come from one fused kernel (isi_label_smoothing_loss_f32)."""
from __future__ import annotations

import torch
from torch import nn

from ...priors._train import label_smoothing_loss


class LabelSmoothingLoss(nn.Module):
    def __init__(self, num_classes: int, smoothing: float = 0.0, dim: int = -1):
        super().__init__()
        self.smoothing = smoothing
        self.confidence = 1.0 - smoothing
        self.num_classes = num_classes
        self.dim = dim

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        # the reference scatters the confidence along dim 1 whatever `dim` is; the two agree for
        # dim=1 (how the training script builds it) and for 2-D [rows, classes] inputs with dim=-1
        if self.dim % pred.dim() != 1:
            raise NotImplementedError("LabelSmoothingLoss: class scores must sit on dim 1 (as in the reference's use)")
        return label_smoothing_loss(pred, target, self.num_classes, self.smoothing, dim=1)

    def __repr__(self):
        return f"LabelSmoothingLoss(num_classes={self.num_classes}, smoothing={self.smoothing}, dim={self.dim})"
