"""Class-weighted soft-target cross entropy (reference: utils/custom_loss.py:7-18).

Only used by the reference's mix-up fine-tuning scripts, never by the pre-training hot path
(the masked reconstruction loss lives in model/vit_autoenc.py:205-232).  Kept as a small
plain-torch module so ``from utils.custom_loss import SoftCrossEntropyWithWeightsLoss`` resolves.
"""
import torch
from torch import nn


class SoftCrossEntropyWithWeightsLoss(nn.Module):
    def __init__(self, weights):
        super().__init__()
        self.weights = nn.Parameter(torch.as_tensor(weights), requires_grad=False)

    def forward(self, y_hat, y):
        # per class c: sum_n -y[n,c] * w[c] * log_softmax(y_hat)[n,c] / sum(w); then mean over classes
        logp = torch.log_softmax(y_hat, dim=-1)
        per_class = (-(y * logp) * self.weights).sum(dim=0) / self.weights.sum()
        return per_class.mean()

    def __repr__(self):
        return f"weights are on {self.weights.device}\n"
