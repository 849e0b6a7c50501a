"""MDiceLoss / MDiceLoss_Val of the reference's MicFormer/loss/dice.py on fused HIP reductions.

MDiceLoss.forward (dice.py:158-166): per class channel, sigmoid-Dice with squared denominator and smooth 1 (sums over
batch AND space) plus BCE on the sigmoid output, (0.7*sum dice + 0.3*sum bce)/K -- one streaming pass forward, one
element-wise pass backward.
"""
import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops


def as_target(logits_shape, target):
    """float one-hot planes stay as they are; an integer class map (one dim fewer, or a singleton channel) becomes uint8."""
    if not target.is_floating_point() and (target.dim() == len(logits_shape) - 1 or target.shape[1] == 1 != logits_shape[1]):
        return target.reshape((target.shape[0],) + tuple(logits_shape[2:])).to(torch.uint8)
    return target.float()


def _as_target(inputs, target):
    return as_target(tuple(inputs.shape), target)


class MDiceLoss(nn.Module):
    def __init__(self, do_sigmoid=True):
        super().__init__()
        if not do_sigmoid:
            raise NotImplementedError("HIP MDiceLoss implements do_sigmoid=True (reference default)")
        self.do_sigmoid = do_sigmoid
        self.labels = ['backgroud', 'CT-A', 'CT-B', 'CT-C', 'CT-D', 'CT-E', 'CT-F', 'CT-G']

    def forward(self, inputs, target):
        """target: one-hot float planes (B, K, D, H, W) as the reference feeds (train.py:177), or -- extension -- the integer
        class map (B, D, H, W) / (B, 1, D, H, W) they are expanded from (uint8 on the wire: 8x fewer bytes)."""
        return Fn.DiceBCEFn.apply(inputs.float(), _as_target(inputs, target))

    def metric(self, inputs, target):
        """Thresholded-sigmoid per-class Dice per sample (dice.py:168-175, binary_dice metric_mode): ONE device reduction
        (micf_dice_metric), returned in the reference's list-of-lists-of-0-d-tensors form (views of one [B, K] tensor, no host sync).
        The reference also prints "No <label> for this patient" for empty target planes; that message is not reproduced."""
        m = ops.dice_metric(inputs.float().contiguous(), _as_target(inputs, target).contiguous())
        return [[m[j, i] for i in range(m.shape[1])] for j in range(m.shape[0])]


class MDiceLoss_Val(MDiceLoss):
    """Validation loss = the Dice part only (dice.py:216-221); computed from the same fused sums."""

    def forward(self, inputs, target):
        inputs, target = inputs.float().contiguous(), _as_target(inputs, target).contiguous()
        _, sums = ops.dice_bce_fwd(inputs, target)
        s = sums.reshape(-1, 4)
        dice = 1.0 - (2.0 * s[:, 0] + 1.0) / (s[:, 1] + s[:, 2] + 1.0)
        return (dice.sum() / s.shape[0]).float()
