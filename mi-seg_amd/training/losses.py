"""Dice / Focal / CE losses with MONAI 1.1.0 semantics as used by the reference (networks/lightning_monai.py:46-67,
utils/training_utils.py:6-33).  Restated from MONAI's public API (MONAI is a pinned third-party dependency of the reference,
monai~=1.1.0, not present here): PARITY UNPINNED by any reference test -- SURVEY.md Appendix B.

On a HIP device ``DiceFocalLoss`` / ``DiceCELoss`` with ``to_onehot_y=True, softmax=True`` (the only way LitMonai builds them) are ONE
fused pass for the loss and one for d(loss)/d(logits) (csrc/training.hip) instead of ~6 torch passes over the
[B, C, 96^3] logits plus their autograd tape.  The torch arithmetic below is what runs for CPU tensors (the reference's ``--infer_cpu``
validation computes its loss on CPU logits, lightning_monai.py:187-189) and is the plain-PyTorch reference the GPU tests compare the
kernels with.

Channel handling, MONAI 1.1.0:
  * DiceFocalLoss one-hots the target, then -- when ``include_background`` is False -- strips channel 0 from BOTH logits and target
    before calling its sub-losses, which it builds WITHOUT include_background / to_onehot_y: the Dice softmax therefore runs over the
    C-1 foreground logits only, and the focal term sees the same C-1 channels.
  * DiceCELoss delegates include_background / to_onehot_y to its DiceLoss (softmax over all C channels, channel 0 dropped afterwards)
    and applies nn.CrossEntropyLoss over all channels."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from ..hip import lib as L
from ..hip import ops


def _one_hot(label, num_classes):
    # label [B,1,...] integer-valued -> [B,C,...]
    return F.one_hot(label[:, 0].long(), num_classes).movedim(-1, 1).to(torch.float32)


class DiceLoss(torch.nn.Module):
    def __init__(self, include_background=True, to_onehot_y=False, softmax=False, squared_pred=False, smooth_nr=1e-5, smooth_dr=1e-5):
        super().__init__()
        self.include_background, self.to_onehot_y, self.softmax = include_background, to_onehot_y, softmax
        self.squared_pred, self.smooth_nr, self.smooth_dr = squared_pred, float(smooth_nr), float(smooth_dr)

    def forward(self, logits, target):
        n_ch = logits.shape[1]
        p = torch.softmax(logits, 1) if self.softmax else logits
        t = _one_hot(target, n_ch) if self.to_onehot_y else target
        if not self.include_background and n_ch > 1:
            p, t = p[:, 1:], t[:, 1:]
        dims = tuple(range(2, logits.dim()))
        inter = (p * t).sum(dims)
        if self.squared_pred:
            den = (t * t).sum(dims) + (p * p).sum(dims)
        else:
            den = t.sum(dims) + p.sum(dims)
        return (1.0 - (2.0 * inter + self.smooth_nr) / (den + self.smooth_dr)).mean()


class FocalLoss(torch.nn.Module):
    """MONAI 1.1.0 FocalLoss: sigmoid/BCE form on RAW logits (not softmax), gamma 2, mean over space then (b, c)."""

    def __init__(self, include_background=True, to_onehot_y=False, gamma=2.0):
        super().__init__()
        self.include_background, self.to_onehot_y, self.gamma = include_background, to_onehot_y, gamma

    def forward(self, logits, target):
        n_ch = logits.shape[1]
        t = _one_hot(target, n_ch) if self.to_onehot_y else target
        x = logits
        if not self.include_background and n_ch > 1:
            x, t = x[:, 1:], t[:, 1:]
        b, c = x.shape[:2]
        x = x.reshape(b, c, -1).float()
        t = t.reshape(b, c, -1).float()
        max_val = (-x).clamp(min=0)
        ce = x - x * t + max_val + ((-max_val).exp() + (-x - max_val).exp()).log()
        loss = (F.logsigmoid(-x * (t * 2 - 1)) * self.gamma).exp() * ce
        return loss.mean(-1).mean()


class _FusedSegLoss(Function):
    """loss = fused kernel pass 1 (+ fixed-order finalize); d(loss)/d(logits) = pass 2 from the saved per-(b, c) sums."""

    @staticmethod
    def forward(ctx, logits, label, cfg):
        logits = logits.contiguous()
        loss, sums = ops.seg_loss_fwd(logits, label, cfg)
        ctx.save_for_backward(logits, label, sums)
        ctx.cfg = cfg
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, label, sums = ctx.saved_tensors
        return ops.seg_loss_bwd(logits, label, ctx.cfg, sums, g), None, None


def _fusable(logits, target, to_onehot_y, softmax):
    return (logits.is_cuda and logits.dtype == torch.float32 and to_onehot_y and softmax and target.shape[1] == 1 and 2 <= logits.shape[1] <= 16
            and logits.shape[0] <= 64)


class DiceFocalLoss(torch.nn.Module):
    def __init__(self, include_background=True, to_onehot_y=False, softmax=False, squared_pred=False, smooth_nr=1e-5, smooth_dr=1e-5, gamma=2.0,
                 lambda_dice=1.0, lambda_focal=1.0):
        super().__init__()
        # MONAI 1.1.0 builds the sub-losses without include_background / to_onehot_y; forward() strips channel 0 itself
        self.dice = DiceLoss(True, False, softmax, squared_pred, smooth_nr, smooth_dr)
        self.focal = FocalLoss(True, False, gamma)
        self.include_background, self.to_onehot_y, self.softmax = include_background, to_onehot_y, softmax
        self.lambda_dice, self.lambda_focal = lambda_dice, lambda_focal
        self.cfg = ops.SegLossCfg(L.LOSS_DICE_FOCAL, include_background, squared_pred, smooth_nr, smooth_dr, gamma, lambda_dice, lambda_focal)

    def forward(self, logits, target):
        if _fusable(logits, target, self.to_onehot_y, self.softmax):
            return _FusedSegLoss.apply(logits, target, self.cfg)
        return self.forward_torch(logits, target)

    def forward_torch(self, logits, target):
        n_ch = logits.shape[1]
        t = _one_hot(target, n_ch) if (self.to_onehot_y and n_ch > 1) else target
        x = logits
        if not self.include_background and n_ch > 1:
            x, t = x[:, 1:], t[:, 1:]
        return self.lambda_dice * self.dice(x, t) + self.lambda_focal * self.focal(x, t)


class DiceCELoss(torch.nn.Module):
    def __init__(self, include_background=True, to_onehot_y=False, softmax=False, squared_pred=False, smooth_nr=1e-5, smooth_dr=1e-5,
                 lambda_dice=1.0, lambda_ce=1.0):
        super().__init__()
        self.dice = DiceLoss(include_background, to_onehot_y, softmax, squared_pred, smooth_nr, smooth_dr)
        self.to_onehot_y, self.softmax = to_onehot_y, softmax
        self.lambda_dice, self.lambda_ce = lambda_dice, lambda_ce
        self.cfg = ops.SegLossCfg(L.LOSS_DICE_CE, include_background, squared_pred, smooth_nr, smooth_dr, 0.0, lambda_dice, lambda_ce)

    def forward(self, logits, target):
        if _fusable(logits, target, self.to_onehot_y, self.softmax):
            return _FusedSegLoss.apply(logits, target, self.cfg)
        return self.forward_torch(logits, target)

    def forward_torch(self, logits, target):
        ce = F.cross_entropy(logits.float(), target[:, 0].long())
        return self.lambda_dice * self.dice(logits, target) + self.lambda_ce * ce
