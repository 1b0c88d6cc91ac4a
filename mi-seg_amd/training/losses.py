"""Dice / Focal / CE losses with MONAI 1.1.0 semantics as used by the reference (networks/lightning_monai.py:46-67,
utils/training_utils.py:6-33).  Restated from MONAI's public API (MONAI is a pinned third-party dependency of the reference,
monai~=1.1.0, not present here): PARITY UNPINNED by any reference test -- SURVEY.md Appendix B.

These run after the hot path (on the fp32 logits); they are plain torch ops today (SURVEY 8(f) ranks fused loss kernels next)."""
import torch
import torch.nn.functional as F


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


class DiceFocalLoss(torch.nn.Module):
    def __init__(self, include_background=True, to_onehot_y=False, softmax=False, squared_pred=False, smooth_nr=1e-5, smooth_dr=1e-5, gamma=2.0,
                 lambda_dice=1.0, lambda_focal=1.0):
        super().__init__()
        self.dice = DiceLoss(include_background, to_onehot_y, softmax, squared_pred, smooth_nr, smooth_dr)
        self.focal = FocalLoss(include_background, to_onehot_y, gamma)
        self.lambda_dice, self.lambda_focal = lambda_dice, lambda_focal

    def forward(self, logits, target):
        return self.lambda_dice * self.dice(logits, target) + self.lambda_focal * self.focal(logits, target)


class DiceCELoss(torch.nn.Module):
    def __init__(self, include_background=True, to_onehot_y=False, softmax=False, squared_pred=False, smooth_nr=1e-5, smooth_dr=1e-5,
                 lambda_dice=1.0, lambda_ce=1.0):
        super().__init__()
        self.dice = DiceLoss(include_background, to_onehot_y, softmax, squared_pred, smooth_nr, smooth_dr)
        self.lambda_dice, self.lambda_ce = lambda_dice, lambda_ce

    def forward(self, logits, target):
        ce = F.cross_entropy(logits.float(), target[:, 0].long())
        return self.lambda_dice * self.dice(logits, target) + self.lambda_ce * ce
