"""Dice metric / AsDiscrete with MONAI 1.1.0 semantics (reference lightning_monai.py:68-79,190-195).  Parity unpinned (Appendix B)."""
import torch
import torch.nn.functional as F


def as_discrete_argmax_onehot(logits, num_classes):
    """AsDiscrete(argmax=True, to_onehot=C) on a batched [B,C,...] tensor."""
    idx = logits.argmax(dim=1)
    return F.one_hot(idx, num_classes).movedim(-1, 1).to(torch.float32)


def as_discrete_onehot(label, num_classes):
    return F.one_hot(label[:, 0].long(), num_classes).movedim(-1, 1).to(torch.float32)


def dice_metric(y_pred_onehot, y_onehot):
    """per (b, c): 2|y & yhat| / (|y| + |yhat|), NaN where |y| == 0  (DiceMetric(include_background=True, get_not_nans=True))."""
    dims = tuple(range(2, y_pred_onehot.dim()))
    inter = (y_pred_onehot * y_onehot).sum(dims)
    y_o = y_onehot.sum(dims)
    den = y_o + y_pred_onehot.sum(dims)
    out = 2.0 * inter / den.clamp(min=1e-30)
    out = torch.where(den > 0, out, torch.ones_like(out))
    return torch.where(y_o > 0, out, torch.full_like(out, float("nan")))


def dice_from_logits(logits, label, num_classes):
    """AsDiscrete(argmax, to_onehot) on the logits + AsDiscrete(to_onehot) on the label + DiceMetric, as LitMonai._shared_eval chains them
    (reference lightning_monai.py:190-195).  On a HIP device: one pass over logits + labels with integer counters (csrc/training.hip,
    bit-reproducible); on CPU tensors (--infer_cpu) the torch arithmetic above."""
    if logits.is_cuda and logits.dtype == torch.float32 and logits.shape[1] == num_classes and num_classes <= 64:
        from ..hip import ops
        return ops.dice_metric(logits.contiguous(), label.to(logits.device))
    return dice_metric(as_discrete_argmax_onehot(logits, num_classes), as_discrete_onehot(label, num_classes))
