"""LitMonai -- the training-harness surface of reference networks/lightning_monai.py:19-305.

PyTorch-Lightning is not installed in this image; when it is, LitMonai subclasses ``LightningModule`` exactly like the reference,
otherwise it is an ``nn.Module`` with the same constructor, ``from_argparse_args``, ``training_step`` / ``validation_step`` /
``test_step`` / ``configure_optimizers`` and no-op logging hooks, so a hand-written loop (bench.py, tests) can drive it.
Loss / metric / inferer arithmetic is MONAI's, restated in mi-seg_amd/training (parity unpinned)."""
import inspect
from functools import partial
from typing import Sequence, Union

import numpy as np
import torch
import torch.nn as nn

from ..training.inferer import sliding_window_inference
from ..training.losses import DiceCELoss, DiceFocalLoss
from ..training.metrics import dice_from_logits
from ..training.schedulers import WarmupCosineSchedule
from .utils.utils import model_from_argparse_args

try:  # pragma: no cover - not available in the build image
    from pytorch_lightning import LightningModule as _Base
    _HAS_PL = True
except Exception:
    _Base, _HAS_PL = nn.Module, False


class LitMonai(_Base):
    def __init__(self, model: nn.Module, out_channels: int, criterion: str = "dice_focal", squared_pred: bool = True, smooth_nr: float = 0.0,
                 smooth_dr: float = 1e-6, learning_rate: float = 1e-4, optim_name: str = "adamw", reg_weight: float = 1e-5, momentum: float = 0.99,
                 roi_size: Union[Sequence[int], int] = (96, 96, 96), infer_overlap: float = 0.5, sw_batch_size: int = 1, infer_cpu: bool = False,
                 batch_size: int = 1, scheduler: str = "reduce_on_plateau", warmup_epochs=None, patience=None, check_val_every_n_epoch=None,
                 max_epochs: int = 5000, t_max: int = 200, cycles: float = 1, include_background: bool = False, **kwargs):
        super().__init__()
        self.model = model
        if criterion == "dice_focal":      # squared_pred hard-coded True like the reference (:53)
            self.criterion = DiceFocalLoss(include_background=include_background, to_onehot_y=True, softmax=True, squared_pred=True,
                                           smooth_nr=smooth_nr, smooth_dr=smooth_dr)
        elif criterion == "dice_ce":
            self.criterion = DiceCELoss(include_background=include_background, to_onehot_y=True, softmax=True, squared_pred=squared_pred,
                                        smooth_nr=smooth_nr, smooth_dr=smooth_dr)
        else:
            raise ValueError("Criterion {} not implemented, please chose another optimizer.".format(criterion))
        self.out_channels = out_channels
        self.learning_rate, self.batch_size, self.optim_name = learning_rate, batch_size, optim_name
        self.reg_weight, self.momentum, self.infer_cpu = reg_weight, momentum, infer_cpu
        self.model_inferer = partial(sliding_window_inference, predictor=self.model, roi_size=roi_size, overlap=infer_overlap,
                                     sw_batch_size=sw_batch_size, device=torch.device("cpu") if infer_cpu else None)
        self.scheduler, self.warmup_epochs, self.patience = scheduler, warmup_epochs, patience
        self.check_val_every_n_epoch, self.max_epochs, self.t_max, self.cycles = check_val_every_n_epoch, max_epochs, t_max, cycles
        self.__dict__.update(kwargs)
        self.logged = {}
        if _HAS_PL:  # pragma: no cover
            self.save_hyperparameters(ignore=["model", "criterion", "model_inferer", "roi_size"])

    # ---- logging hooks (Lightning provides them; the stand-alone variant records the last values)
    if not _HAS_PL:
        def log(self, name, value, **_kw):
            self.logged[name] = float(value.detach()) if isinstance(value, torch.Tensor) else float(value)

        def log_dict(self, d, **_kw):
            for k, v in d.items():
                self.log(k, v)

    @classmethod
    def from_argparse_args(cls, args):
        model = model_from_argparse_args(args)
        params = vars(args)
        known = inspect.signature(cls.__init__).parameters
        extra = {k: v for k, v in params.items() if k not in known}
        return cls(model=model, out_channels=args.out_channels, criterion=args.criterion, squared_pred=args.squared_dice, smooth_nr=args.smooth_nr,
                   smooth_dr=args.smooth_dr, learning_rate=args.lr, optim_name=args.optim_name, reg_weight=args.reg_weight,
                   roi_size=(args.roi_x, args.roi_y, args.roi_z), infer_overlap=args.infer_overlap, sw_batch_size=args.sw_batch_size,
                   infer_cpu=args.infer_cpu, batch_size=args.batch_size, scheduler=args.scheduler, warmup_epochs=args.warmup_epochs,
                   patience=args.patience_scheduler, check_val_every_n_epoch=getattr(args, "check_val_every_n_epoch", None),
                   max_epochs=getattr(args, "max_epochs", 5000), t_max=args.t_max, cycles=args.cycles,
                   include_background=not args.no_include_background, **extra)

    def forward(self, x, modalities=None):
        return self.model(x, modalities) if modalities is not None else self.model(x)

    def training_step(self, batch, batch_idx):
        image, label = batch["image"], batch["label"]
        modality = batch["modality"] if "modality" in batch.keys() else None
        logits = self.model(image, modality)
        loss = self.criterion(logits, label)
        self.log("train/loss", loss, on_step=False, on_epoch=True, prog_bar=True, logger=True, sync_dist=True, batch_size=self.batch_size)
        return {"loss": loss}

    def validation_step(self, batch, batch_idx):
        return self._shared_eval(batch, batch_idx, "val")

    def test_step(self, batch, batch_idx):
        return self._shared_eval(batch, batch_idx, "test")

    def validation_epoch_end(self, outputs):
        self._shared_eval_end(outputs, "val")

    def test_epoch_end(self, outputs):
        self._shared_eval_end(outputs, "test")

    def _shared_eval(self, batch, batch_idx, prefix):
        image, label = batch["image"], batch["label"]
        modality = batch["modality"] if "modality" in batch.keys() else None
        arena = next((getattr(p, "_miseg_arena", None) for p in self.model.parameters()), None)
        if arena is not None:              # validation between optimiser steps: the arena's weight copies must be current
            arena.refresh_weights()
        logits = self.model_inferer(image, modalities=modality)
        if self.infer_cpu:
            label = label.cpu()
        loss = self.criterion(logits, label.to(logits.device))
        accuracy = dice_from_logits(logits, label.to(logits.device), self.out_channels)
        avg = torch.nanmean(accuracy)
        per_class = torch.nanmean(accuracy, dim=0)
        self.log_dict({f"{prefix}/accuracy/class_{i}": a for i, a in enumerate(per_class)}, on_epoch=True, logger=True, sync_dist=True, batch_size=1)
        self.log_dict({f"{prefix}/loss/avg": loss, f"{prefix}/accuracy/avg": avg.item()}, on_epoch=True, prog_bar=True, logger=True, sync_dist=True,
                      batch_size=1)
        return {"loss": loss, "accuracy": avg, "modality": modality}

    def _shared_eval_end(self, outputs, prefix):
        """per-modality means (reference :221-248)."""
        acc = np.array([float(o["accuracy"]) for o in outputs])
        los = np.array([float(o["loss"]) for o in outputs])
        mod = np.array([int(torch.as_tensor(o["modality"]).reshape(-1)[0]) for o in outputs])
        self.log_dict({f"{prefix}/accuracy/modality_{int(m)}": float(np.nanmean(acc[mod == m])) for m in np.unique(mod)}, logger=True, sync_dist=True)
        self.log_dict({f"{prefix}/loss/modality_{int(m)}": float(np.nanmean(los[mod == m])) for m in np.unique(mod)}, logger=True, sync_dist=True)

    def configure_fused_optimizer(self, arena):
        """the optimiser of configure_optimizers() as ONE kernel launch per step over the model's gradient arena
        (training/optim.py::ArenaOptimizer); same hyper-parameters, same "no gradient => untouched" rule"""
        from ..training.optim import ArenaOptimizer
        if self.optim_name not in ("adam", "adamw", "sgd"):
            raise ValueError("Optimization {} not implemented, please chose another optimizer.".format(self.optim_name))
        return ArenaOptimizer(arena, kind=self.optim_name, lr=self.learning_rate, weight_decay=self.reg_weight, momentum=self.momentum)

    def configure_optimizers(self):
        if self.optim_name == "adam":
            opt = torch.optim.Adam(self.parameters(), lr=self.learning_rate, weight_decay=self.reg_weight)
        elif self.optim_name == "adamw":
            opt = torch.optim.AdamW(self.parameters(), lr=self.learning_rate, weight_decay=self.reg_weight)
        elif self.optim_name == "sgd":
            opt = torch.optim.SGD(self.parameters(), lr=self.learning_rate, momentum=self.momentum, nesterov=True, weight_decay=self.reg_weight)
        else:
            raise ValueError("Optimization {} not implemented, please chose another optimizer.".format(self.optim_name))
        if self.scheduler == "warmup_cosine":
            sch = WarmupCosineSchedule(optimizer=opt, warmup_steps=self.warmup_epochs, t_total=self.max_epochs, cycles=self.cycles)
        elif self.scheduler == "cosine":
            sch = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer=opt, T_max=self.t_max)
        elif self.scheduler == "reduce_on_plateau":
            sch = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer=opt, patience=self.patience)
        else:
            raise ValueError("Scheduler {} not implemented, please chose another optimizer.".format(self.scheduler))
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sch, "monitor": "val/loss/avg", "frequency": self.check_val_every_n_epoch}}
