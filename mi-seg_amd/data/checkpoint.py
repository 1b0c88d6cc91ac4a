"""Checkpoint import / export in the reference's formats (SURVEY.md 8(b) "state_dict layout"; 8(f) row 4).

  * tune.py:27-38 saves {"epoch", "best_acc", "state_dict": model.state_dict()} (optionally "optimizer" / "scheduler");
  * Lightning checkpoints hold the LitMonai state under "state_dict" with a "model." prefix on every network key;
  * DataParallel / DDP-saved files carry a "module." prefix.
`load_model_state` accepts all three and loads STRICTLY (a renamed or missing key is an error, as in the reference's own loaders);
`export_state` writes the tune.py layout, so files travel in both directions."""
import torch


def strip_prefixes(sd, prefixes=("model.", "module.")):
    out = {}
    for k, v in sd.items():
        changed = True
        while changed:
            changed = False
            for p in prefixes:
                if k.startswith(p):
                    k, changed = k[len(p):], True
        out[k] = v
    return out


def load_model_state(model, ckpt, strict=True):
    """ckpt: path or an already loaded object; returns the checkpoint's metadata (everything except the weights)"""
    obj = torch.load(ckpt, map_location="cpu", weights_only=False) if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__") else ckpt
    sd = obj["state_dict"] if isinstance(obj, dict) and "state_dict" in obj else obj
    sd = strip_prefixes(sd)
    want = model.state_dict()
    sd = {k: v for k, v in sd.items() if k in want or strict}       # non-strict: extra keys (criterion buffers of a Lightning file) are ignored
    model.load_state_dict(sd, strict=strict)
    return {k: v for k, v in obj.items() if k != "state_dict"} if isinstance(obj, dict) and "state_dict" in obj else {}


def export_state(model, path, epoch=0, best_acc=0.0, optimizer=None, scheduler=None, lightning=False):
    """tune.py:27-38 layout; lightning=True prefixes the keys with "model." like a LitMonai checkpoint"""
    sd = {("model." + k if lightning else k): v.detach().cpu() for k, v in model.state_dict().items()}
    obj = {"epoch": epoch, "best_acc": best_acc, "state_dict": sd}
    if optimizer is not None:
        obj["optimizer"] = optimizer.state_dict()
    if scheduler is not None:
        obj["scheduler"] = scheduler.state_dict()
    torch.save(obj, path)
    return path
