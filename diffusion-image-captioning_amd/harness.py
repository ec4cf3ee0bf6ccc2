"""Epoch driver, LR tables, early-stop rule, log format and checkpoints -- SURVEY.md section 8(f) rows 2-3
(ref CLIP-DDPM.py:451-456, 505-561, 604-631).  A thin host-side harness around `train_func` / `validate` / `sample`.
"""
from __future__ import annotations

import math

import torch

from . import bleu as _bleu
from .config import cfg


# ------------------------------------------------------------------ LR tables (ref :63-70, 451-456)
def cosine_annealing(lr=None, end_lr=None):
    """ref :63-67: 5-epoch cosine from lr to end_lr, repeated 3 times (15 entries)."""
    lr = cfg.LEARNING_RATE if lr is None else lr
    end_lr = cfg.END_LEARNING_RATE if end_lr is None else end_lr
    x = torch.arange(0, 5)
    x = end_lr + (lr - end_lr) * (1 + torch.cos(x / 5 * math.pi)) / 2
    return x.repeat((3,))


def lr_table(scheduler="linspace", lr=None, end_lr=None, epochs=None):
    """The per-epoch learning rates `lrs` of ref :451-456."""
    lr = cfg.LEARNING_RATE if lr is None else lr
    end_lr = cfg.END_LEARNING_RATE if end_lr is None else end_lr
    epochs = cfg.EPOCH_NUM if epochs is None else epochs
    if scheduler == "linspace":
        return torch.linspace(lr, end_lr, epochs)
    if scheduler == "logspace":
        return torch.logspace(torch.tensor([lr]).log10().item(), torch.tensor([end_lr]).log10().item(), epochs)
    if scheduler == "cosine_annealing":
        return cosine_annealing(lr, end_lr)
    raise NotImplementedError(scheduler)


def log_line(epoch, acc, n_batches, val):
    """ref :554 -- same field order, plain floats so the reference's plotting cell (`extract_float`) parses it."""
    a = [float(v) / n_batches for v in acc]
    return (f"epoch {epoch} average x_t_loss, x_1_loss, prob_loss, val losses: {a[0]}, {a[1]}, {a[2]}, "
            f"{float(val[0])}, {float(val[1])}, {float(val[2])}\n")


# ------------------------------------------------------------------ checkpoints (ref :551, 559-561, 570 replace whole-module pickles)
def save_checkpoint(path, model, trainer=None, **extra):
    """State-dict checkpoint: reference-named parameter tensors + (unlike the reference, which drops it on resume :508)
    the AdamW state + the hyper-parameter snapshot."""
    from . import diffusion
    if hasattr(model, "check_ids"):
        model.check_ids(sync=True)               # never persist parameters trained on a batch with an out-of-range token id
    sd = {"params": {k: v.cpu() for k, v in model.params.state_dict().items()}, "n_layers": model.n_layers,
          "cfg": dict(vars(cfg)), "extra": extra, "rng": diffusion.rng_state(model)}     # rng: a resumed run continues the same t / noise / mask streams
    if trainer is not None and hasattr(trainer, "state_dict"):
        o = trainer.state_dict()
        sd["optimizer"] = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in o.items()}
    torch.save(sd, path)


def load_checkpoint(path, model, trainer=None):
    sd = torch.load(path, map_location="cpu", weights_only=False)
    assert sd["n_layers"] == model.n_layers, "checkpoint depth differs from the model"
    model.load_state(sd["params"])
    if trainer is not None and "optimizer" in sd and hasattr(trainer, "load_state_dict"):
        o = sd["optimizer"]
        trainer.load_state_dict({k: (v.to(model.device) if torch.is_tensor(v) else v) for k, v in o.items()})
    if "rng" in sd:
        from . import diffusion
        diffusion.set_rng_state(sd["rng"], model)
    return sd.get("extra", {})


# ------------------------------------------------------------------ epoch loop (ref :505-557)
def fit(model, trainer, train_loader, val_loader, epochs=None, scheduler="linspace", summary=None, checkpoint_path=None,
        train_func=None, validate=None):
    """Mirrors the reference's training section: per-epoch LR from the table (only when END_LEARNING_RATE != LEARNING_RATE,
    ref :520-522), accumulate the four losses lazily (no host sync inside the loop, ref :530-533), validate once per epoch,
    early-stop bookkeeping (`val > EARLY_STOP_RATIO * train`: write "early stop!", save once, keep training; ref :547-553),
    optional dynamic rounding weight (ref :535-536), one log line per epoch."""
    from . import diffusion, parallel
    train_func = train_func or diffusion.train_func
    validate = validate or diffusion.validate
    epochs = cfg.EPOCH_NUM if epochs is None else epochs
    lrs = lr_table(scheduler, epochs=epochs)
    early_stopped = False
    history = []
    model.train()
    if parallel.world_size() > 1:
        # the explicit synchronisation point of the shared timestep stream.  UNCONDITIONAL: the "already shared" flag is rank-local (a seed set or a
        # checkpoint restored on some ranks only would make them disagree on whether to enter the broadcast); re-sharing a stream that is in step is a no-op
        parallel.share_timestep_seed()
    for epoch in range(epochs):
        acc = [0, 0, 0, 0]
        if cfg.END_LEARNING_RATE != cfg.LEARNING_RATE:
            for g in trainer.param_groups:
                g["lr"] = float(lrs[epoch])
        n = 0
        for x in train_loader:
            l, a, b, c = train_func(model, trainer, x)
            acc = [acc[0] + a, acc[1] + b, acc[2] + c, acc[3] + l]      # lazily, on the device (ref :530-533)
            n += 1
            if cfg.DYNAMIC_ROUNDING_WEIGHT > 0:
                cfg.ROUNDING_WEIGHT = float(((acc[0] + acc[1]) / acc[2]).detach()) * cfg.DYNAMIC_ROUNDING_WEIGHT
            if cfg.DEBUG:
                break
        if hasattr(model, "check_ids"):
            model.check_ids(sync=True)           # the epoch's last batch (validate() checks its own)
        if parallel.world_size() > 1:
            parallel.assert_shared_timestep_seed()
        val = validate(model, val_loader)
        n_batches = len(train_loader) if hasattr(train_loader, "__len__") else max(n, 1)      # the reference divides by len(train_loader)
        if float(val[0] + val[1] + val[2]) > cfg.EARLY_STOP_RATIO * float(acc[3]) / max(n_batches, 1):
            if not early_stopped:
                if summary is not None:
                    summary.write("early stop! \n")
                if checkpoint_path:
                    save_checkpoint(checkpoint_path, model, trainer, epoch=epoch, early_stop=True)
            early_stopped = True
        line = log_line(epoch, acc[:3], max(n_batches, 1), val)
        if summary is not None:
            summary.write(line)
        history.append(line)
        if cfg.DEBUG:
            break
    if not early_stopped and checkpoint_path:
        save_checkpoint(checkpoint_path, model, trainer, epoch=epochs - 1, early_stop=False)
    return history


# ------------------------------------------------------------------ BLEU evaluation (ref :604-631)
def caption_references(image_names, captions_by_image):
    """ref :625-627: per image of the batch, every caption of that image as "[CLS] <caption, stripped, lower-cased> [SEP]" -- the
    string form `tokenizer.decode` gives the sampled ids (special tokens kept), so both sides split into the same words."""
    return [["[CLS] " + c.strip().lower() + " [SEP]" for c in captions_by_image[name]] for name in image_names]


@torch.no_grad()
def evaluate_bleu(model, val_loader, references_for, decode=None, steps=5):
    """For each validation batch: sample ids from pure noise (`steps` refinement passes), drop repeated columns
    (`unique_consecutive(dim=-1)`, ref :621), decode, BLEU against that batch's references; average over batches.
    references_for(batch) -> list (per item) of lists of reference token sequences / strings."""
    from . import diffusion
    model.eval()
    batches = []
    for x in val_loader:
        ids = diffusion.dedup_columns(diffusion.sample(model, x["image_clip"], steps=steps))
        cands = [decode(row) if decode is not None else row.tolist() for row in ids]
        batches.append((cands, references_for(x)))
    return _bleu.batch_averaged_bleu(batches)
