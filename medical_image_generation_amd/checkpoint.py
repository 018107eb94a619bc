"""Checkpoints in the reference's `.pth` layout (train_ldm.py:466-505, train_ddpm.py / train_autoencoder.py alike):

    {'epoch', 'network_state_dict', 'optimizer_state_dict', 'validation_loss'[, 'scheduler_state_dict']}

`network_state_dict` is the module's `state_dict()` (reference names and shapes); `optimizer_state_dict` is written in
the layout `torch.optim.Adam[W].state_dict()` has -- per-parameter `step` / `exp_avg` / `exp_avg_sq` keyed by the index of the
parameter in `model.parameters()` order (the order of the reference's modules: tests/test_abi_cpu.py), no entry for parameters that
never receive a gradient (`proj_attn.*`), one param group -- although the fused optimizer of this package keeps its moments in two
flat buffers over the parameter arena.  A checkpoint written by the reference's trainers therefore resumes here and the other way
round.  Files are written with torch.save and read with torch.load(weights_only=True): tensors, numbers, lists and dicts only.
"""
from __future__ import annotations

import os

import torch


def _load(path):
    """torch.load(weights_only=True) with numpy scalars admitted: the reference's trainers store `validation_loss` as whatever their
    validation loop returned, which is a numpy float when it came out of np.mean (tensors, numbers, containers and these only)."""
    import numpy as np
    allow = [np.dtype, np.float64, np.float32, np.int64] + [type(np.dtype(t)) for t in (np.float64, np.float32, np.int64)]
    try:
        allow.append(np._core.multiarray.scalar)
    except AttributeError:  # numpy < 2
        allow.append(np.core.multiarray.scalar)
    with torch.serialization.safe_globals(allow):
        return torch.load(path, map_location="cpu", weights_only=True)


def optimizer_state_dict(trainer) -> dict:
    """torch.optim.Adam / AdamW state of the trainer's fused optimizer (CPU tensors, like a torch checkpoint after map_location='cpu')."""
    a, model = trainer.arena, trainer.model
    trainable = {n for n, _, t in model._entries if t}
    step = trainer.step_count.detach().cpu().reshape(()).clone()
    m, v = trainer.exp_avg.detach().cpu(), trainer.exp_avg_sq.detach().cpu()
    state, names = {}, [n for n, _ in model.named_parameters()]
    if float(step) > 0:  # torch creates a parameter's state at its first update
        for i, n in enumerate(names):
            if n in trainable:
                state[i] = {"step": step.clone(), "exp_avg": a.view(n, m).clone(), "exp_avg_sq": a.view(n, v).clone()}
    group = {"lr": trainer.lr, "betas": tuple(trainer.betas), "eps": trainer.eps, "weight_decay": trainer.weight_decay, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "decoupled_weight_decay": bool(trainer.decoupled), "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(trainer, sd: dict) -> None:
    a, model = trainer.arena, trainer.model
    names = [n for n, _ in model.named_parameters()]
    groups = sd["param_groups"]
    if sum(len(g["params"]) for g in groups) != len(names):
        raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")  # torch's message
    g0 = groups[0]
    trainer.lr, trainer.betas, trainer.eps, trainer.weight_decay = float(g0["lr"]), tuple(g0["betas"]), float(g0["eps"]), float(g0["weight_decay"])
    ids = [i for g in groups for i in g["params"]]  # saved id of the k-th parameter
    m, v = torch.zeros_like(trainer.exp_avg, device="cpu"), torch.zeros_like(trainer.exp_avg_sq, device="cpu")
    trainable = {n for n, _, t in model._entries if t}
    steps = set()
    for k, n in enumerate(names):
        st = sd["state"].get(ids[k])
        if st is None:
            continue
        if n not in trainable:
            raise ValueError(f"optimizer state for {n}, which never receives a gradient in this network")
        a.view(n, m).copy_(st["exp_avg"].to(torch.float32))
        a.view(n, v).copy_(st["exp_avg_sq"].to(torch.float32))
        steps.add(float(st["step"]))
    if len(steps) > 1:
        raise ValueError(f"parameters at different step counts {sorted(steps)}: the fused optimizer keeps one")
    trainer.exp_avg.copy_(m)
    trainer.exp_avg_sq.copy_(v)
    trainer.step_count.fill_(steps.pop() if steps else 0.0)


def save_model(trainer, results_path, epoch, validation_loss, scheduler=None) -> str:
    """train_ldm.py:466-490: checkpoints/last_model.pth always, checkpoints/best_model.pth when the validation loss improved."""
    save_path = os.path.join(results_path, "checkpoints")
    os.makedirs(save_path, exist_ok=True)
    checkpoint = {"epoch": epoch, "network_state_dict": {k: v.detach().cpu() for k, v in trainer.model.state_dict().items()},
                  "optimizer_state_dict": optimizer_state_dict(trainer), "validation_loss": validation_loss}
    if scheduler:
        checkpoint["scheduler_state_dict"] = scheduler.state_dict()
    last = os.path.join(save_path, "last_model.pth")
    torch.save(checkpoint, last)
    best = os.path.join(save_path, "best_model.pth")
    if os.path.isfile(best):
        best_loss = _load(best).get("validation_loss", float("inf"))
        if validation_loss < best_loss:
            torch.save(checkpoint, best)
    else:
        torch.save(checkpoint, best)
    return last


def load_model(trainer, load_model_path, load_optimizer=True, lr_scheduler=None, for_training=False):
    """train_ldm.py:492-505.  Returns the epoch to resume at when for_training."""
    checkpoint = _load(load_model_path)
    trainer.model.load_state_dict(checkpoint["network_state_dict"])
    if load_optimizer:
        load_optimizer_state_dict(trainer, checkpoint["optimizer_state_dict"])
    if lr_scheduler and "scheduler_state_dict" in checkpoint:
        lr_scheduler.load_state_dict(checkpoint["scheduler_state_dict"])
    if for_training:
        return checkpoint["epoch"] + 1
