"""Checkpoint layout of the reference (exp_runner.py:355-386): one torch.save dict with the keys
`nerf`, `sdf_network_fine`, `variance_network_fine`, `color_network_fine`, `optimizer`, `iter_step`, written
to `checkpoints/ckpt_{iter:06d}.pth`.  The drop-in modules keep the reference's parameter names, so these
helpers only fix the dict layout; files written by the reference load here and vice versa."""
from __future__ import annotations

import os

import torch

KEYS = ("nerf", "sdf_network_fine", "variance_network_fine", "color_network_fine", "optimizer", "iter_step")


def save_checkpoint(path, nerf, sdf_network, deviation_network, color_network, optimizer, iter_step):
    """exp_runner.py:373-386."""
    ckpt = {
        "nerf": nerf.state_dict() if nerf is not None else {},
        "sdf_network_fine": sdf_network.state_dict(),
        "variance_network_fine": deviation_network.state_dict(),
        "color_network_fine": color_network.state_dict(),
        "optimizer": optimizer.state_dict() if optimizer is not None else {},
        "iter_step": int(iter_step),
    }
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ckpt, path)
    return path


def load_checkpoint(path, nerf, sdf_network, deviation_network, color_network, optimizer=None, map_location=None):
    """exp_runner.py:355-370.  `weights_only=True`: nothing in the file is executed."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    missing = [k for k in KEYS if k not in ckpt]
    if missing:
        raise KeyError(f"{path}: not an RNb-NeuS checkpoint (missing {missing})")
    if nerf is not None and ckpt["nerf"]:
        nerf.load_state_dict(ckpt["nerf"])
    sdf_network.load_state_dict(ckpt["sdf_network_fine"])
    deviation_network.load_state_dict(ckpt["variance_network_fine"])
    color_network.load_state_dict(ckpt["color_network_fine"])
    if optimizer is not None and ckpt["optimizer"]:
        optimizer.load_state_dict(ckpt["optimizer"])
    return int(ckpt["iter_step"])


def latest_checkpoint(checkpoint_dir, end_iter=None):
    """The reference resumes from the last ckpt_*.pth (sorted by name) with iter <= end_iter
    (exp_runner.py:130-142)."""
    if not os.path.isdir(checkpoint_dir):
        return None
    names = []
    for n in os.listdir(checkpoint_dir):
        if n.startswith("ckpt_") and n.endswith(".pth"):
            try:
                it = int(n[5:-4])
            except ValueError:
                continue
            if end_iter is None or it <= end_iter:
                names.append(n)
    names.sort()
    return os.path.join(checkpoint_dir, names[-1]) if names else None
