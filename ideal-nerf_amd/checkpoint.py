"""Checkpoint files of the reference trainer (audio_exp_nerf.py:498-525,584-592;
TorsoNeRF/train_torso.py:565-573): ``head.tar`` = {'global_step', 'model_state_dict',
'optimizer', 'latent_codes'}, resume from the naturally-sorted last ``*.tar`` of the run
directory, and warm start from AD-NeRF checkpoints (``ft_path``) whose first / skip / view layers
have other input widths and are dropped.
"""
import os
import re

import torch


def natural_key(s: str):
    return [int(t) if t.isdigit() else t.lower() for t in re.split(r"(\d+)", s)]


def latest_checkpoint(run_dir: str):
    """natsorted(listdir)[-1] among names containing '.tar' (audio_exp_nerf.py:516-518), or None."""
    if not os.path.isdir(run_dir):
        return None
    names = sorted((f for f in os.listdir(run_dir) if ".tar" in f), key=natural_key)
    return os.path.join(run_dir, names[-1]) if names else None


def save_checkpoint(path: str, network, optimizer, latent_codes, global_step: int):
    """audio_exp_nerf.py:586-591."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({"global_step": global_step, "model_state_dict": network.state_dict(),
                "optimizer": optimizer.state_dict() if optimizer is not None else None,
                "latent_codes": latent_codes.data}, path)


def load_checkpoint(path: str, network, optimizer=None, map_location=None):
    """-> (global_step, latent_codes).  Keys and strictness as upstream (:520-524)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    network.load_state_dict(ckpt["model_state_dict"])
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    return ckpt["global_step"], ckpt["latent_codes"]


ADNERF_DROP = ("pts_linears.0.weight", "pts_linears.5.weight", "views_linears.0.weight")


def load_adnerf_finetune(path_or_dict, network, map_location=None):
    """Warm start from an AD-NeRF checkpoint (audio_exp_nerf.py:498-514): the coarse / fine
    FaceNeRF weights minus the three layers whose input width differs, and the audio nets,
    all with strict=False."""
    ckpt = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location=map_location,
                                                                          weights_only=False)
    coarse = {k: v for k, v in ckpt["network_fn_state_dict"].items() if k not in ADNERF_DROP}
    fine = {k: v for k, v in ckpt["network_fine_state_dict"].items() if k not in ADNERF_DROP}
    network.face_nerf_coarse.load_state_dict(coarse, strict=False)
    network.face_nerf_fine.load_state_dict(fine, strict=False)
    network.aud_net.load_state_dict(ckpt["network_audnet_state_dict"], strict=False)
    network.aud_att_net.load_state_dict(ckpt["network_audattnet_state_dict"], strict=False)
    return network
