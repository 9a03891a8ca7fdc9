"""Checkpoint helpers: the reference's on-disk format (purejaxql/utils/save_load.py:9-16):
safetensors file of the flax `params` tree flattened with "," as separator, kernels in flax layout
(conv HWIO, dense (in,out)), so files interchange with the reference's load_params."""
from __future__ import annotations

import os
from typing import Dict

import torch
from safetensors.torch import load_file, save_file


def save_params(params: Dict[str, torch.Tensor], filename) -> None:
    """params: {"CNN_0/Conv_0/kernel": tensor, ...} (networks.QNetwork.views keys)."""
    flat = {k.replace("/", ","): v.detach().to("cpu").contiguous().clone() for k, v in params.items()}
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    save_file(flat, str(filename))


def load_params(filename) -> Dict[str, torch.Tensor]:
    return {k.replace(",", "/"): v for k, v in load_file(str(filename)).items()}


def params_to_theta(network, params: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Flat (flax-order) parameter vector from a loaded checkpoint."""
    theta = torch.zeros(network.num_params, dtype=torch.float32)
    for k, (off, n) in network.offsets.items():
        theta[off:off + n] = params[k].reshape(-1).to(torch.float32)
    return theta
