"""Q-networks of the PQN hot path on a FLAT fp32 parameter buffer.

Restates the flax modules of the reference -- CNN/QNetwork at
purejaxql/pqn_minatar.py:24-69 and the MLP QNetwork at
purejaxql/pqn_gymnax.py:29-58 -- with flax/optax numerics:
kernels in flax layout (conv HWIO, dense (in,out)), NHWC activations, flatten
order (h,w,c), LayerNorm over the last axis with eps=1e-6, x/255 on the CNN
input, a dummy input-BatchNorm whose params exist but never receive gradient.

All parameters of one seed live in ONE flat buffer (`theta`), with the
gradient in a matching flat buffer, so the optimizer (ops.FlatRAdam), a
checkpoint and the RCCL gradient bucket are each a single contiguous span.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

LN_EPS = 1e-6  # flax nn.LayerNorm default (torch's is 1e-5)


def cnn_param_shapes(obs_shape: Tuple[int, int, int], action_dim: int) -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameter tree of QNetwork(CNN) in flax auto-naming."""
    h, w, c = obs_shape
    flat = (h - 2) * (w - 2) * 16
    return OrderedDict([
        ("BatchNorm_0/scale", (c,)), ("BatchNorm_0/bias", (c,)),
        ("CNN_0/Conv_0/kernel", (3, 3, c, 16)), ("CNN_0/Conv_0/bias", (16,)),
        ("CNN_0/LayerNorm_0/scale", (16,)), ("CNN_0/LayerNorm_0/bias", (16,)),
        ("CNN_0/Dense_0/kernel", (flat, 128)), ("CNN_0/Dense_0/bias", (128,)),
        ("CNN_0/LayerNorm_1/scale", (128,)), ("CNN_0/LayerNorm_1/bias", (128,)),
        ("Dense_0/kernel", (128, action_dim)), ("Dense_0/bias", (action_dim,)),
    ])


def mlp_param_shapes(obs_dim: int, action_dim: int, hidden: int, layers: int):
    shapes = OrderedDict([("BatchNorm_0/scale", (obs_dim,)), ("BatchNorm_0/bias", (obs_dim,))])
    d = obs_dim
    for l in range(layers):
        shapes[f"Dense_{l}/kernel"] = (d, hidden)
        shapes[f"Dense_{l}/bias"] = (hidden,)
        shapes[f"LayerNorm_{l}/scale"] = (hidden,)
        shapes[f"LayerNorm_{l}/bias"] = (hidden,)
        d = hidden
    shapes[f"Dense_{layers}/kernel"] = (d, action_dim)
    shapes[f"Dense_{layers}/bias"] = (action_dim,)
    return shapes


def _trunc_normal(shape, std, gen):
    # variance_scaling(..., "truncated_normal"): truncnorm(-2,2) * std / 0.87962566
    t = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
    return t * (std / 0.87962566103423978)


class QNetwork:
    """Functional Q-network over a flat parameter buffer.

    kind="cnn": pqn_minatar.py:24-69.  kind="mlp": pqn_gymnax.py:29-58.
    """

    def __init__(self, kind: str, obs_shape, action_dim: int, norm_type: str = "layer_norm",
                 norm_input: bool = False, hidden_size: int = 128, num_layers: int = 2, device="cuda"):
        if norm_type not in ("layer_norm", "none", None):
            raise NotImplementedError("NORM_TYPE=batch_norm is outside the current hot-path scope (DESIGN.md)")
        if norm_input:
            raise NotImplementedError("NORM_INPUT=True (input BatchNorm) is outside the current scope (DESIGN.md)")
        self.kind = kind
        self.obs_shape = tuple(obs_shape)
        self.action_dim = int(action_dim)
        self.use_ln = norm_type == "layer_norm"
        self.hidden, self.layers = int(hidden_size), int(num_layers)
        if kind == "cnn":
            self.shapes = cnn_param_shapes(self.obs_shape, action_dim)
        elif kind == "mlp":
            self.shapes = mlp_param_shapes(int(self.obs_shape[0]), action_dim, self.hidden, self.layers)
        else:
            raise ValueError(kind)
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for k, s in self.shapes.items():
            n = math.prod(s)
            self.offsets[k] = (off, n)
            off += n
        self.num_params = off
        self.device = torch.device(device)

    # -- parameters ----------------------------------------------------------------
    def init(self, seed: int) -> torch.Tensor:
        """network.init: he_normal for Conv/Dense(128) of the CNN (pqn_minatar.py:43,48),
        lecun_normal elsewhere (flax Dense default), zero biases, LN/BN scale 1."""
        gen = torch.Generator(device="cpu")
        gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        theta = torch.zeros(self.num_params, dtype=torch.float32)
        for k, s in self.shapes.items():
            off, n = self.offsets[k]
            if k.endswith("/scale"):
                theta[off:off + n] = 1.0
            elif k.endswith("/kernel"):
                fan_in = math.prod(s[:-1])
                he = self.kind == "cnn" and k.startswith("CNN_0/")
                std = math.sqrt((2.0 if he else 1.0) / fan_in)
                theta[off:off + n] = _trunc_normal(s, std, gen).reshape(-1)
        return theta.to(self.device)

    def views(self, theta: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {k: theta[off:off + n].view(self.shapes[k]) for k, (off, n) in self.offsets.items()}

    def to_flax_dict(self, theta: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Checkpoint keys as utils/save_load.py:9-11 writes them (sep=',')."""
        return {k.replace("/", ","): v.detach().clone().cpu() for k, v in self.views(theta).items()}

    # -- forward (torch ops: plumbing path, also the fp32 torch reference) --------------
    def _ln(self, x, scale, bias):
        return F.layer_norm(x, (x.shape[-1],), scale, bias, LN_EPS) if self.use_ln else x

    def apply(self, p: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
        if self.kind == "cnn":
            b = x.shape[0]
            c = x.shape[-1]
            x = x / 255.0
            # VALID 3x3 conv in NHWC as patches @ kernel (keeps (h,w,c) order, no layout flips)
            patches = x.unfold(1, 3, 1).unfold(2, 3, 1)            # [B,8,8,C,3,3]
            patches = patches.permute(0, 1, 2, 4, 5, 3).reshape(b, -1, 9 * c)  # [B,64,(ky,kx,c)]
            y = patches @ p["CNN_0/Conv_0/kernel"].reshape(9 * c, 16) + p["CNN_0/Conv_0/bias"]
            y = torch.relu(self._ln(y, p["CNN_0/LayerNorm_0/scale"], p["CNN_0/LayerNorm_0/bias"]))
            y = y.reshape(b, -1)                                    # (h,w,c) flatten
            y = y @ p["CNN_0/Dense_0/kernel"] + p["CNN_0/Dense_0/bias"]
            y = torch.relu(self._ln(y, p["CNN_0/LayerNorm_1/scale"], p["CNN_0/LayerNorm_1/bias"]))
            return y @ p["Dense_0/kernel"] + p["Dense_0/bias"]
        y = x
        for l in range(self.layers):
            y = y @ p[f"Dense_{l}/kernel"] + p[f"Dense_{l}/bias"]
            y = torch.relu(self._ln(y, p[f"LayerNorm_{l}/scale"], p[f"LayerNorm_{l}/bias"]))
        return y @ p[f"Dense_{self.layers}/kernel"] + p[f"Dense_{self.layers}/bias"]


class FlatParams:
    """theta / grad flat buffers + leaf views whose .grad alias the flat grad."""

    def __init__(self, net: QNetwork, theta: torch.Tensor):
        self.net = net
        self.theta = theta
        self.grad = torch.zeros_like(theta)
        self.leaves: Dict[str, torch.Tensor] = {}
        gviews = net.views(self.grad)
        for k, v in net.views(theta).items():
            leaf = v.detach()
            leaf.requires_grad_(True)
            leaf.grad = gviews[k]
            self.leaves[k] = leaf

    def zero_grad(self):
        self.grad.zero_()
