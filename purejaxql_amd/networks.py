"""Q-networks of the PQN hot path on a FLAT fp32 parameter buffer.

Restates the flax modules of the reference -- CNN/QNetwork at
purejaxql/pqn_minatar.py:24-69 and the MLP QNetwork at
purejaxql/pqn_gymnax.py:29-58 -- with flax/optax numerics:
kernels in flax layout (conv HWIO, dense (in,out)), NHWC activations, flatten
order (h,w,c), LayerNorm over the last axis with eps=1e-6, x/255 on the CNN
input, a dummy input-BatchNorm whose params exist but never receive gradient;
NORM_TYPE = layer_norm | batch_norm | none and NORM_INPUT as in the reference
(BatchNorm with flax defaults and a `batch_stats` collection of running moments).

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
BN_EPS = 1e-5  # flax nn.BatchNorm defaults: epsilon 1e-5, momentum 0.99, statistics over all axes but the last
BN_MOMENTUM = 0.99


def _norm_kind(norm_type) -> str:
    return norm_type if norm_type in ("layer_norm", "batch_norm") else "none"   # pqn_minatar.py:31-36


def cnn_norm_names(norm_type) -> Tuple[str, str]:
    """flax auto-names of the two normalize(x) modules inside CNN (pqn_minatar.py:31-50); '' = identity."""
    k = _norm_kind(norm_type)
    if k == "layer_norm":
        return "CNN_0/LayerNorm_0", "CNN_0/LayerNorm_1"
    if k == "batch_norm":
        return "CNN_0/BatchNorm_0", "CNN_0/BatchNorm_1"
    return "", ""


def bn_module(renorm: bool) -> str:
    """flax auto-name stem of the batch normalisation module: nn.BatchNorm in pqn_minatar.py / pqn_gymnax.py,
    BatchRenorm (utils/batch_renorm.py) in pqn_craftax.py:33-62."""
    return "BatchRenorm" if renorm else "BatchNorm"


def mlp_norm_name(norm_type, l: int, renorm: bool = False) -> str:
    """normalize(x) of hidden layer l in the MLP QNetwork (pqn_gymnax.py:44-54): BatchNorm_0 is the input /
    dummy BatchNorm, so the hidden-layer BatchNorms are BatchNorm_1..  (BatchRenorm_* in the Craftax script.)"""
    k = _norm_kind(norm_type)
    return f"LayerNorm_{l}" if k == "layer_norm" else (f"{bn_module(renorm)}_{l + 1}" if k == "batch_norm" else "")


def cnn_param_shapes(obs_shape: Tuple[int, int, int], action_dim: int,
                     norm_type="layer_norm") -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameter tree of QNetwork(CNN) in flax auto-naming."""
    h, w, c = obs_shape
    flat = (h - 2) * (w - 2) * 16
    n0, n1 = cnn_norm_names(norm_type)
    shapes = OrderedDict([("BatchNorm_0/scale", (c,)), ("BatchNorm_0/bias", (c,)),
                          ("CNN_0/Conv_0/kernel", (3, 3, c, 16)), ("CNN_0/Conv_0/bias", (16,))])
    if n0:
        shapes[n0 + "/scale"], shapes[n0 + "/bias"] = (16,), (16,)
    shapes["CNN_0/Dense_0/kernel"], shapes["CNN_0/Dense_0/bias"] = (flat, 128), (128,)
    if n1:
        shapes[n1 + "/scale"], shapes[n1 + "/bias"] = (128,), (128,)
    shapes["Dense_0/kernel"], shapes["Dense_0/bias"] = (128, action_dim), (action_dim,)
    return shapes


def mlp_param_shapes(obs_dim: int, action_dim: int, hidden: int, layers: int, norm_type="layer_norm", renorm=False):
    bn0 = bn_module(renorm) + "_0"
    shapes = OrderedDict([(bn0 + "/scale", (obs_dim,)), (bn0 + "/bias", (obs_dim,))])
    d = obs_dim
    for l in range(layers):
        shapes[f"Dense_{l}/kernel"] = (d, hidden)
        shapes[f"Dense_{l}/bias"] = (hidden,)
        n = mlp_norm_name(norm_type, l, renorm)
        if n:
            shapes[n + "/scale"] = (hidden,)
            shapes[n + "/bias"] = (hidden,)
        d = hidden
    shapes[f"Dense_{layers}/kernel"] = (d, action_dim)
    shapes[f"Dense_{layers}/bias"] = (action_dim,)
    return shapes


def batch_stats_shapes(kind: str, obs_shape, hidden: int, layers: int, norm_type, norm_input: bool, renorm: bool = False):
    """The `batch_stats` collection (running mean / var of every BatchNorm whose output is used).  The
    dummy input BatchNorm of NORM_INPUT=False (pqn_minatar.py:63-65) only ever produces dead state -- its
    output is discarded and checkpoints hold `params` only (:467) -- so its statistics are not tracked."""
    feats = OrderedDict()
    if norm_input:
        feats[bn_module(renorm) + "_0"] = int(obs_shape[-1])
    if _norm_kind(norm_type) == "batch_norm":
        if kind == "cnn":
            n0, n1 = cnn_norm_names(norm_type)
            feats[n0], feats[n1] = 16, 128
        else:
            for l in range(layers):
                feats[mlp_norm_name(norm_type, l, renorm)] = hidden
    shapes = OrderedDict()
    for name, f in feats.items():
        shapes[name + "/mean"] = (f,)
        shapes[name + "/var"] = (f,)
        if renorm:   # BatchRenorm also counts its train-mode calls (utils/batch_renorm.py:71-76,116); r_max / d_max are constants
            shapes[name + "/steps"] = ()
    return shapes


# BatchRenorm of the Craftax script (purejaxql/utils/batch_renorm.py:19-131; pqn_craftax.py:44-53): BatchNorm whose
# train-mode moments are pulled towards the running ones by clipped, gradient-free factors r, d once 1000 steps of
# warm-up have passed.  Building block of the Craftax row; not selected by the MinAtar / gymnax yaml.
BRN_EPS, BRN_MOMENTUM, BRN_R_MAX, BRN_D_MAX, BRN_WARMUP = 1e-3, 0.999, 3.0, 5.0, 1000


def batch_renorm(x: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, stats: Dict[str, torch.Tensor], train: bool,
                 new_stats: Dict[str, torch.Tensor] = None) -> torch.Tensor:
    """stats: {"mean", "var", "steps"} (running moments, step counter).  train=True normalises with the (renormalised)
    batch moments and writes the updated statistics into new_stats; train=False uses the running moments."""
    if not train:
        mean, var = stats["mean"], stats["var"]
    else:
        axes = tuple(range(x.dim() - 1))
        bmean = x.mean(dim=axes)
        bvar = torch.clamp((x * x).mean(dim=axes) - bmean * bmean, min=0.0)        # flax fast variance
        ra_std = torch.sqrt(stats["var"] + BRN_EPS)
        r = torch.clamp((torch.sqrt(bvar + BRN_EPS) / ra_std).detach(), 1.0 / BRN_R_MAX, BRN_R_MAX)    # (:100-101)
        d = torch.clamp(((bmean - stats["mean"]) / ra_std).detach(), -BRN_D_MAX, BRN_D_MAX)             # (:102-103)
        # blended on the device like the reference (:107-109): no host read of the step counter
        warmed = (torch.as_tensor(stats["steps"], device=x.device) >= BRN_WARMUP).to(x.dtype)
        var = warmed * (bvar / (r * r)) + (1.0 - warmed) * bvar
        mean = warmed * (bmean - d * torch.sqrt(bvar) / r) + (1.0 - warmed) * bmean
        if new_stats is not None:
            new_stats["mean"] = BRN_MOMENTUM * stats["mean"] + (1.0 - BRN_MOMENTUM) * bmean.detach()
            new_stats["var"] = BRN_MOMENTUM * stats["var"] + (1.0 - BRN_MOMENTUM) * bvar.detach()
            new_stats["steps"] = stats["steps"] + 1
    return (x - mean) * (torch.rsqrt(var + BRN_EPS) * scale) + bias


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
                 norm_input: bool = False, hidden_size: int = 128, num_layers: int = 2, device="cuda",
                 renorm: bool = False):
        self.kind = kind
        self.obs_shape = tuple(obs_shape)
        self.action_dim = int(action_dim)
        self.norm = _norm_kind(norm_type)
        self.norm_type = norm_type
        self.norm_input = bool(norm_input)
        self.use_ln = self.norm == "layer_norm"
        self.hidden, self.layers = int(hidden_size), int(num_layers)
        # renorm: the QNetwork of pqn_craftax.py:33-62 -- BatchRenorm wherever the gymnax script has nn.BatchNorm
        self.renorm = bool(renorm)
        if self.renorm and kind != "mlp":
            raise ValueError("BatchRenorm is the Craftax script's MLP Q-network (pqn_craftax.py:33-62)")
        if kind == "cnn":
            self.shapes = cnn_param_shapes(self.obs_shape, action_dim, norm_type)
        elif kind == "mlp":
            self.shapes = mlp_param_shapes(int(self.obs_shape[0]), action_dim, self.hidden, self.layers, norm_type, self.renorm)
        else:
            raise ValueError(kind)
        self.stats_shapes = batch_stats_shapes(kind, self.obs_shape, self.hidden, self.layers, norm_type, norm_input, self.renorm)
        self.has_batch_stats = len(self.stats_shapes) > 0
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

    def init_batch_stats(self) -> Dict[str, torch.Tensor]:
        """variables["batch_stats"] at init: running mean 0, running var 1 (flax nn.BatchNorm)."""
        return {k: (torch.ones if k.endswith("/var") else torch.zeros)(
                    s, dtype=torch.int32 if k.endswith("/steps") else torch.float32, device=self.device)
                for k, s in self.stats_shapes.items()}

    def views(self, theta: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {k: theta[off:off + n].view(self.shapes[k]) for k, (off, n) in self.offsets.items()}

    def to_flax_dict(self, theta: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Checkpoint keys as utils/save_load.py:9-11 writes them (sep=',')."""
        return {k.replace("/", ","): v.detach().clone().cpu() for k, v in self.views(theta).items()}

    # -- forward (torch ops: plumbing path, also the fp32 torch reference) --------------
    def _bn(self, x, p, name, train, stats, new_stats):
        """flax nn.BatchNorm(use_running_average=not train): statistics over every axis but the last, fast
        variance E[x^2]-E[x]^2 clamped at 0, running <- 0.99*running + 0.01*batch.  (BatchRenorm in the Craftax network.)"""
        if self.renorm:
            st = {k: stats[name + "/" + k] for k in ("mean", "var", "steps")}
            ns = {} if (train and new_stats is not None) else None
            y = batch_renorm(x, p[name + "/scale"], p[name + "/bias"], st, train, ns)
            if ns:
                new_stats.update({name + "/" + k: v for k, v in ns.items()})
            return y
        if train:
            axes = tuple(range(x.dim() - 1))
            mean = x.mean(dim=axes)
            var = torch.clamp((x * x).mean(dim=axes) - mean * mean, min=0.0)
            if new_stats is not None:
                new_stats[name + "/mean"] = BN_MOMENTUM * stats[name + "/mean"] + (1.0 - BN_MOMENTUM) * mean.detach()
                new_stats[name + "/var"] = BN_MOMENTUM * stats[name + "/var"] + (1.0 - BN_MOMENTUM) * var.detach()
        else:
            mean, var = stats[name + "/mean"], stats[name + "/var"]
        return (x - mean) * (torch.rsqrt(var + BN_EPS) * p[name + "/scale"]) + p[name + "/bias"]

    def _normalize(self, x, p, name, train, stats, new_stats):
        if self.norm == "layer_norm":
            return F.layer_norm(x, (x.shape[-1],), p[name + "/scale"], p[name + "/bias"], LN_EPS)
        if self.norm == "batch_norm":
            return self._bn(x, p, name, train, stats, new_stats)
        return x

    def apply(self, p: Dict[str, torch.Tensor], x: torch.Tensor, train: bool = False,
              stats: Dict[str, torch.Tensor] = None, new_stats: Dict[str, torch.Tensor] = None) -> torch.Tensor:
        """network.apply({"params": p, "batch_stats": stats}, x, train=train); with train=True the updated
        running statistics are written into `new_stats` (mutable=["batch_stats"], pqn_minatar.py:272-277)."""
        if self.has_batch_stats and stats is None:
            raise ValueError("this network has BatchNorm layers: pass stats=network.init_batch_stats()")
        if self.norm_input:
            x = self._bn(x, p, bn_module(self.renorm) + "_0", train, stats, new_stats)   # (:61-62) -- and no /255 on this branch
        if self.kind == "cnn":
            b = x.shape[0]
            c = x.shape[-1]
            if not self.norm_input:
                x = x / 255.0                                               # (:66)
            n0, n1 = cnn_norm_names(self.norm_type)
            # VALID 3x3 conv in NHWC as patches @ kernel (keeps (h,w,c) order, no layout flips)
            patches = x.unfold(1, 3, 1).unfold(2, 3, 1)            # [B,8,8,C,3,3]
            patches = patches.permute(0, 1, 2, 4, 5, 3).reshape(b, -1, 9 * c)  # [B,64,(ky,kx,c)]
            y = patches @ p["CNN_0/Conv_0/kernel"].reshape(9 * c, 16) + p["CNN_0/Conv_0/bias"]
            y = torch.relu(self._normalize(y, p, n0, train, stats, new_stats))
            y = y.reshape(b, -1)                                    # (h,w,c) flatten
            y = y @ p["CNN_0/Dense_0/kernel"] + p["CNN_0/Dense_0/bias"]
            y = torch.relu(self._normalize(y, p, n1, train, stats, new_stats))
            return y @ p["Dense_0/kernel"] + p["Dense_0/bias"]
        y = x
        for l in range(self.layers):
            y = y @ p[f"Dense_{l}/kernel"] + p[f"Dense_{l}/bias"]
            y = torch.relu(self._normalize(y, p, mlp_norm_name(self.norm_type, l, self.renorm), train, stats, new_stats))
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
