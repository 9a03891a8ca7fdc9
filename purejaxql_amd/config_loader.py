"""Hydra-grammar config loader for the PQN scripts (host side, no hydra dep).

Reproduces what `@hydra.main(config_path="./config", config_name="config")`
plus `single_run`'s flatten do in the reference, from ONE table file (config/hyperparameters.yaml)
(purejaxql/pqn_minatar.py:437,534-541; README.md:170-187):

    load_config(["+alg=pqn_minatar", "alg.NUM_ENVS=4096", "SEED=3"])
      -> {"NUM_SEEDS":1, "SEED":3, ..., "alg": {...}}
    flatten(config) == {**config, **config["alg"]}

PyYAML parses `1e7` as a *string* (OmegaConf parses it as a float),
so numeric-looking strings are cast here.
"""
from __future__ import annotations

import copy
import os
import re
from typing import Any, Dict, Iterable

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")

_FLOAT_RE = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?$")
_INT_RE = re.compile(r"^[+-]?\d+$")


def _cast_scalar(v: Any) -> Any:
    if isinstance(v, str):
        s = v.strip()
        if _INT_RE.match(s):
            return int(s)
        if _FLOAT_RE.match(s):
            return float(s)
    return v


def _cast_tree(x: Any) -> Any:
    if isinstance(x, dict):
        return {k: _cast_tree(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_cast_tree(v) for v in x]
    return _cast_scalar(x)


def _parse_value(s: str) -> Any:
    return _cast_tree(yaml.safe_load(s))


def load_yaml(path: str) -> Dict[str, Any]:
    with open(path) as f:
        return _cast_tree(yaml.safe_load(f) or {})


def _tables(config_dir: str) -> Dict[str, Any]:
    return load_yaml(os.path.join(config_dir, "hyperparameters.yaml"))


def load_config(overrides: Iterable[str] = (), config_dir: str = CONFIG_DIR) -> Dict[str, Any]:
    """Compose the base options + the `+alg=<name>` group + `alg.KEY=V` / `KEY=V` overrides."""
    tables = _tables(config_dir)
    cfg = copy.deepcopy(tables["base"])
    cfg.setdefault("alg", {})
    overrides = list(overrides)
    for ov in overrides:  # group selection first, as hydra does
        if ov.startswith("+alg=") or ov.startswith("alg="):
            name = ov.split("=", 1)[1]
            if name not in tables["alg"]:
                raise FileNotFoundError(f"no alg group '{name}' (have: {sorted(tables['alg'])})")
            cfg["alg"] = {**cfg["alg"], **copy.deepcopy(tables["alg"][name])}
    for ov in overrides:
        if ov.startswith("+alg=") or ov.startswith("alg="):
            continue
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not KEY=VALUE")
        k, v = ov.split("=", 1)
        k = k.lstrip("+")
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _parse_value(v)
    return cfg


def flatten(config: Dict[str, Any]) -> Dict[str, Any]:
    """`config = {**config, **config["alg"]}` (pqn_minatar.py:437)."""
    config = copy.deepcopy(config)
    return {**config, **config.get("alg", {})}
