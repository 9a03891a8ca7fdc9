"""single_run(config): the launcher half of the reference scripts (purejaxql/pqn_minatar.py:435-483,
534-545) on top of make_train: seeds = independent runs (jax.vmap(make_train) at :459-461), wall-clock
line (:462), per-seed .safetensors + resolved config yaml (:464-483).  wandb is not available offline:
WANDB_MODE other than "disabled" logs metrics as JSON lines to stdout instead."""
from __future__ import annotations

import json
import os
import time
from typing import Any, Dict, List

import torch
import yaml

from .config_loader import flatten, load_config
from .pqn import make_train, seed_keys, vmap_train
from .save_load import save_params


def single_run(config: Dict[str, Any], device: str = "cuda") -> Dict[str, Any]:
    config = flatten(config)                       # {**config, **config["alg"]} (:437)
    alg_name = config.get("ALG_NAME", "pqn")
    env_name = config["ENV_NAME"]
    if config.get("WANDB_MODE", "disabled") != "disabled":
        def cb(u, m):
            print(json.dumps({k: (float(v) if torch.is_tensor(v) else v) for k, v in m.items()}), flush=True)
        config["_CALLBACK"] = cb
    keys = seed_keys(config["SEED"], config["NUM_SEEDS"])      # split(PRNGKey(SEED), NUM_SEEDS) (:456-459)
    t0 = time.time()
    train = make_train(config, device=device)
    outs = vmap_train(train, keys)
    torch.cuda.synchronize()
    print(f"Took {time.time() - t0} seconds to complete.")      # (:462)
    if config.get("SAVE_PATH", None) is not None:
        save_dir = os.path.join(config["SAVE_PATH"], env_name)
        os.makedirs(save_dir, exist_ok=True)
        clean = {k: v for k, v in config.items() if not k.startswith("_")}
        with open(os.path.join(save_dir, f'{alg_name}_{env_name}_seed{config["SEED"]}_config.yaml'), "w") as f:
            yaml.safe_dump(clean, f)
        for i, rs in enumerate(outs["runner_state"]):
            save_params(rs["params"], os.path.join(
                save_dir, f'{alg_name}_{env_name}_seed{config["SEED"]}_vmap{i}.safetensors'))
    return outs


def main(argv: List[str], default_alg: str) -> Dict[str, Any]:
    """`python -m purejaxql_amd.pqn_minatar +alg=pqn_minatar alg.KEY=V KEY=V` (README.md:170-187)."""
    overrides = list(argv)
    if not any(o.startswith("+alg=") or o.startswith("alg=") for o in overrides):
        overrides.insert(0, f"+alg={default_alg}")
    config = load_config(overrides)
    print("Config:\n", yaml.safe_dump(config))
    if config.get("HYP_TUNE", False):
        raise SystemExit("HYP_TUNE (wandb sweep, pqn_minatar.py:486-531) needs the wandb service: out of scope offline")
    outs = single_run(config)
    m = outs["metrics"]
    last = {k: [round(float(x), 4) for x in v[:, -1]] for k, v in m.items()}
    print("final metrics per seed:", json.dumps(last))
    return outs
