#!/usr/bin/env python
"""Reference-fixture generator: the one-command route from "parity unpinned" to pinned.

CANNOT RUN IN THE BUILD IMAGE (no jax / flax / optax / gymnax there, no network).  Run it wherever the reference's
own environment exists (its docker image nvcr.io/nvidia/jax:25.01-py3 + `pip install -e .`, reference
pyproject.toml:27-51), from the root of this repository:

    python tests/golden/make_reference_fixtures.py [--reference /path/to/purejaxql-checkout]

It IMPORTS the reference (never copies it): gymnax==0.0.6 for the env dynamics and `purejaxql/pqn_minatar.py` for the
Q-network module, and writes small .npz fixtures next to this file:

    ref_env_<Name>.npz      env.step_env / env.step traces on fixed action sequences (4 MinAtar games + CartPole-v1):
                            every leaf of the gymnax EnvState before and after each step, obs, reward, done -- the
                            consuming tests IMPORT the "before" state into the oracle / HIP kernels, apply the same
                            action and compare the "after" state (jax PRNG streams cannot be reproduced, so steps
                            whose outcome depends on a draw are flagged and compared on their deterministic fields)
    ref_qnet.npz            QNetwork.apply + value_and_grad(_loss_fn) (pqn_minatar.py:24-69,271-291) on a fixed batch
    ref_radam.npz           5 steps of optax.chain(clip_by_global_norm, radam(linear_schedule)) (:140-147,159-162)
    ref_qlambda.npz         the Q(lambda) scan (:237-260) on the KA1 inputs of SURVEY 8(c) and on a random [32, 64] case
    ref_craftax_qnet.npz    the Craftax script's QNetwork (pqn_craftax.py:33-62: BatchRenorm input, LayerNorm MLP) --
                            value_and_grad of the `Q_LAMBDA: False` loss (:287-304) on concat(obs, next_obs), cold and
                            warm BatchRenorm statistics (utils/batch_renorm.py:95-116), updated batch_stats, eval forward
    ref_optimistic.npz      OptimisticResetVecEnvWrapper(LogWrapper(CartPole-v1), 16, 4).step (utils/craftax_wrappers.py:
                            83-148) over 300 steps: which envs finished, which reset slot each took (its deterministic
                            part: the choice of reset index per done env follows jax.random.choice on the done mask and
                            is compared as "a done env receives ONE of the num_resets fresh states"), obs / reward / done /
                            LogWrapper info
    (ref_env_Acrobot-v1.npz is written by the env loop like the other envs.)

tests/test_reference_fixtures_cpu.py (oracle) and tests/test_reference_fixtures_gpu.py (HIP path) pick the files up and
skip while they are absent.  Commit the .npz files (data: inputs and expected outputs), not anything of the reference.
"""
import argparse
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _flat(tree, prefix=""):
    """{name: np.ndarray} of a pytree / flax struct (field names joined with '/')."""
    import jax
    out = {}
    if hasattr(tree, "__dataclass_fields__"):
        for k in tree.__dataclass_fields__:
            out.update(_flat(getattr(tree, k), f"{prefix}{k}/"))
    elif isinstance(tree, dict):
        for k, v in tree.items():
            out.update(_flat(v, f"{prefix}{k}/"))
    else:
        out[prefix[:-1]] = np.asarray(jax.device_get(tree))
    return out


def env_fixtures(out_dir):
    import gymnax
    import jax
    import jax.numpy as jnp
    assert gymnax.__version__.startswith("0.0.6"), f"the reference pins gymnax==0.0.6, found {gymnax.__version__}"
    n_envs, n_steps = 8, 400
    for name in ("Breakout-MinAtar", "Asterix-MinAtar", "Freeway-MinAtar", "SpaceInvaders-MinAtar", "CartPole-v1", "Acrobot-v1"):
        env, params = gymnax.make(name)                                      # pqn_minatar.py:103
        n_act = env.action_space(params).n
        rng = np.random.default_rng(hash(name) % (2 ** 31))
        actions = rng.integers(0, n_act, size=(n_steps, n_envs)).astype(np.int32)
        keys = jax.random.split(jax.random.PRNGKey(0), n_envs)
        obs, state = jax.vmap(env.reset, in_axes=(0, None))(keys, params)    # :107-109
        rec = {"actions": actions, "max_steps_in_episode": np.int32(params.max_steps_in_episode),
               "obs0": np.asarray(obs)}
        for k, v in _flat(state).items():
            rec[f"state0/{k}"] = v
        before, after, ob, rw, dn, st_obs = [], [], [], [], [], []
        step_env = jax.jit(jax.vmap(env.step_env, in_axes=(0, 0, 0, None)))  # the transition rule alone (no auto-reset)
        step = jax.jit(jax.vmap(env.step, in_axes=(0, 0, 0, None)))          # :110-112 (auto-reset inside)
        for t in range(n_steps):
            ks = jax.random.split(jax.random.PRNGKey(1000 + t), n_envs)
            before.append(_flat(state))
            o_se, s_se, r_se, d_se, _ = step_env(ks, state, jnp.asarray(actions[t]), params)
            after.append(_flat(s_se))
            st_obs.append(np.asarray(o_se))
            o, state, r, d, _info = step(ks, state, jnp.asarray(actions[t]), params)   # continue on the auto-reset trajectory
            ob.append(np.asarray(o)); rw.append(np.asarray(r)); dn.append(np.asarray(d))
        for k in before[0]:
            rec[f"before/{k}"] = np.stack([b[k] for b in before])
            rec[f"after/{k}"] = np.stack([a[k] for a in after])
        rec.update(step_env_obs=np.stack(st_obs), obs=np.stack(ob), reward=np.stack(rw), done=np.stack(dn))
        np.savez_compressed(os.path.join(out_dir, f"ref_env_{name}.npz"), **rec)
        print("wrote", name, {k: v.shape for k, v in rec.items() if k.startswith("before/")})


def qnet_fixtures(ref, out_dir):
    import jax
    import jax.numpy as jnp
    from flax.traverse_util import flatten_dict
    rng = np.random.default_rng(0)
    b, a = 32, 3
    obs = (rng.random((b, 10, 10, 4)) < 0.12).astype(np.float32)
    action = rng.integers(0, a, b).astype(np.int32)
    target = rng.standard_normal(b).astype(np.float32)
    rec = {"obs": obs, "action": action, "target": target}
    for norm_type, norm_input in (("layer_norm", False), ("batch_norm", False), ("layer_norm", True)):
        tag = f"{norm_type}_{int(norm_input)}"
        network = ref.QNetwork(action_dim=a, norm_type=norm_type, norm_input=norm_input)      # :54-69
        variables = network.init(jax.random.PRNGKey(1), jnp.zeros((1, 10, 10, 4)), train=False)   # :156-158
        params, batch_stats = variables["params"], variables.get("batch_stats", {})

        def loss_fn(p):   # the body of _loss_fn, pqn_minatar.py:271-287
            q_vals, updates = network.apply({"params": p, "batch_stats": batch_stats}, jnp.asarray(obs), train=True,
                                            mutable=["batch_stats"])
            chosen = jnp.take_along_axis(q_vals, jnp.expand_dims(jnp.asarray(action), axis=-1), axis=-1).squeeze(axis=-1)
            loss = 0.5 * jnp.square(chosen - jnp.asarray(target)).mean()
            return loss, (updates, chosen, q_vals)

        (loss, (updates, chosen, q_train)), grads = jax.value_and_grad(loss_fn, has_aux=True)(params)
        q_eval = network.apply({"params": params, "batch_stats": batch_stats}, jnp.asarray(obs), train=False)   # :184-191
        for k, v in flatten_dict(params, sep="/").items():
            rec[f"{tag}/params/{k}"] = np.asarray(v)
        for k, v in flatten_dict(grads, sep="/").items():
            rec[f"{tag}/grads/{k}"] = np.asarray(v)
        for k, v in flatten_dict(updates.get("batch_stats", {}), sep="/").items():
            rec[f"{tag}/new_batch_stats/{k}"] = np.asarray(v)
        rec[f"{tag}/loss"], rec[f"{tag}/chosen"] = np.asarray(loss), np.asarray(chosen)
        rec[f"{tag}/q_train"], rec[f"{tag}/q_eval"] = np.asarray(q_train), np.asarray(q_eval)
    np.savez_compressed(os.path.join(out_dir, "ref_qnet.npz"), **rec)
    print("wrote ref_qnet.npz")


def radam_fixtures(out_dir):
    import jax.numpy as jnp
    import optax
    rng = np.random.default_rng(1)
    n, steps = 1000, 5
    p0 = rng.standard_normal(n).astype(np.float32)
    grads = np.stack([(rng.standard_normal(n) * (30.0 if i % 2 else 1e-3)).astype(np.float32) for i in range(steps)])
    lr = optax.linear_schedule(init_value=5e-4, end_value=1e-20, transition_steps=7)            # :140-147
    tx = optax.chain(optax.clip_by_global_norm(10.0), optax.radam(learning_rate=lr))             # :159-162
    params = jnp.asarray(p0)
    state = tx.init(params)
    traj = []
    for i in range(steps):
        updates, state = tx.update(jnp.asarray(grads[i]), state, params)
        params = optax.apply_updates(params, updates)
        traj.append(np.asarray(params))
    eps_sched = optax.linear_schedule(1.0, 0.05, 0.1 * 2441)                                      # :134-138, KA3
    np.savez_compressed(os.path.join(out_dir, "ref_radam.npz"), p0=p0, grads=grads, params=np.stack(traj),
                        lr=np.asarray([float(lr(i)) for i in range(steps)], np.float64),
                        eps=np.asarray([float(eps_sched(c)) for c in (0, 1, 100, 244, 245, 3000)], np.float64),
                        eps_degenerate=np.float64(optax.linear_schedule(1.0, 0.05, 0)(7)))
    print("wrote ref_radam.npz")


def qlambda_fixtures(out_dir):
    import jax
    import jax.numpy as jnp

    def targets(reward, done, q_val, last_q, gamma, lam):   # pqn_minatar.py:237-260, with q_val[t] = max_a already
        def _get_target(carry, tr):
            lambda_returns, next_q = carry
            r, d, q = tr
            target_bootstrap = r + gamma * (1 - d) * next_q
            delta = lambda_returns - next_q
            lambda_returns = target_bootstrap + gamma * lam * delta
            lambda_returns = (1 - d) * lambda_returns + d * r
            return (lambda_returns, q), lambda_returns

        last_q = last_q * (1 - done[-1])
        lambda_returns = reward[-1] + gamma * last_q
        _, tg = jax.lax.scan(_get_target, (lambda_returns, last_q), (reward[:-1], done[:-1], q_val[:-1]), reverse=True)
        return jnp.concatenate((tg, lambda_returns[np.newaxis]))

    rec = {}
    r = jnp.asarray([1.0, 0.0, 2.0, 1.0])[:, None]
    q = jnp.asarray([5.0, 6.0, 7.0, 8.0])[:, None]
    for i, d in enumerate(([0, 0, 0, 0], [0, 1, 0, 1])):
        rec[f"ka1_{i}"] = np.asarray(targets(r, jnp.asarray(d, dtype=bool)[:, None], q, jnp.asarray([9.0]), 0.99, 0.65))
    rng = np.random.default_rng(2)
    rr = ((rng.random((32, 64)) < 0.05) * rng.random((32, 64))).astype(np.float32)
    dd = rng.random((32, 64)) < 0.02
    qq = rng.standard_normal((32, 64)).astype(np.float32)
    ll = rng.standard_normal(64).astype(np.float32)
    rec.update(reward=rr, done=dd, qmax=qq, last_q=ll,
               target=np.asarray(targets(jnp.asarray(rr), jnp.asarray(dd), jnp.asarray(qq), jnp.asarray(ll), 0.99, 0.65)))
    np.savez_compressed(os.path.join(out_dir, "ref_qlambda.npz"), **rec)
    print("wrote ref_qlambda.npz")


def craftax_qnet_fixtures(reference_root, out_dir):
    """pqn_craftax.py's QNetwork + BatchRenorm: the pieces rows a18 / f4 restate (oracle brn_fwd / brn_bwd, net_loss_grad_1step)."""
    import jax
    import jax.numpy as jnp
    from flax.traverse_util import flatten_dict
    spec = importlib.util.spec_from_file_location("ref_pqn_craftax", os.path.join(reference_root, "purejaxql", "pqn_craftax.py"))
    refc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(refc)           # imports craftax as well: fails loudly where it is missing
    rng = np.random.default_rng(3)
    nb, d, a, h, layers, gamma = 64, 40, 5, 64, 2, 0.99
    obs = (rng.standard_normal((nb, d)) * (0.3 + rng.random(d)) + 0.3 * rng.standard_normal(d)).astype(np.float32)
    nxt = (obs + 0.1 * rng.standard_normal((nb, d))).astype(np.float32)
    action = rng.integers(0, a, nb).astype(np.int32)
    reward = rng.standard_normal(nb).astype(np.float32)
    done = (rng.random(nb) < 0.2)
    rec = {"obs": obs, "next_obs": nxt, "action": action, "reward": reward, "done": done, "gamma": np.float32(gamma)}
    network = refc.QNetwork(action_dim=a, hidden_size=h, num_layers=layers, norm_type="layer_norm", norm_input=True)   # :33-62
    variables = network.init(jax.random.PRNGKey(2), jnp.zeros((1, d)), train=False)
    params = variables["params"]
    for tag, steps in (("cold", 0), ("warm", 2000)):
        bs = jax.tree_util.tree_map(lambda x: x, variables["batch_stats"])
        flat_bs = flatten_dict(bs, sep="/")
        # warm: running statistics away from their init and the step counter past the warm-up, so that r / d act (:100-105)
        flat_bs = {k: (jnp.asarray(steps, v.dtype) if k.endswith("steps") else
                       (v + 0.1 * jnp.asarray(rng.standard_normal(v.shape), v.dtype) if k.endswith("mean") else
                        v * jnp.asarray(0.6 + rng.random(v.shape), v.dtype))) if steps else v for k, v in flat_bs.items()}
        from flax.traverse_util import unflatten_dict
        bs = unflatten_dict(flat_bs, sep="/")

        def loss_fn(p):   # the body of _loss_fn with Q_LAMBDA False, pqn_craftax.py:287-304
            all_q, updates = network.apply({"params": p, "batch_stats": bs}, jnp.concatenate((jnp.asarray(obs), jnp.asarray(nxt))),
                                           train=True, mutable=["batch_stats"])
            q_vals, q_next = jnp.split(all_q, 2)
            q_next = jnp.max(jax.lax.stop_gradient(q_next), axis=-1)
            target = jnp.asarray(reward) + (1 - jnp.asarray(done)) * gamma * q_next
            chosen = jnp.take_along_axis(q_vals, jnp.expand_dims(jnp.asarray(action), axis=-1), axis=-1).squeeze(axis=-1)
            return 0.5 * jnp.square(chosen - target).mean(), (updates, chosen, all_q)

        (loss, (updates, chosen, all_q)), grads = jax.value_and_grad(loss_fn, has_aux=True)(params)
        q_eval = network.apply({"params": params, "batch_stats": bs}, jnp.asarray(obs), train=False)
        for k, v in flatten_dict(params, sep="/").items():
            rec[f"{tag}/params/{k}"] = np.asarray(v)
        for k, v in flatten_dict(grads, sep="/").items():
            rec[f"{tag}/grads/{k}"] = np.asarray(v)
        for k, v in flat_bs.items():
            rec[f"{tag}/batch_stats/{k}"] = np.asarray(v)
        for k, v in flatten_dict(updates["batch_stats"], sep="/").items():
            rec[f"{tag}/new_batch_stats/{k}"] = np.asarray(v)
        rec[f"{tag}/loss"], rec[f"{tag}/chosen"], rec[f"{tag}/all_q"], rec[f"{tag}/q_eval"] = (np.asarray(loss), np.asarray(chosen),
                                                                                             np.asarray(all_q), np.asarray(q_eval))
    np.savez_compressed(os.path.join(out_dir, "ref_craftax_qnet.npz"), **rec)
    print("wrote ref_craftax_qnet.npz")


def optimistic_fixtures(reference_root, out_dir):
    """OptimisticResetVecEnvWrapper(LogWrapper(env)) on CartPole-v1 (utils/craftax_wrappers.py:83-148,151-200)."""
    import gymnax
    import jax
    import jax.numpy as jnp
    sys.path.insert(0, reference_root)
    from purejaxql.utils.craftax_wrappers import LogWrapper, OptimisticResetVecEnvWrapper
    n_envs, ratio, n_steps = 16, 4, 300
    env, params = gymnax.make("CartPole-v1")
    wenv = OptimisticResetVecEnvWrapper(LogWrapper(env), num_envs=n_envs, reset_ratio=ratio)
    rng = np.random.default_rng(4)
    actions = rng.integers(0, 2, size=(n_steps, n_envs)).astype(np.int32)
    obs, state = wenv.reset(jax.random.PRNGKey(0), params)
    step = jax.jit(wenv.step)
    rec = {"actions": actions, "obs0": np.asarray(obs), "num_envs": np.int32(n_envs), "reset_ratio": np.int32(ratio)}
    ob, rw, dn, info_rer, info_rel, info_ts = [], [], [], [], [], []
    before, after = [], []
    for t in range(n_steps):
        before.append(_flat(state))
        obs, state, r, d, info = step(jax.random.PRNGKey(5000 + t), state, jnp.asarray(actions[t]), params)
        after.append(_flat(state))
        ob.append(np.asarray(obs)); rw.append(np.asarray(r)); dn.append(np.asarray(d))
        info_rer.append(np.asarray(info["returned_episode_returns"])); info_rel.append(np.asarray(info["returned_episode_lengths"]))
        info_ts.append(np.asarray(info["timestep"]))
    for k in before[0]:
        rec[f"before/{k}"] = np.stack([b[k] for b in before])
        rec[f"after/{k}"] = np.stack([a[k] for a in after])
    rec.update(obs=np.stack(ob), reward=np.stack(rw), done=np.stack(dn), returned_episode_returns=np.stack(info_rer),
               returned_episode_lengths=np.stack(info_rel), timestep=np.stack(info_ts))
    np.savez_compressed(os.path.join(out_dir, "ref_optimistic.npz"), **rec)
    print("wrote ref_optimistic.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("PUREJAXQL_REFERENCE", "/root/reference"))
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    spec = importlib.util.spec_from_file_location("ref_pqn_minatar", os.path.join(args.reference, "purejaxql", "pqn_minatar.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)          # imports jax / flax / optax / gymnax: fails loudly where they are missing
    env_fixtures(args.out)
    qnet_fixtures(ref, args.out)
    radam_fixtures(args.out)
    qlambda_fixtures(args.out)
    for fn in (craftax_qnet_fixtures, optimistic_fixtures):   # need the craftax package besides gymnax: keep the rest if absent
        try:
            fn(args.reference, args.out)
        except ImportError as exc:
            print(f"skipped {fn.__name__}: {exc}")


if __name__ == "__main__":
    main()
