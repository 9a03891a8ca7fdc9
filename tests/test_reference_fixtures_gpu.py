"""The HIP path against the reference-generated fixtures (see test_reference_fixtures_cpu.py); skips while they are absent."""
import numpy as np
import pytest
import torch

from tests.reference_fixture_maps import breakout_canon, flax_grads, flax_params_to_theta, load, spaceinvaders_canon

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def test_hip_q_lambda_vs_reference(gpu):
    from purejaxql_amd import ops
    rec = load("ref_qlambda.npz")
    got = ops.q_lambda(torch.from_numpy(rec["reward"]).to(gpu), torch.from_numpy(rec["done"].astype(np.uint8)).to(gpu),
                       torch.from_numpy(rec["qmax"]).to(gpu), torch.from_numpy(rec["last_q"]).to(gpu), 0.99, 0.65, quirk=True)
    np.testing.assert_allclose(_np(got), rec["target"], rtol=1e-6, atol=1e-6)


def test_hip_radam_vs_reference(gpu):
    from purejaxql_amd import ops
    rec = load("ref_radam.npz")
    p = torch.from_numpy(rec["p0"].copy()).to(gpu)
    opt = ops.FlatRAdam(p, 5e-4, 10.0, lr_decay_steps=7.0)
    for i in range(rec["grads"].shape[0]):
        opt.step(torch.from_numpy(rec["grads"][i]).to(gpu))
        np.testing.assert_allclose(_np(p), rec["params"][i], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_hip_qnetwork_vs_reference(gpu, oracle, mode):
    from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer, cnn_forward, matmul_mode
    from tests.test_qnet_gpu import _random_bits   # noqa: F401  (packing helper lives there)
    rec = load("ref_qnet.npz")
    shapes = oracle.cnn_shapes((10, 10, 4), 3, "layer_norm")
    theta = torch.from_numpy(flax_params_to_theta(rec, "layer_norm_0", shapes)).to(gpu)
    lay = CnnKernelLayout(4, 3, matmul_f16=matmul_mode(mode))
    obs = rec["obs"]
    flat = obs.reshape(obs.shape[0], -1).astype(np.uint64)
    padded = np.zeros((obs.shape[0], 16 * 32), np.uint64)
    padded[:, :400] = flat
    words = (padded.reshape(-1, 16, 32) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
    bits = torch.from_numpy(words.view(np.int32)).to(gpu)
    q, _, _ = cnn_forward(lay, bits, lay.to_kernel(theta))
    np.testing.assert_allclose(_np(q), rec["layer_norm_0/q_eval"], rtol=1e-4, atol=2e-5)
    tr = CnnTrainer(lay, theta, 5e-4, 10.0)
    n = obs.shape[0]
    g = tr.compute_grad(torch.arange(n, dtype=torch.int64, device=gpu), bits, torch.from_numpy(rec["action"]).to(gpu),
                        torch.from_numpy(rec["target"]).to(gpu))
    gr = flax_grads(rec, "layer_norm_0", shapes)
    np.testing.assert_allclose(_np(lay.to_flax(g)), gr, rtol=2e-3, atol=3e-6 * np.abs(gr).max() + 1e-9)


@pytest.mark.parametrize("name,canon", [("Breakout-MinAtar", breakout_canon), ("SpaceInvaders-MinAtar", spaceinvaders_canon)])
def test_hip_step_env_vs_gymnax(gpu, name, canon):
    """HIP transition rule from the reference's imported state: the stepped state / reward / done of gymnax's step_env
    (steps that end an episode are compared on reward and done only: the HIP kernel auto-resets in the same launch)."""
    from purejaxql_amd.envs import make
    rec = load(f"ref_env_{name}.npz")
    env, params = make(name, device=gpu)
    n_steps, n = rec["actions"].shape
    for t in range(n_steps):
        state = env.import_state(torch.from_numpy(canon(rec, "before", t)))
        _obs, state, r, d, _ = env.step(0, state, torch.from_numpy(rec["actions"][t]).to(gpu), params)
        np.testing.assert_array_equal(_np(r), rec["reward"][t])
        np.testing.assert_array_equal(_np(d), rec["done"][t])
        si, _sf, _log = env.export_state(state)
        keep = ~rec["done"][t].astype(bool)
        np.testing.assert_array_equal(_np(si)[keep], canon(rec, "after", t)[keep], err_msg=f"{name} step {t}")
