"""Host-side Q-network (torch ops, the plumbing / non-LayerNorm path) vs the numpy oracle, on CPU:
every NORM_TYPE x NORM_INPUT combination of pqn_minatar.py:24-69 / pqn_gymnax.py:29-58 -- parameter
tree, train-mode loss + gradient (torch autograd vs the oracle's hand-written backward), eval-mode
forward on the running statistics, and the batch_stats update."""
import numpy as np
import pytest
import torch

from oracle import pqn_oracle as O
from purejaxql_amd.networks import FlatParams, QNetwork

CASES = [(k, nt, ni) for k in ("cnn", "mlp") for nt in ("layer_norm", "batch_norm", "none") for ni in (False, True)]


@pytest.mark.parametrize("kind,norm_type,norm_input", CASES)
def test_torch_network_matches_oracle(kind, norm_type, norm_input):
    obs_shape, A = ((10, 10, 4), 3) if kind == "cnn" else ((4,), 2)
    gen = torch.Generator().manual_seed(7)
    net = QNetwork(kind, obs_shape, A, norm_type=norm_type, norm_input=norm_input, hidden_size=32, num_layers=2,
                   device="cpu")
    theta = net.init(1)
    theta = theta + 0.05 * torch.randn(theta.shape, generator=gen)      # non-trivial scales / biases
    fp = FlatParams(net, theta.clone())
    B = 24
    x = ((torch.rand((B, *obs_shape), generator=gen) < 0.3).float() if kind == "cnn"
         else torch.randn((B, *obs_shape), generator=gen))
    act = torch.randint(0, A, (B,), generator=gen)
    tgt = torch.randn(B, generator=gen)
    stats = None
    if net.has_batch_stats:
        stats = {k: v + 0.1 * torch.rand(v.shape, generator=gen) for k, v in net.init_batch_stats().items()}
    new_stats = {}
    q = net.apply(fp.leaves, x, train=True, stats=stats, new_stats=new_stats)
    chosen = q.gather(1, act[:, None]).squeeze(1)
    loss = 0.5 * ((chosen - tgt) ** 2).mean()
    loss.backward()

    shapes = (O.cnn_shapes(obs_shape, A, norm_type) if kind == "cnn" else O.mlp_shapes(obs_shape[0], A, 32, 2, norm_type))
    assert list(shapes.items()) == [(k, tuple(s)) for k, s in net.shapes.items()]       # same flax parameter tree
    ostats0 = O.init_batch_stats(kind, obs_shape, 32, 2, norm_type, norm_input)
    assert list(ostats0) == list(net.stats_shapes)
    p = O.unflatten(theta.numpy().copy(), shapes)
    ostats = {k: v.numpy().copy() for k, v in (stats or {}).items()}
    onew = {}
    ol, _oc, og = O.net_loss_grad(kind, p, shapes, x.numpy(), act.numpy(), tgt.numpy(), norm_type == "layer_norm", 2,
                                  norm_type=norm_type, norm_input=norm_input, stats=ostats, new_stats=onew)
    # conv outputs of x/255 inputs are O(1e-3): the fast variance E[x^2]-E[x]^2 of BatchNorm cancels in f32
    tol = 2e-3 if (kind == "cnn" and norm_type == "batch_norm" and not norm_input) else 2e-5
    assert abs(float(ol) - float(loss.detach())) <= tol * max(1.0, abs(float(loss.detach())))
    g = fp.grad.numpy()
    assert np.abs(og - g).max() <= tol * max(1e-9, np.abs(g).max())
    assert sorted(onew) == sorted(new_stats)
    for k in onew:
        np.testing.assert_allclose(onew[k], new_stats[k].numpy(), rtol=max(1e-5, tol), atol=1e-6)
    # eval mode: running averages (rollout / test policy)
    qe = O.net_forward(kind, p, x.numpy(), norm_type == "layer_norm", 2, norm_type=norm_type, norm_input=norm_input,
                       train=False, stats=ostats)
    qt = net.apply(fp.leaves, x, train=False, stats=stats).detach().numpy()
    np.testing.assert_allclose(qe, qt, rtol=1e-4, atol=1e-5)


def test_batchnorm_network_requires_stats():
    net = QNetwork("mlp", (4,), 2, norm_type="batch_norm", device="cpu")
    theta = net.init(0)
    with pytest.raises(ValueError):
        net.apply(net.views(theta), torch.zeros(3, 4))
    assert set(net.init_batch_stats()) == {"BatchNorm_1/mean", "BatchNorm_1/var", "BatchNorm_2/mean", "BatchNorm_2/var"}
    assert "BatchNorm_0/scale" in net.shapes and "BatchNorm_2/scale" in net.shapes and "LayerNorm_0/scale" not in net.shapes


@pytest.mark.parametrize("steps", [0, 999, 1000, 5000])
def test_batch_renorm_torch_vs_oracle(steps):
    """BatchRenorm (purejaxql/utils/batch_renorm.py:19-131, the Craftax script's normaliser): torch-autograd version vs
    the oracle's hand-written forward / backward, before and after the 1000-step warm-up, plus the running update."""
    from purejaxql_amd.networks import batch_renorm
    gen = torch.Generator().manual_seed(3 + steps)
    B, F = 64, 12
    x = (1.5 * torch.randn((B, F), generator=gen) + 0.7).requires_grad_(True)
    scale = (1.0 + 0.2 * torch.randn(F, generator=gen)).requires_grad_(True)
    bias = (0.1 * torch.randn(F, generator=gen)).requires_grad_(True)
    stats = {"mean": 0.5 * torch.randn(F, generator=gen), "var": 0.5 + torch.rand(F, generator=gen), "steps": steps}
    w = torch.randn((B, F), generator=gen)
    new = {}
    y = batch_renorm(x, scale, bias, stats, True, new)
    (y * w).sum().backward()
    ostats = {"mean": stats["mean"].numpy(), "var": stats["var"].numpy(), "steps": steps}
    onew = {}
    oy, cache = O.brn_fwd(x.detach().numpy(), scale.detach().numpy(), bias.detach().numpy(), ostats, True, onew)
    dx, dscale, dbias = O.brn_bwd(w.numpy(), scale.detach().numpy(), cache)
    np.testing.assert_allclose(oy, y.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dx, x.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dscale, scale.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(dbias, bias.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(onew["mean"], new["mean"].numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(onew["var"], new["var"].numpy(), rtol=1e-6, atol=1e-7)
    assert onew["steps"] == steps + 1 == int(new["steps"])
    # r / d really are active after the warm-up (the two regimes differ) and eval mode uses the running moments
    if steps >= 1000:
        y_plain = batch_renorm(x.detach(), scale.detach(), bias.detach(), {**stats, "steps": 0}, True)
        assert (y_plain - y.detach()).abs().max() > 1e-3
    ye = batch_renorm(x.detach(), scale.detach(), bias.detach(), stats, False)
    oye, _ = O.brn_fwd(x.detach().numpy(), scale.detach().numpy(), bias.detach().numpy(), ostats, False)
    np.testing.assert_allclose(oye, ye.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("norm_type,norm_input,steps,q_lambda", [
    ("layer_norm", True, 0, False), ("layer_norm", True, 2000, False),      # pqn_craftax.yaml: BatchRenorm input, LayerNorm hidden
    ("batch_norm", True, 2000, False), ("batch_norm", False, 0, True), ("none", False, 0, False)])
def test_craftax_network_and_loss_branches_match_oracle(norm_type, norm_input, steps, q_lambda):
    """QNetwork of pqn_craftax.py:33-62 (BatchRenorm wherever the gymnax script has BatchNorm) and both branches of its
    _loss_fn (:277-304): Q_LAMBDA=True (targets are inputs) and False (obs || next_obs as ONE train-mode batch, q_next
    without gradient) -- torch autograd vs the oracle's hand-written backward, incl. the batch_stats update."""
    D, A, H, L, B = 11, 5, 32, 3, 16
    gen = torch.Generator().manual_seed(11 + steps)
    net = QNetwork("mlp", (D,), A, norm_type=norm_type, norm_input=norm_input, hidden_size=H, num_layers=L, device="cpu",
                   renorm=True)
    assert "BatchRenorm_0/scale" in net.shapes and "BatchNorm_0/scale" not in net.shapes
    theta = net.init(2)
    theta = theta + 0.05 * torch.randn(theta.shape, generator=gen)
    fp = FlatParams(net, theta.clone())
    x, xn = torch.randn((B, D), generator=gen) * 2 + 1, torch.randn((B, D), generator=gen) * 2 + 1
    act = torch.randint(0, A, (B,), generator=gen)
    rew, tgt = torch.randn(B, generator=gen), torch.randn(B, generator=gen)
    done = torch.rand(B, generator=gen) < 0.3
    stats = None
    if net.has_batch_stats:
        stats = {}
        for k, v in net.init_batch_stats().items():
            stats[k] = torch.tensor(steps, dtype=torch.int32) if k.endswith("/steps") else v + 0.1 * torch.rand(v.shape, generator=gen)
    new_stats = {}
    if q_lambda:
        q = net.apply(fp.leaves, x, train=True, stats=stats, new_stats=new_stats)
        target = tgt
    else:
        q_all = net.apply(fp.leaves, torch.cat((x, xn)), train=True, stats=stats, new_stats=new_stats)
        q, q_next = q_all[:B], q_all[B:].detach()
        target = rew + (1.0 - done.float()) * 0.99 * q_next.max(-1).values
    chosen = q.gather(1, act[:, None]).squeeze(1)
    loss = 0.5 * ((chosen - target) ** 2).mean()
    loss.backward()

    shapes = O.mlp_shapes(D, A, H, L, norm_type, renorm=True)
    assert list(shapes.items()) == [(k, tuple(s)) for k, s in net.shapes.items()]
    assert list(O.init_batch_stats("mlp", (D,), H, L, norm_type, norm_input, renorm=True)) == list(net.stats_shapes)
    p = O.unflatten(theta.numpy().copy(), shapes)
    ostats = {k: (int(v) if k.endswith("/steps") else v.numpy().copy()) for k, v in (stats or {}).items()}
    onew = {}
    kw = dict(norm_type=norm_type, norm_input=norm_input, stats=ostats, new_stats=onew, renorm=True)
    if q_lambda:
        ol, oc, og = O.net_loss_grad("mlp", p, shapes, x.numpy(), act.numpy(), tgt.numpy(), norm_type == "layer_norm", L, **kw)
    else:
        ol, oc, og = O.net_loss_grad_1step("mlp", p, shapes, x.numpy(), xn.numpy(), act.numpy(), rew.numpy(), done.numpy(), 0.99,
                                           norm_type == "layer_norm", L, **kw)
    assert abs(float(ol) - float(loss.detach())) <= 2e-5 * max(1.0, abs(float(loss.detach())))
    np.testing.assert_allclose(oc, chosen.detach().numpy(), rtol=1e-4, atol=1e-5)
    g = fp.grad.numpy()
    assert np.abs(og - g).max() <= 1e-4 * max(1e-9, np.abs(g).max())
    assert sorted(onew) == sorted(new_stats)
    for k in onew:
        np.testing.assert_allclose(np.asarray(onew[k], dtype=np.float64), new_stats[k].numpy().astype(np.float64), rtol=1e-5, atol=1e-6)
    if stats:
        assert int(new_stats[[k for k in new_stats if k.endswith("/steps")][0]]) == steps + 1
