"""GPU parity tests of the wide-MLP kernels (csrc/pqn_bigmlp.hip: tiled bf16x3 MFMA GEMMs + row / column kernels) against
the oracle's numpy network -- the QNetwork of pqn_craftax.py:33-62 with NORM_TYPE = layer_norm, incl. the C5 shape
(1345 -> 4 x 1024 -> 17, BatchRenorm input, 1-step loss on concat(obs, next_obs), pqn_craftax.py:287-304).
Tolerances as for the other fused networks: q 1e-4 / 2e-5, gradients rtol 2e-3 + 1e-5 max|g|."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _setup(gpu, oracle, d, h, layers, a, norm_input, renorm, seed):
    from purejaxql_amd.networks import QNetwork, bn_module
    from purejaxql_amd.qnet import BigMlpKernelLayout, BigMlpTrainer
    torch.manual_seed(seed)
    net = QNetwork("mlp", (d,), a, norm_type="layer_norm", norm_input=norm_input, hidden_size=h, num_layers=layers, device=gpu,
                   renorm=renorm)
    lay = BigMlpKernelLayout(d, h, layers, a, (2 if renorm else 1) if norm_input else 0)
    assert lay.num_flax == net.num_params
    theta = net.init(seed) + 0.03 * torch.randn(net.num_params, device=gpu)
    tr = BigMlpTrainer(lay, theta, 1e-4, 1.0, lr_decay_steps=500.0)
    torch.testing.assert_close(tr.theta_flax(), theta, rtol=0, atol=0)
    shapes = oracle.mlp_shapes(d, a, h, layers, "layer_norm", renorm)
    assert list(shapes) == list(net.shapes)
    p = oracle.unflatten(_np(theta), shapes)
    return net, lay, tr, theta, shapes, p, bn_module(renorm) + "_0"


@pytest.mark.parametrize("m,n,k", [(64, 64, 32), (2048, 1024, 1345), (1345, 1024, 1024), (1024, 1345, 1024), (1024, 17, 1024),
                                   (1000, 1024, 17), (130, 70, 45), (1, 1, 1)])
@pytest.mark.parametrize("ta,tb", [(0, 1), (1, 1), (0, 0), (1, 0)])
@pytest.mark.parametrize("tile,nsplit", [(64, 1), (128, 3)])
def test_bigmlp_gemm_vs_float64_matmul(gpu, m, n, k, ta, tb, tile, nsplit):
    """pqn_bigmlp_gemm (bf16x3 products, f32 accumulate) in all four operand orientations, both tile heights, with and
    without K splits, ragged tiles, odd leading dimensions and unaligned bases (scalar loaders) as well as aligned ones
    (vector loaders), against a float64 matmul: error <= 4e-7 * (|A| |B|) elementwise -- f32 grade."""
    from purejaxql_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=gpu)
    g.manual_seed(m * 7 + n * 3 + k + ta * 2 + tb)
    pad = 4 if (m + n + k) % 2 else 3          # even / odd leading dimensions: vector and scalar loaders
    a_shape, b_shape = ((k, m) if ta else (m, k)), ((k, n) if tb else (n, k))
    lda, ldb = (a_shape[1] + 3) // 4 * 4 + (0 if pad == 4 else 3), (b_shape[1] + 3) // 4 * 4 + (0 if pad == 4 else 3)
    a_buf = torch.randn(a_shape[0] * lda + 1, device=gpu, generator=g)
    b_buf = torch.randn(b_shape[0] * ldb + 1, device=gpu, generator=g)
    off = 1 if pad == 3 else 0                  # unaligned base pointer in the scalar case
    a_v = a_buf[off:off + a_shape[0] * lda].view(a_shape[0], lda)[:, :a_shape[1]]
    b_v = b_buf[off:off + b_shape[0] * ldb].view(b_shape[0], ldb)[:, :b_shape[1]]
    bias = torch.randn(n, device=gpu, generator=g) if nsplit == 1 else None
    c = torch.full((nsplit, m, n + 2), 7.0, device=gpu)
    scratch = torch.empty(int(lib.pqn_bigmlp_gemm_scratch_floats(m, n, k)), device=gpu)
    _lib.check(lib.pqn_bigmlp_gemm(m, n, k, a_v.data_ptr(), lda, ta, b_v.data_ptr(), ldb, tb,
                                   bias.data_ptr() if bias is not None else None, c.data_ptr(), n + 2, nsplit, m * (n + 2), tile,
                                   scratch.data_ptr(), _lib.stream_ptr()), "pqn_bigmlp_gemm")
    a64 = (a_v.T if ta else a_v).double()
    b64 = (b_v if tb else b_v.T).double()
    ref = a64 @ b64 + (bias.double() if bias is not None else 0.0)
    bound = 4e-7 * (a64.abs() @ b64.abs() + (bias.abs().double() if bias is not None else 0.0)) + 1e-30
    kp = -(-k // 32) * 32                        # K padded to the 32-wide MFMA step
    klen = -(-(-(-kp // nsplit)) // 32) * 32     # ceil(ceil(kp / nsplit) / 32) * 32: the split the library takes
    used = -(-kp // klen)
    err = (c[:, :, :n].double().sum(0) - ref).abs()   # the contract: ALL nsplit partials are valid, the caller sums them
    assert bool((err <= bound).all()), float((err / bound).max())
    assert bool((c[:, :, n:] == 7.0).all())      # nothing written beyond the n valid columns
    assert bool((c[used:, :, :n] == 0.0).all())  # partials the K range did not need are zero-filled (round 4), not left unwritten


@pytest.mark.parametrize("d,h,layers,a,n,norm_input,renorm", [
    (1345, 1024, 4, 17, 1024, True, True),      # C5's acting forward
    (1345, 1024, 4, 17, 517, True, True),       # ragged rows (tile guards)
    (37, 256, 2, 5, 130, False, True),          # no input normalisation (dummy BatchRenorm_0)
    (20, 512, 1, 3, 64, True, False),           # nn.BatchNorm input (the gymnax script's module)
])
def test_bigmlp_forward_vs_oracle(gpu, oracle, d, h, layers, a, n, norm_input, renorm):
    net, lay, tr, theta, shapes, p, bn0 = _setup(gpu, oracle, d, h, layers, a, norm_input, renorm, 5)
    rng = np.random.default_rng(n + d)
    obs = (rng.standard_normal((n, d)) * (rng.random(d) * 2.0) + rng.standard_normal(d)).astype(np.float32)
    stats = None
    if norm_input:   # non-trivial running moments
        rm, rv = rng.standard_normal(d).astype(np.float32) * 0.3, (0.5 + rng.random(d)).astype(np.float32)
        tr.in_mean.copy_(torch.from_numpy(rm))
        tr.in_var.copy_(torch.from_numpy(rv))
        stats = {bn0 + "/mean": rm, bn0 + "/var": rv, bn0 + "/steps": 0}
    q_ref = oracle.net_forward("mlp", p, obs, layers=layers, norm_type="layer_norm", norm_input=norm_input, train=False,
                               stats=stats, renorm=renorm)
    q, action, qmax = tr.forward(torch.from_numpy(obs).to(gpu), eps=0.3, key=77)
    np.testing.assert_allclose(_np(q), q_ref, rtol=1e-4, atol=2e-5)
    oa, oq = oracle.eps_greedy(_np(q), 0.3, key=77)
    np.testing.assert_array_equal(_np(action), oa)
    np.testing.assert_array_equal(_np(qmax), oq)
    q2, _a, _m = tr.forward(torch.from_numpy(obs).to(gpu))
    assert torch.equal(q, q2)     # deterministic


@pytest.mark.parametrize("d,h,layers,a,nb,norm_input,renorm,warm", [
    (1345, 1024, 4, 17, 1024, True, True, False),    # C5's optimizer step (pqn_craftax.yaml), BatchRenorm still warming up
    (1345, 1024, 4, 17, 1024, True, True, True),     # the same after 1000 steps: r / d renormalisation active
    (37, 256, 2, 5, 96, False, True, False),
    (20, 512, 1, 3, 200, True, False, False),
    (37, 256, 2, 5, 1300, True, True, False),        # more than 1024 gradient rows: second pass of the loss kernel's workgroups, ragged tiles
])
def test_bigmlp_one_step_loss_grad_vs_oracle(gpu, oracle, d, h, layers, a, nb, norm_input, renorm, warm):
    """value_and_grad of the `Q_LAMBDA: False` branch (pqn_craftax.py:287-304): obs and next_obs as one batch of 2 nb rows
    gathered out of a [T+1][N] record by a permutation, batch statistics over both halves, stop_gradient(q_next)."""
    net, lay, tr, theta, shapes, p, bn0 = _setup(gpu, oracle, d, h, layers, a, norm_input, renorm, 11)
    rng = np.random.default_rng(nb + d + warm)
    n_env, t_len = nb // 2, 2                     # record of T = 2 steps x N envs (+ 1 slot): nb transitions
    rows = (t_len + 1) * n_env
    # columns with a spread of 0.3 .. 1.5 around means of ~0.3, plus constant 0 / 1 columns as a symbolic observation has
    # them.  (Columns whose spread is tiny against their mean are ill-conditioned under flax's fast variance
    # E[x^2] - E[x]^2, and BatchRenorm's mean correction d sqrt(var) / r carries that noise into the output: two f32
    # implementations then legitimately differ at the 1e-3 level.)
    obs_all = (rng.standard_normal((rows, d)) * (0.3 + rng.random(d) * 1.2) + 0.3 * rng.standard_normal(d)).astype(np.float32)
    obs_all[:, :min(6, d // 4)] = 0.0
    obs_all[:, d - min(6, d // 4):] = 1.0
    action = rng.integers(0, a, nb).astype(np.int32)
    reward = rng.standard_normal(nb).astype(np.float32)
    done = (rng.random(nb) < 0.2)
    idx = rng.permutation(nb).astype(np.int64)
    stats = new_stats = None
    if norm_input:
        rm, rv = rng.standard_normal(d).astype(np.float32) * 0.2, (0.6 + rng.random(d)).astype(np.float32)
        steps = 1500 if warm else 3
        tr.in_mean.copy_(torch.from_numpy(rm))
        tr.in_var.copy_(torch.from_numpy(rv))
        tr.in_steps[0] = steps
        stats = {bn0 + "/mean": rm.copy(), bn0 + "/var": rv.copy()}
        if renorm:
            stats[bn0 + "/steps"] = steps
        new_stats = {}
    lo_t, qv_t = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
    g = tr.compute_grad(torch.from_numpy(idx).to(gpu), torch.from_numpy(obs_all).to(gpu), torch.from_numpy(action).to(gpu),
                        reward=torch.from_numpy(reward).to(gpu), done=torch.from_numpy(done).to(gpu), gamma=0.99,
                        next_offset=n_env, loss_out=lo_t, qv_out=qv_t).clone()
    # The reference forward (oracle.net_loss_grad_1step, spelled out so that the backward can take the KERNEL's relu
    # decisions): with ~1e6 relu inputs per layer a handful sit within f32 rounding of zero, the two forwards then
    # disagree on them, and one flipped unit of one sample moves whole gradient columns by O(1/nb) -- far outside any
    # f32 tolerance without either side being wrong.  So: (1) activations agree to 1e-4 / 2e-5 and relu decisions differ
    # only AT the threshold (|h| <= 1e-4); (2) given the same decisions, every gradient entry agrees to the usual tolerance.
    rows = 2 * nb
    xx = np.concatenate((obs_all[idx], obs_all[idx + n_env])).astype(np.float32)
    q_all, cache = oracle.net_forward("mlp", p, xx, layers=layers, want_cache=True, norm_type="layer_norm", norm_input=norm_input,
                                      train=True, stats=stats, new_stats=new_stats, renorm=renorm)
    flips = 0
    for l in range(layers):
        hk, ho = _np(tr.intermediate(rows, nb, "h", l)), cache["hs"][l + 1]
        # two f32 GEMMs over K = d terms of O(1) magnitude each carry ~4e-7 * sum|a b| ~ 1e-5 of rounding
        np.testing.assert_allclose(hk, ho, rtol=1e-4, atol=1e-4, err_msg=f"h_{l}")
        mism = (hk > 0) != (ho > 0)
        flips += int(mism.sum())
        assert not mism.any() or float(np.maximum(hk, ho)[mism].max()) <= 1e-4, l
        cache["hs"][l + 1] = hk.copy()
    assert flips <= 64, flips
    qo, q_next = q_all[:nb], q_all[nb:]
    tgt = (reward[idx] + (np.float32(1) - done[idx].astype(np.float32)) * np.float32(0.99) * q_next.max(-1)).astype(np.float32)
    chosen = qo[np.arange(nb), action[idx]]
    diff = (chosen - tgt).astype(np.float32)
    lo = np.float32(0.5) * np.mean(diff * diff, dtype=np.float32)
    dq = np.zeros_like(q_all)
    dq[np.arange(nb), action[idx]] = diff / np.float32(nb)
    g_ref = oracle._net_backward("mlp", p, shapes, xx, cache, dq, layers, norm_input)
    assert abs(float(lo_t) - lo) <= 1e-4 * max(1.0, abs(lo)), (float(lo_t), lo)
    assert abs(float(qv_t) - chosen.mean()) <= 1e-4
    g_flax = _np(lay.to_flax(g))
    np.testing.assert_allclose(g_flax, g_ref, rtol=2e-3, atol=1e-5 * np.abs(g_ref).max() + 1e-9)
    # padding of the kernel layout never receives gradient
    assert abs(float(g.sum()) - float(g_flax.astype(np.float64).sum())) <= 1e-3 * max(1.0, float(np.abs(g_flax).sum()))
    if norm_input:
        np.testing.assert_allclose(_np(tr.in_mean), new_stats[bn0 + "/mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(_np(tr.in_var), new_stats[bn0 + "/var"], rtol=1e-5, atol=1e-6)
        if renorm:
            assert int(tr.in_steps[0]) == new_stats[bn0 + "/steps"]
    # repeat: bit-identical gradient (fixed summation orders); then clip + RAdam == oracle step on the flax-flat vector
    if norm_input:
        tr.in_mean.copy_(torch.from_numpy(rm)); tr.in_var.copy_(torch.from_numpy(rv)); tr.in_steps[0] = steps
    g2 = tr.compute_grad(torch.from_numpy(idx).to(gpu), torch.from_numpy(obs_all).to(gpu), torch.from_numpy(action).to(gpu),
                         reward=torch.from_numpy(reward).to(gpu), done=torch.from_numpy(done).to(gpu), gamma=0.99,
                         next_offset=n_env)
    assert torch.equal(g, g2)
    th, m, v = _np(theta).copy(), np.zeros(net.num_params, np.float32), np.zeros(net.num_params, np.float32)
    tr.apply()
    gn = oracle.radam_clip_step(th, g_flax, m, v, 0, np.float32(oracle.linear_schedule(1e-4, 1e-20, 500.0, 0)), 1.0)
    assert abs(float(tr.gnorm[0]) - gn) <= 1e-5 * gn
    np.testing.assert_allclose(_np(tr.theta_flax()), th, rtol=1e-5, atol=1e-7)


def test_bigmlp_q_lambda_loss_grad_vs_oracle(gpu, oracle):
    """The Q(lambda) branch of the Craftax script's loss (pqn_craftax.py:280-286): nb rows, given targets."""
    d, h, layers, a, nb = 50, 256, 3, 6, 320
    net, lay, tr, theta, shapes, p, bn0 = _setup(gpu, oracle, d, h, layers, a, True, True, 3)
    rng = np.random.default_rng(9)
    pool = 1000
    obs = rng.standard_normal((pool, d)).astype(np.float32)
    action = rng.integers(0, a, pool).astype(np.int32)
    target = rng.standard_normal(pool).astype(np.float32)
    idx = rng.permutation(pool)[:nb].astype(np.int64)
    stats = {bn0 + "/mean": np.zeros(d, np.float32), bn0 + "/var": np.ones(d, np.float32), bn0 + "/steps": 0}
    new_stats = {}
    lo_t, qv_t = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
    g = tr.compute_grad(torch.from_numpy(idx).to(gpu), torch.from_numpy(obs).to(gpu), torch.from_numpy(action).to(gpu),
                        target=torch.from_numpy(target).to(gpu), loss_out=lo_t, qv_out=qv_t)
    lo, chosen, g_ref = oracle.net_loss_grad("mlp", p, shapes, obs[idx], action[idx], target[idx], layers=layers,
                                             norm_type="layer_norm", norm_input=True, stats=stats, new_stats=new_stats, renorm=True)
    assert abs(float(lo_t) - lo) <= 1e-4 * max(1.0, abs(lo)) and abs(float(qv_t) - chosen.mean()) <= 1e-4
    np.testing.assert_allclose(_np(lay.to_flax(g)), g_ref, rtol=2e-3, atol=1e-5 * np.abs(g_ref).max() + 1e-9)
    np.testing.assert_allclose(_np(tr.in_mean), new_stats[bn0 + "/mean"], rtol=1e-5, atol=1e-6)


def test_bigmlp_backward_side_stream_is_bit_identical_to_one_stream(gpu, oracle):
    """Two launch orders of the backward pass: the default runs the input-gradient chain first and then the
    parameter-gradient side in batched launches (one transpose, one column-sum, the dW GEMMs, one fold); option
    bm_overlap = 1 runs that side layer by layer on a second stream beside the chain.  Same kernel bodies, same buffers, same
    summation orders: the gradient, the loss and the updated input statistics must be bit-identical between the two, over
    repeated calls (a missing event dependency would show as run-to-run differences) and for both branches of the loss."""
    from purejaxql_amd import _lib
    d, h, layers, a, nb = 77, 512, 3, 7, 384
    rng = np.random.default_rng(4)
    pool = 2 * nb
    obs = torch.from_numpy(rng.standard_normal((pool, d)).astype(np.float32)).to(gpu)
    action = torch.from_numpy(rng.integers(0, a, pool).astype(np.int32)).to(gpu)
    target = torch.from_numpy(rng.standard_normal(pool).astype(np.float32)).to(gpu)
    reward = torch.from_numpy(rng.standard_normal(pool).astype(np.float32)).to(gpu)
    done = torch.from_numpy((rng.random(pool) < 0.1).astype(np.uint8)).to(gpu)
    idx = torch.from_numpy(rng.permutation(nb).astype(np.int64)).to(gpu)
    results = {}
    prev = _lib.get_option("bm_overlap")
    try:
        for ov in (1, 0):
            _lib.set_option("bm_overlap", ov)
            net, lay, tr, theta, shapes, p, bn0 = _setup(gpu, oracle, d, h, layers, a, True, True, 3)
            outs = []
            for rep in range(3):
                lo_t, qv_t = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
                g1 = tr.compute_grad(idx, obs, action, reward=reward, done=done, gamma=0.99, next_offset=nb, loss_out=lo_t, qv_out=qv_t).clone()
                g2 = tr.compute_grad(idx, obs, action, target=target, loss_out=lo_t, qv_out=qv_t).clone()
                tr.apply()
                outs.append((g1, g2, lo_t.clone(), tr.theta.clone(), tr.in_mean.clone(), tr.in_var.clone()))
            torch.cuda.synchronize()
            results[ov] = outs
    finally:
        _lib.set_option("bm_overlap", prev)
    for ra, rb in zip(results[1], results[0]):
        for ta, tb in zip(ra, rb):
            assert torch.equal(ta, tb)
    assert float(results[1][0][0].abs().sum()) > 0 and not torch.equal(results[1][0][3], results[1][2][3])


def test_c5_runs_on_the_wide_mlp_kernels(gpu):
    """`+alg=pqn_craftax alg.ENV_NAME=Craftax-Classic-Symbolic-v1` selects the wide-MLP kernels by itself (the whole-loop
    oracle comparison at this shape is tests/test_craftax_gpu.py::test_c5_yaml_shape_loop_vs_oracle_on_craftax_classic)."""
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.pqn import make_train
    cfg = flatten(load_config(["+alg=pqn_craftax", "alg.ENV_NAME=Craftax-Classic-Symbolic-v1"]))
    cfg.update({"TOTAL_TIMESTEPS": 2 * 1024, "TOTAL_TIMESTEPS_DECAY": 2 * 1024})
    train = make_train(cfg, device="cuda:0", script="craftax")
    assert train.backend == "fused_big"
    cfg2 = dict(cfg, NORM_TYPE="batch_norm")      # hidden BatchRenorm layers: the torch-op network
    assert make_train(cfg2, device="cuda:0", script="craftax").backend == "torch"
