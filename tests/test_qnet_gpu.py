"""GPU parity tests of the fused CNN Q-network kernels against the oracle's numpy network
(and therefore, transitively, torch fp32 autograd -- tests/test_oracle_cpu.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _random_bits(rng, n, c, density):
    obs = (rng.random((n, 10, 10, c)) < density).astype(np.float32)
    flat = obs.reshape(n, -1).astype(np.uint64)
    ow = (((100 * c + 31) // 32) + 3) // 4 * 4
    padded = np.zeros((n, ow * 32), np.uint64)
    padded[:, :100 * c] = flat
    words = (padded.reshape(n, ow, 32) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
    return obs, words


# operand modes of the fc1 products that claim f32-grade results: both are held to the SAME tolerances below
F32_MODES = ["f32", "bf16x3"]


@pytest.mark.parametrize("mode", F32_MODES)
@pytest.mark.parametrize("c,a,n", [(4, 3, 16), (4, 3, 1000), (4, 3, 4096), (6, 4, 100), (7, 3, 50), (10, 6, 33)])
def test_cnn_forward_vs_oracle(gpu, oracle, c, a, n, mode):
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, cnn_forward, matmul_mode
    rng = np.random.default_rng(c * 1000 + n)
    torch.manual_seed(1234)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    lay = CnnKernelLayout(c, a, matmul_f16=matmul_mode(mode))
    assert lay.alloc == lay.total + (3 * 1024 * 128 if mode == "bf16x3" else 0)   # bf16x3: six bf16 planes of the fc1 kernel
    assert lay.num_flax == net.num_params
    theta = net.init(5) + 0.05 * torch.randn(net.num_params, device=gpu)
    theta_k = lay.to_kernel(theta)
    torch.testing.assert_close(lay.to_flax(theta_k), theta, rtol=0, atol=0)
    obs, words = _random_bits(rng, n, c, density=0.15)
    bits = torch.from_numpy(words.view(np.int32)).to(gpu)
    p = oracle.unflatten(_np(theta), oracle.cnn_shapes((10, 10, c), a))
    q_ref = oracle.net_forward("cnn", p, obs)
    q, action, qmax = cnn_forward(lay, bits, theta_k, eps=0.25, key=77)
    # f32 everywhere; summation order differs (sparse conv, MFMA K-permutation): rtol 1e-4 / atol 2e-5
    np.testing.assert_allclose(_np(q), q_ref, rtol=1e-4, atol=2e-5)
    np.testing.assert_array_equal(_np(qmax), _np(q).max(-1))
    oa, _ = oracle.eps_greedy(_np(q), 0.25, key=77)       # same q -> identical draws and argmax
    np.testing.assert_array_equal(_np(action), oa)


def test_cnn_forward_on_real_breakout_observations(gpu, oracle):
    from purejaxql_amd.envs import make
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, cnn_forward
    env, params = make("Breakout-MinAtar", device=gpu)
    net = QNetwork("cnn", (10, 10, 4), 3, device=gpu)
    lay = CnnKernelLayout(4, 3)
    theta = net.init(0)
    (obs, bits), state = env.reset(3, params, 777, want_bits=True)
    for t in range(20):
        a = torch.randint(0, 3, (777,), dtype=torch.int32, device=gpu)
        (obs, bits), state, *_ = env.step(t, state, a, params, want_bits=True)
    q, _, _ = cnn_forward(lay, bits, lay.to_kernel(theta))
    q_torch = net.apply(net.views(theta), obs)
    np.testing.assert_allclose(_np(q), _np(q_torch), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("mode", F32_MODES)
@pytest.mark.parametrize("c,a,nb,pool", [(4, 3, 16, 64), (4, 3, 128, 1000), (4, 3, 4096, 20000), (10, 6, 48, 100), (7, 3, 1024, 1024)])
def test_cnn_grad_vs_oracle(gpu, oracle, c, a, nb, pool, mode):
    """value_and_grad(_loss_fn) through the fused kernels vs the oracle's numpy backward.
    Tolerance: rtol 2e-3 + atol 3e-6*max|g| (f32, different summation orders over up to 4096x64 terms)."""
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer, matmul_mode
    rng = np.random.default_rng(nb + c)
    torch.manual_seed(1234)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    lay = CnnKernelLayout(c, a, matmul_f16=matmul_mode(mode))
    theta = net.init(11) + 0.05 * torch.randn(net.num_params, device=gpu)
    tr = CnnTrainer(lay, theta, 5e-4, 10.0, lr_decay_steps=1000.0)
    obs, words = _random_bits(rng, pool, c, density=0.12)
    bits = torch.from_numpy(words.view(np.int32)).to(gpu)
    action = rng.integers(0, a, pool).astype(np.int32)
    target = rng.standard_normal(pool).astype(np.float32)
    idx = rng.permutation(pool)[:nb] if pool >= nb else rng.integers(0, pool, nb)
    loss_t = torch.zeros(1, device=gpu)
    qv_t = torch.zeros(1, device=gpu)
    g = tr.compute_grad(torch.from_numpy(idx.astype(np.int64)).to(gpu), bits, torch.from_numpy(action).to(gpu),
                        torch.from_numpy(target).to(gpu), loss_t, qv_t)
    shapes = oracle.cnn_shapes((10, 10, c), a)
    p = oracle.unflatten(_np(theta), shapes)
    lo, chosen, g_ref = oracle.net_loss_grad("cnn", p, shapes, obs[idx], action[idx], target[idx])
    assert abs(float(loss_t) - lo) <= 1e-4 * max(1.0, abs(lo))
    assert abs(float(qv_t) - chosen.mean()) <= 1e-4
    g_flax = _np(lay.to_flax(g))
    np.testing.assert_allclose(g_flax, g_ref, rtol=2e-3, atol=3e-6 * np.abs(g_ref).max() + 1e-9)
    pads = torch.ones(lay.total, dtype=torch.bool)
    pads[lay.kidx] = False
    assert float(g[:lay.total][pads.to(gpu)].abs().sum()) == 0.0
    # optimizer half: clip + RAdam in kernel layout == oracle step on the flax-flat vector
    th = _np(theta).copy()
    m = np.zeros_like(th)
    v = np.zeros_like(th)
    for step in range(3):
        if step:
            g_flax = _np(lay.to_flax(tr.compute_grad(torch.from_numpy(idx.astype(np.int64)).to(gpu), bits,
                                                     torch.from_numpy(action).to(gpu), torch.from_numpy(target).to(gpu))))
        tr.apply()
        lr = oracle.linear_schedule(5e-4, 1e-20, 1000.0, step)
        gn = oracle.radam_clip_step(th, g_flax, m, v, step, np.float32(lr), 10.0)
        assert abs(float(tr.gnorm[0]) - gn) <= 1e-5 * gn
        np.testing.assert_allclose(_np(tr.theta_flax()), th, rtol=1e-5, atol=1e-7)
    # the operand copies of the fc1 kernel stay in step with theta
    w1 = _np(tr.theta_flax())[2 * c + 9 * c * 16 + 48:][:1024 * 128].reshape(1024, 128)
    i = np.arange(1024)[:, None]
    o = np.arange(128)[None, :]
    if mode == "f32":     # f32 dgrad-fragment copy
        addr = (((o // 16) * 64 + i // 16) * 64 + ((o % 16) // 4) * 16 + (i % 16)) * 4 + (o % 4)
        np.testing.assert_array_equal(_np(tr.w1b)[addr], w1)
    else:                 # bf16x3: hi + mid + lo == w EXACTLY, in the forward- and the dgrad-order planes
        planes = _np(tr.theta[lay.total:lay.total + 3 * 1024 * 128].view(torch.int16)).view(np.uint16).reshape(6, 1024 * 128)
        to_f32 = lambda b: (b.astype(np.uint32) << 16).view(np.float32)
        jf = ((((i // 32) * 8 + o // 16) * 64 + ((i % 16) // 4) * 16 + o % 16) * 8) + 4 * ((i // 16) % 2) + i % 4
        jd = ((((i // 16) * 4 + o // 32) * 64 + ((o % 16) // 4) * 16 + i % 16) * 8) + 4 * ((o // 16) % 2) + o % 4
        for base, j in ((0, jf), (3, jd)):
            h, m, l = (to_f32(planes[base + k][j]) for k in range(3))
            np.testing.assert_array_equal((h.astype(np.float64) + m + l).astype(np.float32), w1)
            assert np.abs(m).max() <= np.abs(h).max() * 2.0 ** -7 and np.abs(l).max() <= np.abs(h).max() * 2.0 ** -15


@pytest.mark.parametrize("d,h,layers,a,n", [(4, 256, 2, 2, 16), (4, 256, 2, 2, 128), (6, 64, 1, 3, 37), (4, 128, 3, 2, 100)])
def test_mlp_forward_and_grad_vs_oracle(gpu, oracle, d, h, layers, a, n):
    """Fused MLP kernels (pqn_gymnax.py:29-58) vs the oracle's numpy network: forward 1e-4/2e-5,
    gradients rtol 2e-3 + 3e-6*max|g|, post-step parameters 1e-5."""
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import MlpKernelLayout, MlpTrainer, mlp_forward
    rng = np.random.default_rng(d * 100 + n)
    torch.manual_seed(1234)
    net = QNetwork("mlp", (d,), a, hidden_size=h, num_layers=layers, device=gpu)
    lay = MlpKernelLayout(d, h, layers, a)
    assert lay.num_flax == net.num_params
    theta = net.init(2) + 0.05 * torch.randn(net.num_params, device=gpu)
    tr = MlpTrainer(lay, theta, 1e-4, 10.0, lr_decay_steps=500.0, max_minibatch=n)
    torch.testing.assert_close(tr.theta_flax(), theta, rtol=0, atol=0)
    pool = 3 * n
    obs = rng.standard_normal((pool, d)).astype(np.float32)
    obs_t = torch.from_numpy(obs).to(gpu)
    shapes = oracle.mlp_shapes(d, a, h, layers)
    p = oracle.unflatten(_np(theta), shapes)
    q, action, qmax = mlp_forward(lay, obs_t, tr.theta, eps=0.3, key=5)
    q_ref = oracle.net_forward("mlp", p, obs, layers=layers)
    np.testing.assert_allclose(_np(q), q_ref, rtol=1e-4, atol=2e-5)
    oa, oq = oracle.eps_greedy(_np(q), 0.3, key=5)
    np.testing.assert_array_equal(_np(action), oa)
    np.testing.assert_array_equal(_np(qmax), oq)
    act = rng.integers(0, a, pool).astype(np.int32)
    tgt = rng.standard_normal(pool).astype(np.float32)
    idx = rng.permutation(pool)[:n]
    loss_t, qv_t = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
    args = (torch.from_numpy(idx.astype(np.int64)).to(gpu), obs_t, torch.from_numpy(act).to(gpu), torch.from_numpy(tgt).to(gpu))
    g = tr.compute_grad(*args, loss_t, qv_t)
    lo, chosen, g_ref = oracle.net_loss_grad("mlp", p, shapes, obs[idx], act[idx], tgt[idx], layers=layers)
    assert abs(float(loss_t) - lo) <= 1e-4 * max(1.0, abs(lo)) and abs(float(qv_t) - chosen.mean()) <= 1e-4
    np.testing.assert_allclose(_np(lay.to_flax(g)), g_ref, rtol=2e-3, atol=1e-5 * np.abs(g_ref).max() + 1e-9)
    th, m, v = _np(theta).copy(), np.zeros(net.num_params, np.float32), np.zeros(net.num_params, np.float32)
    for step in range(3):
        g_flax = _np(lay.to_flax(tr.compute_grad(*args)))
        tr.apply()
        lr = oracle.linear_schedule(1e-4, 1e-20, 500.0, step)
        oracle.radam_clip_step(th, g_flax, m, v, step, np.float32(lr), 10.0)
        np.testing.assert_allclose(_np(tr.theta_flax()), th, rtol=1e-5, atol=1e-7)
    if layers > 1:   # transposed hidden kernels track theta
        w1 = _np(tr.theta)[lay.struct.off_w[1]:lay.struct.off_w[1] + h * h].reshape(h, h)
        np.testing.assert_array_equal(_np(tr.wt)[:h * h].reshape(h, h), w1.T)


@pytest.mark.parametrize("name,c,a,n,t", [("Breakout-MinAtar", 4, 3, 37, 6), ("Breakout-MinAtar", 4, 3, 256, 40),
                                          ("Asterix-MinAtar", 4, 5, 50, 12), ("Freeway-MinAtar", 7, 3, 33, 8),
                                          ("SpaceInvaders-MinAtar", 6, 4, 20, 10)])
@pytest.mark.parametrize("mode", F32_MODES)
def test_cnn_rollout_equals_step_by_step(gpu, name, c, a, n, t, mode):
    """pqn_cnn_rollout (persistent scan) == T x (pqn_qnet_cnn_forward with eps-greedy, pqn_env_step) with the
    same step keys: actions, rewards, dones, LogWrapper info, packed observations and final env state
    bit-exact; max_a Q to f32 round-off (same kernel code, so in practice identical)."""
    from purejaxql_amd import _lib
    from purejaxql_amd.envs import LogWrapper, make
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, cnn_forward, cnn_rollout, matmul_mode
    lib = _lib.load()
    env, params = make(name, device=gpu)
    env = LogWrapper(env)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    lay = CnnKernelLayout(c, a, matmul_f16=matmul_mode(mode))
    torch.manual_seed(0)
    theta_k = lay.to_kernel(net.init(3) + 0.05 * torch.randn(net.num_params, device=gpu))
    (_o, bits0), state = env.reset(11, params, n, want_obs=False, want_bits=True)
    # play in a little so episodes end inside the window
    for i in range(30):
        (_o, bits0), state, *_ = env.step(500 + i, state, torch.randint(0, a, (n,), dtype=torch.int32, device=gpu), params,
                                          want_obs=False, want_bits=True)
    K = 0x1234567
    keys = torch.empty(t, dtype=torch.int64, device=gpu)
    _lib.check(lib.pqn_fold_in_range(K, 7, t, _lib.ptr(keys), _lib.stream_ptr()), "pqn_fold_in_range")
    assert [int(k) & 0xFFFFFFFFFFFFFFFF for k in keys.tolist()] == [_lib.fold_in(K, 7 + i) for i in range(t)]
    eps = torch.full((1,), 0.3, dtype=torch.float32, device=gpu)

    # fused scan
    words_a = state.words.clone()
    ow = bits0.shape[1]
    bits_a = torch.zeros((t + 1, n, ow), dtype=bits0.dtype, device=gpu)
    bits_a[0] = bits0
    rec = cnn_rollout(lay, env.env_id if hasattr(env, "env_id") else env._env.env_id, words_a, bits_a, theta_k, keys, eps)

    # step by step
    st = state
    bits = bits0
    for i in range(t):
        k = _lib.fold_in(K, 7 + i)
        _q, act, qmax = cnn_forward(lay, bits, theta_k, want_q=False, eps=0.3, key=k)
        (_o, bits), st, r, d, info = env.step(k, st, act, params, want_obs=False, want_bits=True)
        assert torch.equal(rec["action"][i], act), i
        torch.testing.assert_close(rec["qmax"][i], qmax, rtol=1e-6, atol=1e-6)
        assert torch.equal(rec["reward"][i], r) and torch.equal(rec["done"][i].bool(), d.bool()), i
        assert torch.equal(rec["returned_episode_returns"][i], info["returned_episode_returns"])
        assert torch.equal(rec["returned_episode_lengths"][i], info["returned_episode_lengths"].to(torch.int32))
        assert torch.equal(rec["timestep"][i], info["timestep"].to(torch.int32))
        assert torch.equal(rec["discount"][i], info["discount"])
        assert torch.equal(bits_a[i + 1], bits), i
    assert torch.equal(words_a, st.words)
    _q, _a, last = cnn_forward(lay, bits, theta_k, want_q=False)
    torch.testing.assert_close(rec["last_q"], last, rtol=1e-6, atol=1e-6)
    assert rec["done"].sum() > 0 or t < 20     # the longer windows contain episode ends (auto-reset inside the scan)

    # evaluation mode: nothing recorded but the running observation
    words_b = state.words.clone()
    bits_b = bits0.clone().unsqueeze(0)
    rec_b = cnn_rollout(lay, env.env_id if hasattr(env, "env_id") else env._env.env_id, words_b, bits_b, theta_k, keys, eps,
                        store_obs=False, want_last_q=False)
    assert torch.equal(words_b, st.words) and torch.equal(bits_b[0], bits) and torch.equal(rec_b["done"], rec["done"])


@pytest.mark.parametrize("name,c,a,n,t,dtype,waves", [("Breakout-MinAtar", 4, 3, 512, 40, "bf16x3", 0), ("Asterix-MinAtar", 4, 5, 256, 24, "bf16x3", 0),
                                                      ("Freeway-MinAtar", 7, 3, 256, 16, "bf16x3", 0), ("SpaceInvaders-MinAtar", 6, 4, 512, 24, "bf16x3", 0),
                                                      ("Breakout-MinAtar", 4, 3, 512, 40, "f16x2", 0), ("Asterix-MinAtar", 4, 5, 256, 24, "f16x2", 4),
                                                      ("Freeway-MinAtar", 7, 3, 192, 16, "f16x2", 2), ("SpaceInvaders-MinAtar", 6, 4, 384, 24, "f16x2", 4),
                                                      ("Breakout-MinAtar", 4, 3, 320, 24, "f16x2", 0)])
def test_position_structure_rollout_is_consistent_step_by_step(gpu, name, c, a, n, t, dtype, waves):
    """cnn_pos_rollout_kernel (pqn_qnet_pos.hip: one workgroup per 256 envs, the K loop of the training forward kernel per env
    step; option rollout_pos = 2 forces it at these sizes) against the step-by-step pieces of the product: its q values are
    summed in another order than the 16-env kernels', so actions may differ where the two best q values tie to f32 rounding
    -- and nowhere else.  At every step, on the RECORDED observations: max_a Q equals pqn_qnet_cnn_forward's to 1e-5; the
    recorded action equals the eps-greedy action of that kernel (same key, same draws) unless the top-two gap of q is below
    1e-5 max|q|; and pqn_env_step from the recorded state with the RECORDED action reproduces reward, done, LogWrapper info and
    the next packed observation bit for bit, as well as the final env state.  Also the evaluation mode (store_obs = 0) and
    the bootstrap value.  _step_env scan, pqn_minatar.py:181-235.
    dtype f16x2: the kernel on two-piece fp16 operands (the step-by-step reference stays the bf16x3 16-env kernel of the same layout);
    waves = 4 / 2: workgroups of 128 / 64 envs (option pos_waves; 0 = what the launch picks: 8 when the envs come in 256s, else the
    finest cut that divides them -- 320 envs: 2 waves)."""
    from purejaxql_amd import _lib
    from purejaxql_amd.envs import LogWrapper, make
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, cnn_forward, cnn_rollout, matmul_mode
    lib = _lib.load()
    env, params = make(name, device=gpu)
    env = LogWrapper(env)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    lay = CnnKernelLayout(c, a, matmul_f16=matmul_mode(dtype))
    torch.manual_seed(0)
    theta_k = lay.to_kernel(net.init(3) + 0.05 * torch.randn(net.num_params, device=gpu))
    (_o, bits0), state = env.reset(11, params, n, want_obs=False, want_bits=True)
    for i in range(30):
        (_o, bits0), state, *_ = env.step(500 + i, state, torch.randint(0, a, (n,), dtype=torch.int32, device=gpu), params,
                                          want_obs=False, want_bits=True)
    K = 0x7654321
    keys = torch.empty(t, dtype=torch.int64, device=gpu)
    _lib.check(lib.pqn_fold_in_range(K, 3, t, _lib.ptr(keys), _lib.stream_ptr()), "pqn_fold_in_range")
    eps = torch.full((1,), 0.3, dtype=torch.float32, device=gpu)
    env_id = env.env_id if hasattr(env, "env_id") else env._env.env_id
    words_a = state.words.clone()
    bits_a = torch.zeros((t + 1, n, bits0.shape[1]), dtype=bits0.dtype, device=gpu)
    bits_a[0] = bits0
    with _lib.options(rollout_pos=2, pos_waves=waves):
        rec = cnn_rollout(lay, env_id, words_a, bits_a, theta_k, keys, eps)
        assert _lib.last_kernel_form()[1] == "pos"
        rec2 = cnn_rollout(lay, env_id, state.words.clone(), bits_a.clone()[:1].repeat(t + 1, 1, 1).contiguous(), theta_k, keys, eps)
    for k_ in rec:
        assert torch.equal(rec[k_], rec2[k_]), k_          # repeats are bit-identical
    st, ties = state, 0
    for i in range(t):
        k = _lib.fold_in(K, 3 + i)
        q, act_ref, qmax_ref = cnn_forward(lay, bits_a[i].contiguous(), theta_k, eps=0.3, key=k)
        torch.testing.assert_close(rec["qmax"][i], qmax_ref, rtol=1e-5, atol=1e-6)
        top2 = torch.topk(q, 2, dim=1).values
        tie = (top2[:, 0] - top2[:, 1]).abs() <= 1e-5 * q.abs().max()
        diff = rec["action"][i] != act_ref
        assert not bool((diff & ~tie).any()), (i, int((diff & ~tie).sum()))
        ties += int(diff.sum())
        (_o, nbits), st, r, d, info = env.step(k, st, rec["action"][i].contiguous(), params, want_obs=False, want_bits=True)
        assert torch.equal(rec["reward"][i], r) and torch.equal(rec["done"][i].bool(), d.bool()), i
        assert torch.equal(rec["returned_episode_returns"][i], info["returned_episode_returns"])
        assert torch.equal(rec["returned_episode_lengths"][i], info["returned_episode_lengths"].to(torch.int32))
        assert torch.equal(rec["timestep"][i], info["timestep"].to(torch.int32))
        assert torch.equal(rec["discount"][i], info["discount"])
        assert torch.equal(bits_a[i + 1], nbits), i
    assert torch.equal(words_a, st.words) and ties <= max(2, n * t // 2000)
    _q, _a, last = cnn_forward(lay, bits_a[t].contiguous(), theta_k, want_q=False)
    torch.testing.assert_close(rec["last_q"], last, rtol=1e-5, atol=1e-6)
    assert rec["done"].sum() > 0 or name == "Freeway-MinAtar"     # episode ends (auto-reset) inside the window; Freeway's episodes last 2500 steps
    # evaluation mode: nothing recorded but the running observation
    words_b = state.words.clone()
    bits_b = bits0.clone().unsqueeze(0)
    with _lib.options(rollout_pos=2):
        rec_b = cnn_rollout(lay, env_id, words_b, bits_b, theta_k, keys, eps, store_obs=False, want_last_q=False)
    assert torch.equal(words_b, words_a) and torch.equal(bits_b[0], bits_a[t]) and torch.equal(rec_b["done"], rec["done"])


@pytest.mark.parametrize("c,a,nb,pool", [(4, 3, 64, 256), (4, 3, 4096, 20000), (7, 3, 256, 1024)])
def test_cnn_f16_matmul_mode_vs_oracle(gpu, oracle, c, a, nb, pool):
    """MATMUL_DTYPE=f16 (pqn_cnn_layout_t.matmul_f16): fc1 forward / input gradient with fp16 operands and f32
    accumulation, everything else f32.  Against the f32 oracle: q within 5e-3 of max|q| (10-bit mantissa operands
    over a 1024-term dot product), gradient within 2 % relative L2 with cosine > 0.9995; the fp16 copies of the
    fc1 kernel in theta's tail follow the optimizer exactly."""
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer, cnn_forward
    rng = np.random.default_rng(nb + c)
    torch.manual_seed(1234)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    lay = CnnKernelLayout(c, a, matmul_f16=True)
    assert lay.matmul_f16 and lay.alloc == lay.total + 1024 * 128
    theta = net.init(11) + 0.05 * torch.randn(net.num_params, device=gpu)
    tr = CnnTrainer(lay, theta, 5e-4, 10.0, lr_decay_steps=1000.0)
    obs, words = _random_bits(rng, pool, c, density=0.12)
    bits = torch.from_numpy(words.view(np.int32)).to(gpu)
    action = rng.integers(0, a, pool).astype(np.int32)
    target = rng.standard_normal(pool).astype(np.float32)
    idx = rng.permutation(pool)[:nb]
    shapes = oracle.cnn_shapes((10, 10, c), a)
    p = oracle.unflatten(_np(theta), shapes)
    q_ref = oracle.net_forward("cnn", p, obs[idx])
    q, _a, _m = cnn_forward(lay, bits[torch.from_numpy(idx).to(gpu)].contiguous(), tr.theta)
    assert np.abs(_np(q) - q_ref).max() <= 5e-3 * np.abs(q_ref).max()
    loss_t = torch.zeros(1, device=gpu)
    g = tr.compute_grad(torch.from_numpy(idx.astype(np.int64)).to(gpu), bits, torch.from_numpy(action).to(gpu),
                        torch.from_numpy(target).to(gpu), loss_t)
    lo, _chosen, g_ref = oracle.net_loss_grad("cnn", p, shapes, obs[idx], action[idx], target[idx])
    assert abs(float(loss_t) - lo) <= 5e-3 * max(1.0, abs(lo))
    g_flax = _np(lay.to_flax(g)).astype(np.float64)
    rel = np.linalg.norm(g_flax - g_ref) / np.linalg.norm(g_ref)
    cos = float(g_flax @ g_ref / (np.linalg.norm(g_flax) * np.linalg.norm(g_ref)))
    assert rel <= 2e-2 and cos >= 0.9995, (rel, cos)
    # per layer: the f32 parts (conv / LN / head) see only the fp16 noise propagated through fc1
    off = 0
    for k, s in shapes.items():
        n = int(np.prod(s))
        ref, got = g_ref[off:off + n], g_flax[off:off + n]
        if np.linalg.norm(ref) > 0:
            assert np.linalg.norm(got - ref) <= 5e-2 * np.linalg.norm(ref), k
        off += n
    tr.apply()
    tail = tr.theta[lay.total:lay.alloc].view(torch.float16)
    w1k = tr.theta[int(lay.struct.off_w1):int(lay.struct.off_w1) + 1024 * 128]
    assert torch.equal(tail[:1024 * 128], w1k.to(torch.float16))             # forward fragments: same order as the f32 kernel
    assert torch.equal(tail[1024 * 128:].float().sort().values, w1k.to(torch.float16).float().sort().values)   # dgrad copy: a permutation


@pytest.mark.parametrize("c,a,nb,pool", [(4, 3, 128, 1000), (4, 3, 4096, 20000), (6, 4, 1024, 2000), (7, 3, 1024, 1024), (10, 6, 1024, 2000)])
def test_bf16x3_is_deterministic_and_matches_f32_mode(gpu, c, a, nb, pool):
    """Guard for the bf16x3 MFMA issue path (csrc/pqn_qnet.hip, x3_mfma_tied): the training kernels in bf16x3 mode must
    give bit-identical gradients on repeated launches and agree with the f32-MFMA mode of the same kernels to f32
    rounding, for every channel count.  (The builtin form of v_mfma_f32_16x16x32_bf16, and a re-orderable inline-asm
    form, produced run-to-run different forward results in the conv phase for c = 6, 7, 10.)"""
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer
    rng = np.random.default_rng(nb + c)
    torch.manual_seed(1234)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    theta = net.init(11) + 0.05 * torch.randn(net.num_params, device=gpu)
    obs, words = _random_bits(rng, pool, c, density=0.12)
    bits = torch.from_numpy(words.view(np.int32)).to(gpu)
    action = torch.from_numpy(rng.integers(0, a, pool).astype(np.int32)).to(gpu)
    target = torch.from_numpy(rng.standard_normal(pool).astype(np.float32)).to(gpu)
    idx = torch.from_numpy(rng.permutation(pool)[:nb].astype(np.int64)).to(gpu)
    out = {}
    for mode in (0, 2):
        lay = CnnKernelLayout(c, a, matmul_f16=mode)
        tr = CnnTrainer(lay, theta, 5e-4, 10.0, lr_decay_steps=1000.0)
        reps = []
        for _ in range(4):
            lo = torch.zeros(1, device=gpu)
            qv = torch.zeros(1, device=gpu)
            g = tr.compute_grad(idx, bits, action, target, lo, qv)[:lay.total].clone()
            reps.append((float(lo), float(qv), g))
        for r in reps[1:]:
            assert r[0] == reps[0][0] and r[1] == reps[0][1]
            assert torch.equal(r[2], reps[0][2])
        out[mode] = reps[0]
    assert abs(out[0][1] - out[2][1]) <= 2e-6 and abs(out[0][0] - out[2][0]) <= 2e-6 * max(1.0, abs(out[0][0]))
    scale = float(out[0][2].abs().max())
    assert float((out[0][2] - out[2][2]).abs().max()) <= 2e-5 * scale


def _grads_under_options(gpu, c, a, nb, pool, mode, reps=3, **opts):
    """gradient of one fixed minibatch under run-time kernel-selection switches (pqn_set_option); returns
    (gradient, kernel form that ran).  Repeats must be bit-identical."""
    from purejaxql_amd import _lib
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer
    rng = np.random.default_rng(nb + c)
    torch.manual_seed(1234)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    theta = net.init(11) + 0.05 * torch.randn(net.num_params, device=gpu)
    _obs, words = _random_bits(rng, pool, c, density=0.12)
    bits = torch.from_numpy(words.view(np.int32)).to(gpu)
    action = torch.from_numpy(rng.integers(0, a, pool).astype(np.int32)).to(gpu)
    target = torch.from_numpy(rng.standard_normal(pool).astype(np.float32)).to(gpu)
    idx = torch.from_numpy(rng.permutation(pool)[:nb].astype(np.int64)).to(gpu)
    want_lq = opts.pop("want_loss", False)
    with _lib.options(**opts):
        lay = CnnKernelLayout(c, a, matmul_f16=mode)
        tr = CnnTrainer(lay, theta, 5e-4, 10.0, max_minibatch=nb)
        out, lq = [], []
        for _ in range(reps):
            lo, qv = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
            out.append(tr.compute_grad(idx, bits, action, target, lo, qv)[:lay.total].clone())
            lq.append((float(lo), float(qv)))
        form = _lib.last_kernel_form()[0]
    for g, v in zip(out[1:], lq[1:]):
        assert torch.equal(g, out[0]) and v == lq[0], (opts, c, nb)
    if want_lq:
        return out[0], form, lq[0]
    return out[0], form


@pytest.mark.parametrize("c,a,nb,pool", [(4, 3, 4096, 20000), (4, 3, 512, 3000), (4, 5, 1024, 4000), (6, 4, 2048, 6000), (7, 3, 1024, 3000),
                                         (4, 3, 256, 1000), (4, 3, 8192, 20000), (6, 4, 768, 2000)])
@pytest.mark.parametrize("pmode,cut", [(2, None), (3, None), (3, (4, 4)), (3, (2, 8)), (3, (4, 1))], ids=["bf16x3", "f16x2", "f16x2-4w4c", "f16x2-2w8c", "f16x2-4w1c"])
def test_position_parallel_form_of_the_training_step(gpu, oracle, c, a, nb, pool, pmode, cut):
    """bwd_pos=2 routes a minibatch through the position-parallel kernels of pqn_qnet_pos.hip -- minibatch gather +
    bit-transpose, cnn_pos_fwd_kernel (wave = 32 samples, conv on the fly, z in registers, head on the accumulator layout),
    cnn_pos_bwd_kernel (wave = conv position, its dW1 rows in registers, one partial slab per sample chunk) -- and the
    reduction, from ONE forward workgroup (256 samples) over one- and two-chunk backward shapes (512,
    768 / 1024 .. 8192: up to 128 super-tiles per workgroup): the form is reported, repeats are bit-identical (gradient, loss, mean chosen q), loss
    and chosen q equal the f32-MFMA mode of the default kernels, the gradient equals it to f32 rounding and the oracle's numpy
    backward at the tolerance of test_cnn_grad_vs_oracle.  pqn_minatar.py:271-291.
    Both operand modes of the form -- bf16x3 (three bf16 pieces, 6 matrix instructions per product) and f16x2 (two range-scaled fp16
    pieces, 3 per product) -- are held to the SAME bounds; `cut` = (waves per forward workgroup, sample chunks of the backward) forces the
    finer cuts that f16x2 launches of fewer seeds / smaller minibatches take to fill the chip (options pos_waves / pos_chunks; a cut the
    minibatch does not divide into falls back to the launch's own plan)."""
    g_f32, f0, lq0 = _grads_under_options(gpu, c, a, nb, pool, 0, t1_pair=0, t1_ksplit=0, want_loss=True)
    extra = {} if cut is None else {"pos_waves": cut[0], "pos_chunks": cut[1]}
    g_pos, f1, lq1 = _grads_under_options(gpu, c, a, nb, pool, pmode, bwd_pos=2, want_loss=True, **extra)
    assert (f0, f1) == ("single", "pos")
    scale = float(g_f32.abs().max())
    assert abs(lq1[0] - lq0[0]) <= 2e-6 * max(1.0, abs(lq0[0])) and abs(lq1[1] - lq0[1]) <= 2e-6 * max(1.0, abs(lq0[1])), (lq1, lq0)
    assert float((g_f32 - g_pos).abs().max()) <= 2e-5 * scale, float((g_f32 - g_pos).abs().max()) / scale
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout
    rng = np.random.default_rng(nb + c)
    torch.manual_seed(1234)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    theta = net.init(11) + 0.05 * torch.randn(net.num_params, device=gpu)
    obs, _words = _random_bits(rng, pool, c, density=0.12)
    action = rng.integers(0, a, pool).astype(np.int32)
    target = rng.standard_normal(pool).astype(np.float32)
    idx = rng.permutation(pool)[:nb]
    shapes = oracle.cnn_shapes((10, 10, c), a)
    _lo, _chosen, g_ref = oracle.net_loss_grad("cnn", oracle.unflatten(_np(theta), shapes), shapes, obs[idx], action[idx], target[idx])
    lay = CnnKernelLayout(c, a, matmul_f16=2)
    np.testing.assert_allclose(_np(lay.to_flax(g_pos)), g_ref, rtol=2e-3, atol=3e-6 * np.abs(g_ref).max() + 1e-9)


@pytest.mark.parametrize("c,a,nb,pool,spread", [(4, 3, 4096, 20000, 0.0), (4, 3, 4096, 20000, 6.0), (6, 4, 2048, 6000, 0.0)])
def test_operand_modes_of_the_position_form_against_float64(gpu, oracle, c, a, nb, pool, spread):
    """How far is each operand mode from EXACT arithmetic?  One minibatch gradient through the position-parallel kernels in bf16x3
    and f16x2, and through the single-tile kernels in the f32-MFMA mode, against the float64 backward of oracle/pqn_oracle_f64.py
    on the same inputs.  f16x2 carries 22 significand bits per operand where bf16x3 carries 24, but runs half the matrix instructions
    (half the f32 accumulator roundings): both are held to rel-L2 within 4x of the f32 mode's (+1e-7) and under 2e-6 absolutely, the
    worst entry within 1e-5 of max|g|, and f16x2 to no more than 1.5x bf16x3's distance.  spread > 0 multiplies the targets of a few samples by up to e^spread and shrinks others: rows of dz whose
    magnitudes differ by orders of magnitude exercise the per-sample scaling of the f16x2 weight gradient.
    (Numbers: profiles/r06_v7_f16x2_accuracy.txt.)"""
    import pqn_oracle_f64 as o64
    from purejaxql_amd import _lib
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer
    rng = np.random.default_rng(nb + c)
    torch.manual_seed(1234)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    theta = net.init(11) + 0.05 * torch.randn(net.num_params, device=gpu)
    obs, words = _random_bits(rng, pool, c, density=0.12)
    action = rng.integers(0, a, pool).astype(np.int32)
    target = rng.standard_normal(pool).astype(np.float32)
    if spread > 0:
        target = (target * np.exp(rng.uniform(-spread, spread, pool))).astype(np.float32)
    idx = rng.permutation(pool)[:nb]
    shapes = oracle.cnn_shapes((10, 10, c), a)
    _l64, _c64, g64 = o64.cnn_loss_grad(oracle.unflatten(_np(theta).astype(np.float64), shapes), shapes, obs[idx], action[idx], target[idx])
    bits = torch.from_numpy(words.view(np.int32)).to(gpu)
    at, tt, it = torch.from_numpy(action).to(gpu), torch.from_numpy(target).to(gpu), torch.from_numpy(idx.astype(np.int64)).to(gpu)
    err = {}
    for name, mode, opts, form in (("f32", 0, dict(t1_pair=0, t1_ksplit=0), "single"), ("bf16x3", 2, dict(bwd_pos=2), "pos"), ("f16x2", 3, dict(bwd_pos=2), "pos")):
        with _lib.options(**opts):
            lay = CnnKernelLayout(c, a, matmul_f16=mode)
            tr = CnnTrainer(lay, theta, 5e-4, 10.0, max_minibatch=nb)
            lo, qv = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
            g = _np(lay.to_flax(tr.compute_grad(it, bits, at, tt, lo, qv)[:lay.total].clone())).astype(np.float64)
            assert _lib.last_kernel_form()[0] == form
        assert np.isfinite(g).all()
        err[name] = (float(np.linalg.norm(g - g64) / np.linalg.norm(g64)), float(np.abs(g - g64).max() / np.abs(g64).max()))
    print(f"\nC={c} nb={nb} target spread e^+-{spread}: gradient vs float64 (rel-L2, worst entry / max|g|): " +
          ", ".join(f"{k} {v[0]:.2e} {v[1]:.2e}" for k, v in err.items()))
    # measured (profiles/r06_v7_f16x2_accuracy.txt): f32 6.0e-8 / 2.0e-7 / 8.8e-8, bf16x3 2.0e-7 / 4.7e-7 / 2.8e-7, f16x2 1.3e-7 / 3.9e-7 / 1.7e-7
    # -- f16x2 sits BETWEEN the f32 fma chains and bf16x3 (half the matrix instructions = half the accumulator roundings)
    for m in ("bf16x3", "f16x2"):
        assert err[m][0] <= 4 * err["f32"][0] + 1e-7 and err[m][0] < 2e-6 and err[m][1] < 1e-5, err
    assert err["f16x2"][0] <= 1.5 * err["bf16x3"][0], err


@pytest.mark.parametrize("case", ["conv_kernel_x1e3", "conv_kernel_x1e-4", "ln0_scale_x60", "ln0_all_zero", "fc1_kernel_x40", "targets_x1e6", "targets_x1e-9",
                                  "one_sample_dominates", "zero_td_error"])
def test_f16x2_range_scaling_holds_at_the_extremes(gpu, case):
    """fp16 has five exponent bits: the f16x2 mode is only as good as its power-of-two scales (pqn_qnet_x3.h).  Parameters and targets
    pushed orders of magnitude away from an initialised network -- conv kernel x 1e3 / 1e-4, LayerNorm_0 scale x 60 or scale = bias = 0,
    fc1 kernel x 40 (|w| up to ~8: the static 2^7 plane scale holds to 511), TD errors of 1e6 and 1e-9, one sample 1e8 times the rest, targets
    equal to the network's own output (dz = 0 rows) -- must give a finite gradient that agrees with bf16x3 (8-bit exponents, no scaling
    anywhere) as closely as in the benign case."""
    from purejaxql_amd import _lib
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer, cnn_forward
    c, a, nb, pool = 4, 3, 1024, 4000
    rng = np.random.default_rng(77)
    torch.manual_seed(5)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    theta = net.init(3) + 0.05 * torch.randn(net.num_params, device=gpu)
    lay0 = CnnKernelLayout(c, a, matmul_f16=0)
    tk = lay0.to_kernel(theta)[:lay0.total].clone()
    s0 = lay0.struct
    if case == "conv_kernel_x1e3": tk[s0.off_wc:s0.off_wc + 9 * c * 16] *= 1e3
    if case == "conv_kernel_x1e-4": tk[s0.off_wc:s0.off_wc + 9 * c * 16] *= 1e-4
    if case == "ln0_scale_x60": tk[s0.off_ln0s:s0.off_ln0s + 16] *= 60.0
    if case == "ln0_all_zero": tk[s0.off_ln0s:s0.off_ln0s + 32] = 0.0
    if case == "fc1_kernel_x40": tk[s0.off_w1:s0.off_w1 + 1024 * 128] *= 40.0
    theta = lay0.to_flax(tk)
    _obs, words = _random_bits(rng, pool, c, density=0.12)
    bits = torch.from_numpy(words.view(np.int32)).to(gpu)
    action = torch.from_numpy(rng.integers(0, a, pool).astype(np.int32)).to(gpu)
    target = rng.standard_normal(pool).astype(np.float32)
    if case == "targets_x1e6": target *= 1e6
    if case == "targets_x1e-9": target *= 1e-9
    if case == "one_sample_dominates": target[rng.integers(0, pool, 40)] *= 1e8
    target = torch.from_numpy(target).to(gpu)
    if case == "zero_td_error":
        q, _, _ = cnn_forward(CnnKernelLayout(c, a, matmul_f16=2), bits, CnnKernelLayout(c, a, matmul_f16=2).to_kernel(theta))
        target = q.gather(1, action.long().view(-1, 1)).view(-1).clone()
        target[::7] += 1e-3      # most rows of dz exactly (or nearly) zero
    idx = torch.from_numpy(rng.permutation(pool)[:nb].astype(np.int64)).to(gpu)
    g = {}
    with _lib.options(bwd_pos=2):
        for mode in (2, 3):
            lay = CnnKernelLayout(c, a, matmul_f16=mode)
            tr = CnnTrainer(lay, theta, 5e-4, 10.0, max_minibatch=nb)
            lo, qv = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
            g[mode] = tr.compute_grad(idx, bits, action, target, lo, qv)[:lay.total].double().clone()
            assert _lib.last_kernel_form()[0] == "pos" and bool(torch.isfinite(g[mode]).all()) and bool(torch.isfinite(lo).all()), (case, mode)
    if case == "ln0_all_zero":     # h1 = 0 everywhere: the fc1 kernel's gradient is exactly zero in both modes
        w1 = slice(s0.off_w1, s0.off_w1 + 1024 * 128)
        assert float(g[2][w1].abs().max()) == 0.0 and float(g[3][w1].abs().max()) == 0.0
    num, den = float((g[3] - g[2]).norm()), float(g[2].norm())
    print(f"\n{case}: |g_f16x2 - g_bf16x3| / |g_bf16x3| = {num / max(den, 1e-300):.2e}  (|g| = {den:.3e})")
    # zero_td_error: six of seven TD errors ARE the two forwards' rounding difference (q - q' ~ 1e-7 against 1e-3 in the others), so the
    # modes' gradients legitimately differ by that share; the case is there for the all-but-zero dz rows (finite, no 0 * inf)
    assert num <= (2e-3 if case == "zero_td_error" else 2e-6) * den + 1e-30, (case, num, den)


@pytest.mark.parametrize("c,a,nb,pool", [(4, 3, 128, 1000), (4, 3, 16, 64), (6, 4, 256, 600), (7, 3, 96, 300), (10, 6, 48, 100)])
def test_ksplit_form_of_the_training_kernel_for_small_minibatches(gpu, oracle, c, a, nb, pool):
    """f32 operand mode, minibatches of at most 256 samples (the yaml-default MinAtar run has 128): the K-split kernels
    (a tile's work cut along the conv positions over 16 workgroups, fc1 weight gradient without the h1 hand-over) are what
    runs (`pqn_cnn_last_kernel_form`), repeats are bit-identical, and the gradient agrees (a) with the oracle's numpy
    backward at the tolerance of test_cnn_grad_vs_oracle and (b) with the single-tile kernel of the same library
    (option t1_ksplit = 0) to f32 summation-order noise."""
    g_ks, form = _grads_under_options(gpu, c, a, nb, pool, 0, t1_ksplit=1)
    assert form == "ksplit"
    for other in (2, 3, 4):   # the three-launch forms with 4 / 8 / 16 positions per workgroup: same arithmetic per position, other fold orders
        g_o, form_o = _grads_under_options(gpu, c, a, nb, pool, 0, t1_ksplit=other)
        assert form_o == "ksplit" and float((g_o - g_ks).abs().max()) <= 3e-6 * float(g_ks.abs().max())
    g_single, form1 = _grads_under_options(gpu, c, a, nb, pool, 0, t1_ksplit=0)
    assert form1 == "single"
    scale = float(g_single.abs().max())
    assert float((g_ks - g_single).abs().max()) <= 3e-6 * scale, float((g_ks - g_single).abs().max()) / scale
    # the oracle on the same inputs (regenerated exactly as _grads_under_options draws them)
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout
    rng = np.random.default_rng(nb + c)
    torch.manual_seed(1234)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    theta = net.init(11) + 0.05 * torch.randn(net.num_params, device=gpu)
    obs, _words = _random_bits(rng, pool, c, density=0.12)
    action = rng.integers(0, a, pool).astype(np.int32)
    target = rng.standard_normal(pool).astype(np.float32)
    idx = rng.permutation(pool)[:nb]
    shapes = oracle.cnn_shapes((10, 10, c), a)
    _lo, _chosen, g_ref = oracle.net_loss_grad("cnn", oracle.unflatten(_np(theta), shapes), shapes, obs[idx], action[idx], target[idx])
    lay = CnnKernelLayout(c, a, matmul_f16=0)
    np.testing.assert_allclose(_np(lay.to_flax(g_ks)), g_ref, rtol=2e-3, atol=3e-6 * np.abs(g_ref).max() + 1e-9)
