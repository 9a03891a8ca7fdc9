"""CPU tests (-m "not gpu") of the host logic: config grammar, C-ABI exports, and
the no-fallback rule.  No kernel is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_hydra_grammar_and_flatten():
    from purejaxql_amd.config_loader import flatten, load_config
    cfg = load_config(["+alg=pqn_minatar", "alg.NUM_ENVS=4096", "alg.ENV_NAME=Breakout-MinAtar", "SEED=3",
                       "alg.TOTAL_TIMESTEPS=2e6"])
    assert cfg["SEED"] == 3 and cfg["alg"]["NUM_ENVS"] == 4096
    flat = flatten(cfg)                                   # {**config, **config["alg"]} (pqn_minatar.py:437)
    assert flat["NUM_ENVS"] == 4096 and flat["ENV_NAME"] == "Breakout-MinAtar" and flat["NUM_SEEDS"] == 1
    assert isinstance(flat["TOTAL_TIMESTEPS_DECAY"], float) and flat["TOTAL_TIMESTEPS_DECAY"] == 1e7   # F9
    assert flat["TOTAL_TIMESTEPS"] == 2e6
    assert flat["LR"] == 0.0005 and flat["LAMBDA"] == 0.65 and flat["NUM_MINIBATCHES"] == 32 and flat["NUM_EPOCHS"] == 2
    cp = flatten(load_config(["+alg=pqn_cartpole"]))
    assert cp["NUM_STEPS"] == 64 and cp["REW_SCALE"] == 0.1 and cp["HIDDEN_SIZE"] == 256 and cp["LAMBDA"] == 0.95
    with pytest.raises(FileNotFoundError):
        load_config(["+alg=nope"])


def test_make_train_divisibility_assert():
    from purejaxql_amd.pqn import derive_config
    with pytest.raises(AssertionError):   # pqn_minatar.py:99-101
        derive_config({"TOTAL_TIMESTEPS": 1e4, "TOTAL_TIMESTEPS_DECAY": 1e4, "NUM_STEPS": 3, "NUM_ENVS": 5,
                       "NUM_MINIBATCHES": 4})


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "pqn_hotpath.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pqn_[a-z0-9_]+)\s*\(", text)))


def test_c_abi_library_exports_every_declared_symbol():
    from purejaxql_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 12
    assert os.path.exists(_lib.LIB_PATH), "build libpqn_hip.so first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == syms            # the ctypes table binds exactly the header
    _lib.load()
    assert _lib.load().pqn_version() >= 1


def test_host_prng_helpers_match_oracle(oracle):
    from purejaxql_amd import _lib
    for key, d in [(0, 0), (0x0123456789ABCDEF, 77), (2**64 - 1, 2**32 - 1)]:
        assert _lib.fold_in(key, d) == oracle.fold_in(key, d)


def test_no_cpu_fallback():
    import torch
    from purejaxql_amd.pqn import make_train
    cfg = {"TOTAL_TIMESTEPS": 1e4, "TOTAL_TIMESTEPS_DECAY": 1e4, "NUM_STEPS": 4, "NUM_ENVS": 4, "NUM_MINIBATCHES": 2,
           "ENV_NAME": "Breakout-MinAtar"}
    with pytest.raises(RuntimeError):
        make_train(cfg, device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            make_train(dict(cfg))  # no GPU -> loud failure, never a silent CPU path


def test_unknown_env_is_an_error():
    from purejaxql_amd import _lib
    lib = _lib.load()
    for i, name in enumerate([b"Breakout-MinAtar", b"CartPole-v1", b"Asterix-MinAtar", b"Freeway-MinAtar",
                              b"SpaceInvaders-MinAtar", b"Craftax-Classic-Symbolic-v1", b"Acrobot-v1"]):
        assert lib.pqn_env_id(name) == i
    assert lib.pqn_env_id(b"Seaquest-MinAtar") < 0      # gymnax 0.0.6 cannot make it either
    assert lib.pqn_env_id(b"Pong-v5") < 0
    assert b"Pong-v5" in lib.pqn_last_error()
    spec = _lib.EnvSpec()
    assert lib.pqn_env_spec(99, ctypes.byref(spec)) < 0
    assert lib.pqn_env_spec(0, ctypes.byref(spec)) == 0
    assert tuple(spec.obs_dim) == (10, 10, 4) and spec.num_actions == 3 and spec.max_steps == 1000
    assert spec.state_words == 7 and spec.obs_words == 16 and spec.canon_si == 109


def test_craftax_entry_point_default_env_name_fails_with_a_pointer_to_the_classic_env():
    """`python -m purejaxql_amd.pqn_craftax` without overrides keeps the reference yaml's ENV_NAME (full Craftax,
    config/alg/pqn_craftax.yaml:23), which this build does not implement: the failure must say what to pass instead."""
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.envs import make
    assert flatten(load_config(["+alg=pqn_craftax"]))["ENV_NAME"] == "Craftax-Symbolic-v1"
    with pytest.raises(ValueError, match="Craftax-Classic-Symbolic-v1"):
        make("Craftax-Symbolic-v1", device="cpu")


def test_run_time_options_and_kernel_form_query():
    """pqn_set_option / pqn_get_option / pqn_cnn_last_kernel_form (host-only calls): unknown names are errors, the
    context manager restores values, nothing has run yet in this process."""
    from purejaxql_amd import _lib
    assert _lib.get_option("t1_pair") == 1 and _lib.get_option("bwd_pos") == 1
    with _lib.options(t1_pair=2, rollout_pair=0):
        assert (_lib.get_option("t1_pair"), _lib.get_option("rollout_pair")) == (2, 0)
    assert (_lib.get_option("t1_pair"), _lib.get_option("rollout_pair")) == (1, 1)
    with pytest.raises(RuntimeError, match="unknown option"):
        _lib.set_option("no_such_switch", 1)
    assert _lib.last_kernel_form() == ("none", "none")


def test_no_mfma_with_partially_overlapping_accumulator_in_any_kernel():
    """Compiler finding 2 of DESIGN.md: a v_mfma whose destination tuple partially overlaps its accumulator input gave
    run-to-run different results on MI355X.  Every MFMA of the bf16x3 / fp16 / wide-MLP paths is therefore issued as tied
    inline asm; tools/check_mfma_overlap.py compiles every csrc/*.hip to gfx950 assembly and scans for the pattern."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_overlap.py")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-2000:]
    assert out.stdout.count(" 0 partially overlapping") >= 7, out.stdout


def test_header_is_plain_c(tmp_path):
    """include/pqn_hotpath.h is the drop-in boundary: it must compile as C99 (no C++-isms, no torch / HIP types) and a
    C translation unit must be able to reference every declared entry point."""
    import re
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "pqn_hotpath.h")
    names = sorted(set(re.findall(r"\b(pqn_[a-z0-9_]+)\s*\(", open(hdr).read())))
    src = tmp_path / "use.c"
    src.write_text('#include "pqn_hotpath.h"\nvoid *table[] = {' + ", ".join(f"(void *){n}" for n in names) + "};\n")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-Wno-pedantic", "-I", os.path.join(root, "include"), "-c", str(src),
                    "-o", str(tmp_path / "use.o")], check=True)


def test_integration_md_ctypes_stub_runs_against_the_library():
    """The ctypes stub printed in INTEGRATION.md is real: its struct definitions match the library (host-only entry
    points are called; the device entry points only get their argtypes)."""
    import ctypes as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = text.split("```python")[1].split("```")[0]
    head = block.split("env = lib.pqn_env_id")[0]                    # definitions + argtypes, no device calls
    head = head.replace('C.CDLL("purejaxql_amd/csrc/libpqn_hip.so")',
                        'C.CDLL(%r)' % os.path.join(root, "purejaxql_amd", "csrc", "libpqn_hip.so"))
    import torch  # noqa: F401  (the library must see torch's HIP runtime first, as the stub's comment says)
    ns = {}
    exec(head, ns)
    lib, EnvSpec = ns["lib"], ns["EnvSpec"]
    env = lib.pqn_env_id(b"Breakout-MinAtar")
    assert env >= 0 and lib.pqn_env_id(b"NoSuchEnv-v0") < 0
    spec = EnvSpec()
    lib.pqn_env_spec.argtypes = [C.c_int, C.POINTER(EnvSpec)]
    assert lib.pqn_env_spec(env, C.byref(spec)) == 0
    assert list(spec.obs_dim) == [10, 10, 4] and spec.obs_size == 400 and spec.num_actions == 3 and spec.max_steps == 1000
    assert spec.obs_words == 16 and spec.state_words == 2 + 5
    lib.pqn_fold_in.restype = C.c_uint64
    lib.pqn_fold_in.argtypes = [C.c_uint64, C.c_uint32]
    from purejaxql_amd import _lib
    assert lib.pqn_fold_in(42, 7) == _lib.fold_in(42, 7)
    assert b"unknown" in lib.pqn_last_error().lower() or lib.pqn_last_error() is not None


def test_c_only_example_builds_and_reaches_the_device_boundary(tmp_path):
    """examples/c_api_demo.c: plain gcc + the header + the shared library.  Without a GPU it must get through the host-only
    entry points (env id / spec / version) and stop at the first device call with an error, not a crash."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    exe = str(tmp_path / "c_api_demo")
    lib_dir = os.path.join(root, "purejaxql_amd", "csrc")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I",
                    os.path.join(root, "include"), os.path.join(root, "examples", "c_api_demo.c"), "-L", lib_dir, "-lpqn_hip",
                    "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
                   check=True)
    import torch
    if torch.cuda.is_available():
        pytest.skip("the device part is exercised by the GPU suite's own callers")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert "obs 10x10x4, 3 actions, 7 state words, 16 packed obs words" in out.stdout
    assert out.returncode == 1 and "hipStreamCreate" in out.stderr


def test_non_default_env_options_are_rejected_loudly():
    """EnvParams.max_steps_in_episode and ENV_KWARGS are compiled into / absent from the kernels: asking for anything
    but the defaults raises instead of being silently ignored (the check runs before any kernel launch)."""
    import pytest
    from purejaxql_amd.envs import EnvParams, Environment, make
    with pytest.raises(ValueError):
        make("Breakout-MinAtar", device="cpu", use_minimal_action_set=False)
    env = Environment.__new__(Environment)
    env.name, env.default_params = "Breakout-MinAtar", EnvParams(max_steps_in_episode=1000)
    env._check_params(None)
    env._check_params(EnvParams(max_steps_in_episode=1000))
    with pytest.raises(ValueError):
        env._check_params(EnvParams(max_steps_in_episode=500))


def test_degenerate_linear_schedule_is_constant_init_value(oracle):
    """optax.linear_schedule with transition_steps <= 0 (EPS_DECAY = 0) is a constant schedule at init_value."""
    from purejaxql_amd.pqn import linear_schedule
    assert linear_schedule(1.0, 0.05, 0.0)(7) == 1.0 and linear_schedule(1.0, 0.05, -3)(0) == 1.0
    assert oracle.linear_schedule(1.0, 0.05, 0.0, 7) == 1.0
    assert abs(linear_schedule(1.0, 0.05, 244.1)(122.05) - 0.525) < 1e-12 and linear_schedule(1.0, 0.05, 244.1)(1e9) == 0.05


def test_runner_state_answers_the_reference_tuple_protocol():
    """pqn_minatar.py:420-424: runner_state = (train_state, (obs, env_state), test_metrics, rng)."""
    from purejaxql_amd.pqn import RunnerState
    rs = RunnerState({"params": {"Dense_0/kernel": 1}, "theta": 2, "env_state": "S", "last_obs": "O", "test_metrics": {"a": 1},
                      "rng": 7, "timesteps": 10, "n_updates": 2, "grad_steps": 128, "opt_count": 3})
    train_state, (obs, env_state), test_metrics, rng = rs
    assert rs[0].params == {"Dense_0/kernel": 1} and train_state.grad_steps == 128 and train_state.n_updates == 2
    assert (obs, env_state) == ("O", "S") and test_metrics == {"a": 1} and rng == 7 and len(rs) == 4
    assert rs["theta"] == 2 and "driver" not in rs and rs[1] == ("O", "S")


def test_update_argument_structs_match_the_header(tmp_path):
    """The ctypes mirrors of the three whole-update argument blocks have the size and the field offsets gcc gives the
    structs of include/pqn_hotpath.h (a silent mismatch would hand the C side shifted pointers)."""
    import ctypes as C
    import subprocess
    from purejaxql_amd import qnet
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pqn_hotpath.h"\n'
                   '#define P(T) printf("%zu %zu %zu %zu\\n", sizeof(T), offsetof(T, layout), offsetof(T, clock), offsetof(T, metrics))\n'
                   'int main(void) { P(pqn_update_args_t); P(pqn_mlp_update_args_t); P(pqn_bigmlp_update_args_t); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    rows = [tuple(int(x) for x in ln.split()) for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n") if ln]
    for row, cls in zip(rows, (qnet.UpdateArgs, qnet.MlpUpdateArgs, qnet.BigMlpUpdateArgs)):
        assert row == (C.sizeof(cls), cls.layout.offset, cls.clock.offset, cls.metrics.offset), (cls.__name__, row)


def test_test_rows_are_spans_of_the_latest_evaluation():
    """pqn._TestRows: metrics["test/<k>"][u] is the result of the latest evaluation at update u (pqn_minatar.py:340-350), kept as
    (first update, values) spans and written once -- for one seed ([5] values) and for a seed batch ([S, 5])."""
    import torch
    from purejaxql_amd.pqn import INFO_KEYS, _TestRows
    k = len(INFO_KEYS)
    tr = _TestRows(10)
    tr.note(0, torch.arange(k, dtype=torch.float32))
    tr.note(4, 10.0 + torch.arange(k, dtype=torch.float32))
    tr.note(9, 20.0 + torch.arange(k, dtype=torch.float32))
    rows = tr.rows()
    assert rows.shape == (10, k)
    for u in range(10):
        base = 0.0 if u < 4 else (10.0 if u < 9 else 20.0)
        assert torch.equal(rows[u], base + torch.arange(k, dtype=torch.float32)), u
    tb = _TestRows(6)
    tb.note(0, torch.zeros((3, k)))
    tb.note(2, torch.ones((3, k)) * torch.tensor([[1.0], [2.0], [3.0]]))
    rb = tb.rows()
    assert rb.shape == (3, 6, k) and float(rb[:, :2].abs().max()) == 0.0
    assert torch.equal(rb[:, 2:, 0], torch.tensor([[1.0] * 4, [2.0] * 4, [3.0] * 4]))
    one = _TestRows(3)
    one.note(0, torch.full((k,), float("nan")))          # an evaluation in which no episode finished
    assert bool(torch.isnan(one.rows()).all())


def test_tune_runs_the_references_sweep_space_offline(capsys):
    """run.tune (pqn_minatar.py:484-531 without the wandb service): every value of the reference's sweep space (LR in {1e-3, 5e-4,
    1e-4, 5e-5}) is run once through the launcher path with the drawn value written over the config -- also inside the nested
    `alg` group, which single_run's flatten would otherwise let win --, checkpoints off; the metric is the mean over seeds of
    the last returned_episode_returns (NaN-safe), the goal maximize.  Driven with a stand-in run function (no GPU here)."""
    import torch
    from purejaxql_amd.config_loader import load_config
    from purejaxql_amd.run import SWEEP_METRIC, SWEEP_PARAMETERS, tune
    assert SWEEP_PARAMETERS == {"LR": [0.001, 0.0005, 0.0001, 0.00005]} and SWEEP_METRIC == "returned_episode_returns"
    seen = []

    def fake_run(config, script="gymnax"):
        from purejaxql_amd.config_loader import flatten
        flat = flatten(config)
        seen.append((flat["LR"], flat.get("SAVE_PATH"), script, config["alg"]["LR"]))
        lr = float(flat["LR"])
        # two seeds, three updates; "returns" peak at 5e-4; the 5e-5 run finishes no episode in its last update (NaN)
        last = float("nan") if lr == 0.00005 else 100.0 - abs(lr - 0.0005) * 1e5
        m = torch.tensor([[1.0, 2.0, last], [1.0, 2.0, last + 2.0]])
        return {"metrics": {SWEEP_METRIC: m}, "rank": 0, "world_size": 1}

    cfg = load_config(["+alg=pqn_minatar", "HYP_TUNE=True", "SAVE_PATH=/tmp/should_not_be_used"])
    res = tune(cfg, script="gymnax", run_fn=fake_run)
    assert [s[0] for s in seen] == SWEEP_PARAMETERS["LR"] and all(s[1] is None and s[0] == s[3] for s in seen)
    assert cfg["alg"]["LR"] == 0.0005 and cfg["SAVE_PATH"] == "/tmp/should_not_be_used"      # the default config is not modified
    assert res["best"]["parameters"] == {"LR": 0.0005} and abs(res["best"][SWEEP_METRIC] - 101.0) < 1e-9
    assert [t["parameters"]["LR"] for t in res["trials"]] == SWEEP_PARAMETERS["LR"]
    import math
    assert math.isnan(res["trials"][3][SWEEP_METRIC])
    out = capsys.readouterr().out
    assert out.count("sweep trial") == 4 and "sweep ranking" in out
    two = tune(cfg, parameters={"LR": [1e-3], "LAMBDA": [0.5, 0.9]}, run_fn=fake_run)
    assert len(two["trials"]) == 2 and {t["parameters"]["LAMBDA"] for t in two["trials"]} == {0.5, 0.9}
