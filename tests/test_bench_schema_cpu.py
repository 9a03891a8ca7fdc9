"""The bench line contract (driver-facing): the committed round artefact profiles/r06_bench.json -- the JSON line
bench.py printed on an MI355X at the end of round 6 -- carries every field the contract names, with consistent
arithmetic."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].split(",")[0] == base["metric"].split(",")[0] and "4096 envs" in d["metric"]
    # what changed against the plain reading of the metric is IN the label: seeds batched per GPU and the operand mode
    assert "16 seeds/GPU" in d["metric"] and "f16x2" in d["metric"]
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    # the arithmetic type is f32 throughout; the label says how the f32 products are evaluated
    assert d["dtype"].startswith("f32") and "f16x2" in d["dtype"] and "22 significand bits" in d["dtype"] and "f32 accumulate" in d["dtype"]
    # round 6: the operand mode of round 5 (three bf16 pieces) is timed in the same run and printed at the top level beside the headline
    assert d["value_bf16x3_operands"] == d["matmul_modes"]["bf16x3"]["value"] and 7.0e7 < d["value_bf16x3_operands"] < d["value"]
    assert "profiles/r06_v7_f16x2_accuracy.txt" in d["operand_mode_note"]
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    c = d["config"]
    assert c["seeds_per_gpu"] == 16 and c["env_steps_per_step"] == 16 * 4096 * 32 and c["matmul_dtype"] == "f16x2"
    # value = env-steps of K updates / time
    assert abs(d["value"] - c["env_steps_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12) <= 1e-6 * r["achieved"]
    # round 5: the dominant kernel is the backward of the position-parallel form (fc1 dgrad + fc1 wgrad + conv wgrad)
    assert "cnn_pos_bwd_kernel" in r["kernel"] and r["flop_per_launch"] == (2 * 262144 + 73728) * 4096 * 16
    # priced against the pipe it runs on: dense fp16 peak / 3 products per f32 product (round 5, bf16x3: / 6); the f32-MFMA-peak basis
    # of rounds 1-4 (on which the kernel is past 1) is kept beside it
    assert abs(r["peak"] - 2500.0 / 3.0) < 1e-9 and 0.25 < r["frac"] < 1.0 and r["frac_f32_mfma_peak"] > 1.4
    assert abs(r["frac_f32_mfma_peak"] - r["achieved"] / 157.3) < 1e-9 and abs(r["fp16_pipe"]["frac"] - r["frac"]) < 1e-9
    ts = r["training_step"]
    # round 6: the gather runs once per epoch; its 1/32 share per optimizer step is part of the step time, and the step's numbers are
    # also flat scalars of `roofline`
    assert ts["forward_kernel_us"] + r["avg_launch_us"] < ts["gather_forward_backward_us"] + 3.0 and ts["gather_forward_backward_us"] < 450.0
    assert r["forward_kernel_us"] == ts["forward_kernel_us"] and r["gather_forward_backward_us"] == ts["gather_forward_backward_us"]
    assert 150.0 < r["epoch_gather_us"] < 260.0 and 0.25 < r["value_and_grad_frac"] < 1.0
    assert r["avg_launch_us"] <= 170.0 and r["forward_kernel_us"] <= 95.0          # backward <= 170 us, forward <= 95 us
    assert r["traffic"] is None or (r["traffic"] > 0 and "from file" in r["traffic_source"] and "NOT measured in this run" in r["traffic_source"])
    # the headline's traffic comes from PMC passes of the 16-seed launch shape itself, not from a scaled single-seed pass
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_pos_bwd_kernel_f16x2_seeds16.json")))
    assert pmc["matmul"] == "f16x2"
    assert "r06_pmc" in r["traffic_source"] and pmc["date"] in r["traffic_source"] and abs(r["l2_to_cu_bytes"] - pmc["l2_to_cu_bytes_per_launch"]) <= 0.02 * r["l2_to_cu_bytes"]
    assert pmc["seeds_per_launch"] == 16 and "seeds16" in r["traffic_source"]
    assert abs(r["traffic"] - pmc["hbm_bytes_per_launch"]) <= 0.02 * r["traffic"]     # the file was re-measured in the same call, after the line
    assert r["traffic"] < 150e6                                                       # T1 WRITE_SIZE <= 150 MB (was 446)
    assert abs(pmc["hbm_bytes_per_launch"] - (2 * pmc["FETCH_SIZE_KB_avg"] + pmc["WRITE_SIZE_KB_avg"]) * 1024.0) < 1.0
    assert 0.0 < r["fp16_pipe"]["frac"] < 1.0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert "NUM_ENVS=4096" in cb["sample"]                      # the CPU leg runs the bench shape itself
    # round 4: >= 3 timed whole updates, thread count calibrated (more torch threads are slower) and reported as `cores`
    assert cb["updates_timed"] >= 3 and str(cb["cores"]) in cb["seconds_per_update_by_threads"] and cb["host_logical_cores"] >= cb["cores"]
    assert cb["seconds_per_update_by_threads"][str(cb["cores"])] == min(cb["seconds_per_update_by_threads"].values())
    # round 6: every native thread pool of the process is limited to the calibrated count (round 5: openblas at 64 beside torch's 16)
    assert cb["cores_used_of_present"] == f"{cb['cores']} of {cb['host_logical_cores']}"
    pools = dict(x.split(":") for x in cb["thread_pools"].replace(" ", "").split(","))
    assert all(int(v) <= cb["cores"] for v in pools.values()), pools
    # which kernels ran is asked of the library, the timed region is backed by a longer one, the env-step roofline says
    # which level of the memory system it measures
    assert d["config"]["kernel_forms"] == {"train": "pos", "rollout": "pos"} and d["config"]["driver"] == "hipGraph replay"
    assert d["value"] >= 1.0e8 and d["ms_per_step"] <= 20.97                       # the north star's line: 1e8 env-steps/s on one MI355X
    assert d["sustained"]["seconds"] >= 5.0 and abs(d["sustained"]["value"] / d["value"] - 1.0) < 0.1
    levels = [e["level"].split(" ")[0] for e in d["roofline_env_step"]]
    assert levels == ["Infinity", "Infinity", "HBM"] and d["roofline_env_step"][-1]["frac"] < 0.9
    # BASELINE.json configs[4] beside the headline: Craftax-Classic at the yaml shape through the wide-MLP kernels
    c5 = d["craftax_c5"]
    assert c5["backend"] == "fused_big" and c5["driver"] == "hipGraph replay" and c5["value"] > 5e5
    assert abs(c5["roofline"]["frac"] - c5["roofline"]["achieved"] / c5["roofline"]["peak"]) < 1e-9
    # round 6: on the basis of its own operand mode (bf16x3: dense bf16 peak / 6), the f32-MFMA-peak fraction kept beside it
    assert abs(c5["roofline"]["peak"] - 2500.0 / 6.0) < 1e-9 and c5["roofline"]["frac"] < c5["roofline"]["frac_f32_mfma_peak"] < 1.0
    # the reference's yaml defaults (128 envs): the small-minibatch regime runs the K-split training kernels
    yd = d["yaml_default"]
    assert yd["kernel_forms"]["train"] == "ksplit" and yd["value"] > 1.2e6 and yd["seconds_for_1e7_steps"] < 8.0
    # BASELINE.json configs[2] / configs[1] beside the headline: the suite at 4096 envs on the position-parallel kernels; Breakout at
    # 1024 envs x 16 seeds (64 workgroups of 256 samples per launch: below round 5's threshold, the pair kernels then) takes the finer cut of
    # round 6 -- 2-wave forward / rollout workgroups -- and stays in the form
    suite = {(g["env"], g["num_envs"]): g for g in d["minatar_suite"] if g["seeds_per_gpu"] == 16}
    assert set(suite) == {("Asterix-MinAtar", 4096), ("Freeway-MinAtar", 4096), ("SpaceInvaders-MinAtar", 4096),
                          ("Breakout-MinAtar", 4096), ("Breakout-MinAtar", 1024)}
    for g in suite.values():
        assert g["kernel_forms"] == {"train": "pos", "rollout": "pos"} and g["seeds_per_gpu"] == 16 and g["value"] > (9e7 if g["num_envs"] == 4096 else 4.5e7), g
        assert g["t1_flop_per_sample"] == 18432 * g["channels"] + 524288 and 0.2 < g["t1_frac"] < 0.6 and g["t1_frac_f32_peak"] > 1.0
    assert d["config"]["seed_groups"] == 1
    # the extras report the other operand modes and the single-seed run beside the headline, never instead of it
    # (the fp16-operand mode has no position-parallel form: since round 5 it is slower than the split-operand modes)
    assert d["matmul_modes"]["f32"]["value"] < d["matmul_modes"]["f16"]["value"] < d["matmul_modes"]["bf16x3"]["value"] < d["value"]
    assert d["single_seed"]["seeds_per_gpu"] == 1 and d["single_seed"]["value"] < d["value"]
