O=gpurun_out/r6g; mkdir -p $O
timeout 2400 python -m pytest tests/test_headline_gpu.py tests/test_qnet_gpu.py tests/test_fullsize_gpu.py tests/test_parity_gpu.py tests/test_replay_fault_gpu.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench$i.json 2>/dev/null; python - <<PY
import json
d = json.loads(open("gpurun_out/r6g/bench$i.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("value %.4g ms/update %.2f bwd %.1f fwd %.1f step %.1f" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r["forward_kernel_us"], r["gather_forward_backward_us"]))
PY
done
