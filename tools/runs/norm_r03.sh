mkdir -p gpurun_out
python -m pytest tests/test_gpu_spec_search.py tests/test_gpu_fields.py tests/test_gpu_render.py -q -x > gpurun_out/r03_norm_tests.log 2>&1; tail -3 gpurun_out/r03_norm_tests.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --aten-profile gpurun_out/r03_aten_ops.txt > gpurun_out/r03_bench_n.json 2> gpurun_out/r03_bench_n.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_bench_n.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["mfma"]["frac"], d["mfma"]["ms_per_step"], d["roofline"]["frac"], d["abi_kernel_ms_per_step"])
PY
