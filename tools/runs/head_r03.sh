mkdir -p gpurun_out
python -m pytest tests/test_gpu_fields.py tests/test_gpu_headline_call.py tests/test_gpu_render.py -q -x > gpurun_out/r03_head_tests.log 2>&1; tail -3 gpurun_out/r03_head_tests.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench_p2.json 2> gpurun_out/r03_bench_p2.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_bench_p2.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["mfma"]["frac"], d["mfma"]["ms_per_step"], d["roofline"]["frac"])
PY
