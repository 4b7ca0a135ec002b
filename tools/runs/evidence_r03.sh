# the round's evidence run on one MI355X box (gpurun): full GPU test suite, both headline benches, micro-benchmarks, rocprofv3 kernel stats and
# PMC traffic of the headline step; outputs under gpurun_out/, copied to profiles/ by hand
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_tests.log 2>&1; tail -5 gpurun_out/r03_gpu_tests.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r03_bench_headline.json 2> gpurun_out/r03_bench_headline.err; tail -c 600 gpurun_out/r03_bench_headline.json
python bench.py --steps 5 --warmup 2 --pose synthetic:0 --no-cpu-baseline > gpurun_out/r03_bench_headline_synthetic_pose.json 2> gpurun_out/r03_bench_syn.err; tail -c 300 gpurun_out/r03_bench_headline_synthetic_pose.json
python tools/microbench.py > gpurun_out/r03_microbench.json 2> gpurun_out/r03_microbench.err; tail -c 300 gpurun_out/r03_microbench.json
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-breakdown > $R/gpurun_out/r03_prof_bench.json 2> $R/gpurun_out/r03_prof_bench.err
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r03_headline_kernel_stats.csv; head -12 $R/gpurun_out/r03_headline_kernel_stats.csv
cd $R; IA_PROFILE_TAG=r03 python tools/pmc_traffic.py > gpurun_out/r03_pmc_traffic.log 2>&1; head -12 gpurun_out/r03_pmc_traffic.log
