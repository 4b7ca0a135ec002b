# round-6 evidence: bench lines, kernel traces (headline one / two streams, config 4), launch audit, micro-benchmarks, PMC traffic.
# Every profiler call is bounded.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 900 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_form.json 2> $O/r06_bench_driver_form.err
export IA_SECONDARY_STREAMS=1
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-config4 --no-breakdown --no-search-modes > /dev/null 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/r06_headline_kernel_stats.csv
unset IA_SECONDARY_STREAMS
rm -rf /tmp/kt2 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-config4 --no-breakdown --no-search-modes > /dev/null 2>&1
cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) $O/r06_headline_kernel_stats_two_streams.csv
rm -rf /tmp/kt4 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -- python $R/tools/config4_bench.py --steps 20 --repeat 1 > $O/r06_config4_bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/kt4 -name "*kernel_stats.csv" | head -1) $O/r06_config4_kernel_stats.csv
python $R/tools/overlap_timeline.py $(find /tmp/kt4 -name "*kernel_trace.csv" | head -1) --steps 12 --skip-last 4 > $O/r06_config4_timeline.json 2>/dev/null
timeout 300 python $R/tools/config4_bench.py --steps 30 --repeat 3 > $O/r06_config4_bench.json 2>/dev/null
timeout 300 python $R/tools/launch_audit.py --out $O/r06_launch_audit_after.json > /dev/null 2>&1
timeout 300 python $R/tools/microbench.py > $O/r06_microbench.json 2>/dev/null
timeout 300 python $R/tools/relight_bench.py --spp 256 > $O/r06_relight_spp256.json 2>/dev/null
timeout 300 python $R/tools/relight_bench.py --spp 1024 --gi > $O/r06_relight_spp1024_gi.json 2>/dev/null
IA_PROFILE_TAG=r06 timeout 2400 python $R/tools/pmc_traffic.py > $O/r06_pmc_traffic.log 2>&1
tail -c 300 $O/r06_bench_driver_form.json
