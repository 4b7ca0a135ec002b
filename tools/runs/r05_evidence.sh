# round-5 evidence: bench lines, kernel traces (one stream / two streams), micro-benchmarks, PMC traffic.  Every profiler call is bounded.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 900 python $R/bench.py --steps 7 --warmup 2 > $O/r05_bench_headline.json 2> $O/r05_bench_headline.err
export IA_SECONDARY_STREAMS=1
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-config4 --no-breakdown --no-search-modes > /dev/null 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/r05_headline_kernel_stats.csv
unset IA_SECONDARY_STREAMS
rm -rf /tmp/kt2 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-config4 --no-breakdown --no-search-modes > /dev/null 2>&1
cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) $O/r05_headline_kernel_stats_two_streams.csv
timeout 300 python $R/tools/microbench.py > $O/r05_microbench.json 2>/dev/null
timeout 300 python $R/tools/relight_bench.py --spp 256 > $O/r05_relight_spp256.json 2>/dev/null
timeout 300 python $R/tools/relight_bench.py --spp 1024 --gi > $O/r05_relight_spp1024_gi.json 2>/dev/null
timeout 300 python $R/tools/train_phys_bench.py > $O/r05_train_phys_config4.json 2>/dev/null
timeout 300 python $R/tools/sort_probe.py > $O/r05_sort_probe.json 2>/dev/null
timeout 2400 python $R/tools/pmc_traffic.py > $O/r05_pmc_traffic.log 2>&1
tail -c 300 $O/r05_bench_headline.json
