set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --steps 7 --warmup 2 > $R/gpurun_out/r04_bench_headline.json 2> $R/gpurun_out/r04_bench_headline.err
python $R/bench.py --steps 5 --warmup 2 --pose synthetic:0 --no-cpu-baseline --no-config2 > $R/gpurun_out/r04_bench_headline_synthetic_pose.json 2>/dev/null
# kernel trace with the secondary march on ONE stream: per-kernel durations of kernels that have the device to themselves (what bench.py's
# roofline prices; on two streams launches of two chunks overlap and every duration in the trace includes the other chunk's share)
export IA_SECONDARY_STREAMS=1
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-breakdown --no-search-modes > /dev/null 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r04_headline_kernel_stats.csv
unset IA_SECONDARY_STREAMS
python $R/tools/microbench.py > $R/gpurun_out/r04_microbench.json 2>/dev/null
python $R/tools/relight_bench.py --spp 256 > $R/gpurun_out/r04_relight_spp256.json 2>/dev/null
python $R/tools/relight_bench.py --spp 1024 --gi > $R/gpurun_out/r04_relight_spp1024_gi.json 2>/dev/null
python $R/tools/train_phys_bench.py > $R/gpurun_out/r04_train_phys_config4.json 2>/dev/null
tail -c 400 $R/gpurun_out/r04_bench_headline.json
