cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt5 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt5 -- python $R/tools/train_phys_bench.py > /dev/null 2>&1
F=$(find /tmp/kt5 -name "*kernel_trace.csv" | head -1)
python $R/tools/gap_timeline.py $F 835 | tee $R/gpurun_out/r05_config4_gaps.json
