cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt4 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -- python $R/tools/train_phys_bench.py > $R/gpurun_out/s2_c4_bench.json 2>/dev/null
cp $(find /tmp/kt4 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/s2_c4_kernel_stats.csv
IA_CPROFILE=1 timeout 200 python $R/tools/train_phys_bench.py > /dev/null 2> $R/gpurun_out/s2_c4_cprofile.txt
cat $R/gpurun_out/s2_c4_bench.json
