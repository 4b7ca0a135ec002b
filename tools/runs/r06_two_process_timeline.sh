#!/bin/bash
# the two-process half of tools/runs/r06_concurrency_timelines.sh alone (4 Mi-ray secondary chunks, one march stream per process: 2 x ~90 GiB)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F="--steps 4 --warmup 2 --no-search-modes --no-cpu-baseline --no-config2 --no-config4 --no-breakdown"
tr() { find $1 -name "*kernel_trace.csv" | head -1; }
export IA_SECONDARY_CHUNK=$((1 << 22)) IA_SECONDARY_STREAMS=1 IA_MAX_SEARCH_POINTS=60000000 IA_BENCH_ARENA_GIB=40
rm -rf /tmp/ka0 /tmp/ka1 /tmp/ka2
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ka0 -- python $R/bench.py $F > $R/gpurun_out/r06_tl_a0.json 2>/dev/null
(timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ka1 -- python $R/bench.py $F > $R/gpurun_out/r06_tl_a1.json 2>$R/gpurun_out/r06_tl_a1.err) &
(timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ka2 -- python $R/bench.py $F > $R/gpurun_out/r06_tl_a2.json 2>$R/gpurun_out/r06_tl_a2.err) &
wait
for k in a0 a1 a2; do python -c "import json,sys; d=json.loads(open('$R/gpurun_out/r06_tl_$k.json').read().strip().splitlines()[-1]); print('$k', d['ms_per_step'], d['value'], d['config']['peak_reserved_memory_GiB'])"; done | tee $R/gpurun_out/r06_timelines_ms_two_processes.txt
python $R/tools/overlap_timeline.py $(tr /tmp/ka0) --steps 3 > $R/gpurun_out/r06_timeline_4mi_chunks_alone.json
python $R/tools/overlap_timeline.py $(tr /tmp/ka1) $(tr /tmp/ka2) --steps 3 > $R/gpurun_out/r06_timeline_two_processes.json
head -c 400 $R/gpurun_out/r06_timeline_two_processes.json
