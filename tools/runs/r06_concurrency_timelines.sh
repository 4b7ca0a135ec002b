#!/bin/bash
# where does the 1.16 x of two processes on one GPU come from?  Kernel timelines (rocprofv3 --kernel-trace) of
#   (c) the default step (one process, secondary march on two streams / host threads),
#   (s) the same process with the march on ONE stream,
#   (a) two processes sharing the GPU (4 Mi-ray secondary chunks on one stream each, so that both fit the HBM), each traced, plus (a0) one such process alone,
# analysed by tools/overlap_timeline.py: time with >= 2 kernels in flight, which pairs co-run, hardware queues used.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F="--steps 4 --warmup 2 --no-search-modes --no-cpu-baseline --no-config2 --no-config4 --no-breakdown"
tr() { find $1 -name "*kernel_trace.csv" | head -1; }
rm -rf /tmp/kc /tmp/ks /tmp/ka0 /tmp/ka1 /tmp/ka2
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kc -- python $R/bench.py $F > $R/gpurun_out/r06_tl_c.json 2>/dev/null
IA_SECONDARY_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -- python $R/bench.py $F > $R/gpurun_out/r06_tl_s.json 2>/dev/null
export IA_SECONDARY_CHUNK=$((1 << 22)) IA_SECONDARY_STREAMS=1 IA_MAX_SEARCH_POINTS=60000000 IA_BENCH_ARENA_GIB=40      # two processes must fit 288 GB
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ka0 -- python $R/bench.py $F > $R/gpurun_out/r06_tl_a0.json 2>/dev/null
(timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ka1 -- python $R/bench.py $F > $R/gpurun_out/r06_tl_a1.json 2>/dev/null) &
(timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ka2 -- python $R/bench.py $F > $R/gpurun_out/r06_tl_a2.json 2>/dev/null) &
wait
for k in c s a0 a1 a2; do python -c "import json,sys; d=json.loads(open('$R/gpurun_out/r06_tl_$k.json').read().strip().splitlines()[-1]); print('$k', d['ms_per_step'], d['value'], d['config']['peak_reserved_memory_GiB'])"; done | tee $R/gpurun_out/r06_timelines_ms.txt
python $R/tools/overlap_timeline.py $(tr /tmp/kc) --steps 3 > $R/gpurun_out/r06_timeline_default_two_streams.json
python $R/tools/overlap_timeline.py $(tr /tmp/ks) --steps 3 > $R/gpurun_out/r06_timeline_one_stream.json
python $R/tools/overlap_timeline.py $(tr /tmp/ka0) --steps 3 > $R/gpurun_out/r06_timeline_8mi_chunks_alone.json
python $R/tools/overlap_timeline.py $(tr /tmp/ka1) $(tr /tmp/ka2) --steps 3 > $R/gpurun_out/r06_timeline_two_processes.json
head -c 600 $R/gpurun_out/r06_timeline_two_processes.json
