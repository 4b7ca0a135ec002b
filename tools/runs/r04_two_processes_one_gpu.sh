#!/bin/bash
# do kernels of two independent processes on ONE GPU overlap usefully?  the headline step (secondary chunks of 8 Mi rays so that two
# processes fit the HBM), alone and twice at the same time: per-process ms per step.  2 x alone = no gain from concurrency.
export IA_SECONDARY_CHUNK=${IA_SECONDARY_CHUNK:-$((1 << 23))} IA_MAX_SEARCH_POINTS=${IA_MAX_SEARCH_POINTS:-80000000} IA_BENCH_ARENA_GIB=${IA_BENCH_ARENA_GIB:-90}
F="--hw ${HW:-540} --steps ${STEPS:-24} --warmup 3 --no-search-modes --no-cpu-baseline --no-config2 --no-breakdown"
python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('solo', d['ms_per_step'], d['value'], d['config']['peak_device_memory_GiB'])"
(python bench.py $F 2>gpurun_out/share_a.err > gpurun_out/share_a.json) &
(python bench.py $F 2>gpurun_out/share_b.err > gpurun_out/share_b.json) &
wait
for f in a b; do python -c "import json,sys; d=json.loads(open('gpurun_out/share_$f.json').read().strip().splitlines()[-1]); print('concurrent $f', d['ms_per_step'], d['value'])" || tail -3 gpurun_out/share_$f.err; done
