#!/bin/bash
# inference only (relight, spp 256): (i) one process, 1 vs 2 streams; (ii) two processes at once (1 stream each)
export IA_SECONDARY_CHUNK=$((1 << 22)) IA_MAX_SEARCH_POINTS=40000000
get() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['s_per_frame_per_gpu'], d['frames'])"; }
for n in 1 2; do IA_SECONDARY_STREAMS=$n python tools/relight_bench.py --spp 256 --frames 8 > gpurun_out/rl_s$n.json 2>gpurun_out/rl_s$n.err; get gpurun_out/rl_s$n.json "solo streams=$n"; done
(IA_SECONDARY_STREAMS=1 python tools/relight_bench.py --spp 256 --frames 8 > gpurun_out/rl_a.json 2>gpurun_out/rl_a.err) &
(IA_SECONDARY_STREAMS=1 python tools/relight_bench.py --spp 256 --frames 8 > gpurun_out/rl_b.json 2>gpurun_out/rl_b.err) &
wait
get gpurun_out/rl_a.json "concurrent a"; get gpurun_out/rl_b.json "concurrent b"
