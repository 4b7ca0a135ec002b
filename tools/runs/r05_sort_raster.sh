#!/bin/bash
# the order inside a block of cells: Morton (default) against x-fastest raster of 8^3 / 16^3 / 32^3 blocks (IA_SORT_RASTER_BITS = 3 / 4 / 5); per-kernel ms of the
# instrumented step (one stream), same box
R=${GRAFT_REPO_ROOT:-/root/repo}
for rb in 0 3 4 5 0; do
  IA_SORT_RASTER_BITS=$rb timeout 400 python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-config2 --no-search-modes 2>/dev/null | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read()); k = b['kernel_breakdown_ms_per_step']
print(json.dumps(dict(raster_bits=$rb, ms_per_step=b['ms_per_step'], search=k['ia_fuse_broyden_spec_rows']['ms_per_step'], hash_gather=k['ia_hashgrid_fwd_xcd']['ms_per_step'], head=k['ia_sdf_levels_fwd']['ms_per_step'], sort=k['ia_morton_order']['ms_per_step'])))"
done
