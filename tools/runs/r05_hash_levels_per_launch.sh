#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for l in 1 2 3; do echo -n "levels_per_launch=$l "; IA_HASH_LEVELS_PER_LAUNCH=$l IA_N=50000000 timeout 300 python $R/tools/sdf_head_probe.py 2>/dev/null | tail -1; done; done
