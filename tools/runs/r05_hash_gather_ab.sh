#!/bin/bash
# hash gather variants (-D builds of csrc/hashgrid.hip) against the tree's kernel: ms per 50 M sorted points (tools/sdf_head_probe.py), same box, twice
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for lib in tree "$@"; do
    if [ $lib = tree ]; then unset IA_AMD_LIB; else export IA_AMD_LIB=$R/intrinsicavatar_amd/_ab/libia_amd_$lib.so; fi
    echo -n "$lib "; IA_N=50000000 timeout 300 python $R/tools/sdf_head_probe.py 2>/dev/null | tail -1
  done
done
