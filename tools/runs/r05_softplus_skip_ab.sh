#!/bin/bash
# SDF value head: wave-uniform skip of the Softplus transcendentals (-DIA_SOFTPLUS_SKIP build) against the tree's kernel, same box, twice
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for lib in tree softskip; do
    if [ $lib = tree ]; then unset IA_AMD_LIB; else export IA_AMD_LIB=$R/intrinsicavatar_amd/_ab/libia_amd_$lib.so; fi
    echo -n "$lib "; IA_N=50000000 timeout 300 python $R/tools/sdf_head_probe.py 2>/dev/null | tail -1
  done
done
