# HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory: a shorter launch path) on the launch-bound config-4 step and on the headline step
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 0 1; do
  echo -n "DEV_KERNARG=$v config4 "; HIP_FORCE_DEV_KERNARG=$v timeout 200 python tools/train_phys_bench.py 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done; done
for rep in 1 2 3; do for v in 0 1; do
  echo -n "DEV_KERNARG=$v headline "; HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-config2 --no-config4 --no-breakdown --no-search-modes 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done; done
