# the headline step on the eight committed frames of the reference's pose files (same binary, same box): ms per step, rays/s, sample counts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_bench_poses.jsonl
: > $O
for pose in male-3-casual:0 male-3-casual:40 male-3-casual:80 male-3-casual:113 aist:0 aist:100 aist:200 aist:319; do
  timeout 300 python $R/bench.py --pose $pose --steps 5 --warmup 2 --no-cpu-baseline --no-config2 --no-config4 --no-breakdown 2>/dev/null | tail -1 | \
    python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['config']['samples']; ds=d.get('deformer_search') or {}
print(json.dumps(dict(pose='$pose', ms_per_step=d['ms_per_step'], rays_per_s=d['value'], n_secondary=s.get('n_secondary'), n_fg=s.get('n_fg'), deform_points=s.get('deform_points'),
  search_to_the_end_ms=(ds.get('search_to_the_end') or {}).get('ms_per_step'), candidate_set_differs=(ds.get('vs_search_to_the_end_on_this_frame') or {}).get('candidate_set_differs'),
  canary=ds.get('canary_on_one_step'))))" >> $O
done
cat $O
