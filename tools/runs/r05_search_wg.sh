cd $GRAFT_REPO_ROOT
for wg in 256 128 64; do
  echo -n "WG=$wg "; IA_BR_SPEC_WG=$wg timeout 200 python tools/search_ab.py intrinsicavatar_amd/libia_amd.so 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['candidates_rows_ms_eps0.001'], d['deform_sdf_ms_rows_eps0.001'], d['search_ms_eps0.001'])"
done
