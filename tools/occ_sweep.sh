cd "$GRAFT_REPO_ROOT"
export ESDG_HIP_LIB=$PWD/esdg_cns_amd/libesdg_hip_ab.so   # the A/B build reads ESDG_T2_WG_PER_CU
for w in default 3 4 5 6 8; do
  if [ $w = default ]; then unset ESDG_T2_WG_PER_CU; else export ESDG_T2_WG_PER_CU=$w; fi
  echo -n "wg_per_cu=$w: "
  timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
done
