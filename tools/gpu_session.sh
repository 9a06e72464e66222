#!/bin/bash
# One parameterised GPU-box session script (round 4 on; replaces the per-pass tools/session_r03*.sh of round 3):
#   gpurun -- 'bash tools/gpu_session.sh <step> [args] [-- <step> [args] ...]'
# Steps (each writes under gpurun_out/<step>*; a failing step stops the chain):
#   test [pytest args]     GPU suite (default: tests -m gpu -q), log in gpurun_out/pytest_gpu.log
#   smoke                  __graft_entry__.smoke()
#   abenv VAR=VAL [bench args]   same-box A/B of an environment switch of libesdg_hip.so (e.g. ESDG_V2=rhs): new/base x 3
#   ablib alt.so [bench args]    same-box A/B of two builds (libesdg_hip.so vs esdg_cns_amd/variants/<alt.so>)
#   testlib alt.so [pytest args] the GPU tests with esdg_cns_amd/variants/<alt.so> in place of the library
#   round1 TAG / round2 TAG      the profile set of a round in two calls of under 20 minutes (collect with tools/collect_round.sh TAG)
#   kstats TAG [bench args]      rocprofv3 --kernel-trace --stats of one bench run, per-kernel averages
#   bench [bench args]           python bench.py ... > gpurun_out/bench_<n>.json
#   py script.py [args]          python <script> (log in gpurun_out/py_<name>.log)
#   sq TAG [bench args]          SQ instruction-mix / LDS counters per kernel (tools/pmc_sq.sh), gpurun_out/sq_TAG.txt
#   ubench name                  build + run tools/ubench/<name>.hip (log in gpurun_out/ubench_<name>.log)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
grepms() { grep -o '"ms_per_step": [0-9.]*\|"phase_ms": \[[^]]*\]\|"kernel_ms": [0-9.]*\|"lsrk[a-z0-9_]*": [0-9.]*' | tr '\n' ' '; }
run_step() {
  local step=$1; shift
  case $step in
    test)
      local a=("$@"); [ ${#a[@]} -eq 0 ] && a=(tests -m gpu -q)
      timeout -k 10 1000 python -m pytest "${a[@]}" > gpurun_out/pytest_gpu.log 2>&1; local rc=$?
      tail -5 gpurun_out/pytest_gpu.log; return $rc ;;
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; local rc=$?
      tail -2 gpurun_out/smoke.log; return $rc ;;
    abenv)   # (both legs on the A/B build of the library -- the one that reads environment switches)
      local kv=$1; shift
      export ESDG_HIP_LIB=$PWD/esdg_cns_amd/libesdg_hip_ab.so
      [ -f "$ESDG_HIP_LIB" ] || { echo "missing $ESDG_HIP_LIB"; return 2; }
      for v in new base new base new base; do
        echo -n "$v: "
        if [ $v = base ]; then (export "$kv"; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grepms)
        else timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grepms; fi
        echo
      done
      unset ESDG_HIP_LIB ;;
    ablib)   # (the base build is selected through ESDG_HIP_LIB: the in-tree library is never overwritten)
      local alt=esdg_cns_amd/variants/$1; shift
      [ -f "$alt" ] || { echo "missing variant $alt"; return 2; }
      for v in new base new base new base; do
        echo -n "$v: "
        if [ $v = base ]; then (export ESDG_HIP_LIB=$PWD/$alt; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grepms)
        else timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grepms; fi
        echo
      done ;;
    testlib)
      local alt=esdg_cns_amd/variants/$1; local nm=$1; shift
      [ -f "$alt" ] || { echo "missing variant $alt"; return 2; }
      local a=("$@"); [ ${#a[@]} -eq 0 ] && a=(tests -m gpu -q)
      (export ESDG_HIP_LIB=$PWD/$alt; timeout -k 10 1000 python -m pytest "${a[@]}" > gpurun_out/pytest_$nm.log 2>&1); local rc=$?
      tail -4 gpurun_out/pytest_$nm.log; return $rc ;;
    round1)   # final pass of a round, part 1: GPU suite, smoke, rocprofv3 + PMC passes and the default bench line for cns (cfg3)
      local tag=$1; local O=gpurun_out/$tag; mkdir -p $O
      timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
      cp gpurun_out/parity_errors.json $O/parity_errors.json
      python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
      bash tools/run_round.sh $tag ;;
    round2)   # part 2: Euler (cfg2) and hex (cfg5 per GPU) profile sets, SQ counters of the three workloads
      local tag=$1; local O=gpurun_out/$tag; mkdir -p $O
      bash tools/profile_other_configs.sh $tag
      bash tools/pmc_sq.sh ${tag}_cns > $O/sq_counters.txt 2>&1; tail -3 $O/sq_counters.txt
      bash tools/pmc_sq.sh ${tag}_hex --formulation hex > $O/hex_sq_counters.txt 2>&1; tail -3 $O/hex_sq_counters.txt
      bash tools/pmc_sq.sh ${tag}_euler --formulation euler --kx 256 --ky-per-gpu 256 > $O/euler_sq_counters.txt 2>&1; tail -3 $O/euler_sq_counters.txt ;;
    kstats)
      local tag=$1; shift
      bash tools/kstats.sh "$tag" "$@" ;;
    bench)
      local n=$(ls gpurun_out/bench_*.json 2>/dev/null | wc -l)
      timeout -k 10 900 python bench.py "$@" > gpurun_out/bench_$n.json 2> gpurun_out/bench_$n.err; local rc=$?
      tail -c 3000 gpurun_out/bench_$n.json; return $rc ;;
    py)
      local s=$1; shift
      timeout -k 10 900 python "$s" "$@" > gpurun_out/py_$(basename $s .py).log 2>&1; local rc=$?
      tail -25 gpurun_out/py_$(basename $s .py).log; return $rc ;;
    sq)      # SQ instruction-mix counters of one bench run: sq TAG [bench args]
      local tag=$1; shift
      bash tools/pmc_sq.sh "$tag" "$@" > gpurun_out/sq_$tag.txt 2>&1; local rc=$?
      tail -12 gpurun_out/sq_$tag.txt; return $rc ;;
    ubench)
      local n=$1; shift
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ub_$n tools/ubench/$n.hip || return 1
      timeout -k 10 300 /tmp/ub_$n "$@" > gpurun_out/ubench_$n.log 2>&1; local rc=$?
      cat gpurun_out/ubench_$n.log | tail -40; return $rc ;;
    *) echo "unknown step $step"; return 2 ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then
    echo "=== ${args[*]}"; run_step "${args[@]}" || { echo "step failed: ${args[*]}"; exit 1; }
    args=()
  else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && { echo "=== ${args[*]}"; run_step "${args[@]}" || { echo "step failed: ${args[*]}"; exit 1; }; }
exit 0
