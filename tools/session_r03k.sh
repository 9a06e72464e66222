#!/bin/bash
# round-3 GPU session K: profile set of HEAD (cns cfg3, euler cfg2, hex cfg5 per GPU), smoke, 2-rank rehearsal of bench.py over gloo
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03k; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
bash tools/run_round.sh r03k
bash tools/profile_other_configs.sh r03k
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --oversubscribe --kx 256 --ky-per-gpu 64 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2-rank gloo rc=$?"; tail -c 1200 $O/bench_2rank_gloo.json; tail -3 $O/bench_2rank_gloo.err
