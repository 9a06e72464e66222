#!/bin/bash
# round-3 GPU session AB (after kt2_project): default bench lines of cns / euler against the regenerated pmc_traffic.json, SQ
# counters of the three CNS kernels, the sharded path on 2 and 4 ranks sharing the GPU (gloo, bit for bit against one engine;
# also with bench.py's own multi-rank code path), 300 repeated evaluations compared bit for bit
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03ab; mkdir -p $O
python bench.py > $O/bench_cns.json 2> $O/bench_cns.err; echo "bench cns rc=$?"
python bench.py --formulation euler --kx 256 --ky-per-gpu 256 > $O/bench_euler.json 2> $O/bench_euler.err; echo "bench euler rc=$?"
bash tools/pmc_sq.sh r03ab > $O/sq_cns.txt 2>&1; echo "pmc_sq rc=$?"; grep -A16 "kt2_project" $O/sq_cns.txt | head -17
for f in cns euler hex; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/check_sharded.py --backend gloo --formulation $f > $O/sharded2_$f.log 2>&1; echo "check_sharded 2 ranks $f rc=$?"; tail -2 $O/sharded2_$f.log
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29612 tools/check_sharded.py --backend gloo --formulation cns > $O/sharded4_cns.log 2>&1; echo "check_sharded 4 ranks cns rc=$?"; tail -2 $O/sharded4_cns.log
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --backend gloo --oversubscribe --kx 512 --ky-per-gpu 128 --steps 5 --warmup 2 > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err; echo "bench 2 ranks (gloo, one GPU) rc=$?"; tail -c 600 $O/bench_2ranks_gloo.json
timeout -k 10 600 python tools/soak_repro.py > $O/soak.log 2>&1; echo "soak rc=$?"; tail -4 $O/soak.log
