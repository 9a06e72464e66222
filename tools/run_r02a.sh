cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z_0-9]*\|SQ_VALU_MFMA[A-Z_0-9]*\|SQ_INSTS_MFMA[A-Z_0-9]*" | sort -u > gpurun_out/r02a/counters.txt
python bench.py > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
tail -c 3000 gpurun_out/r02a/bench_default.json
python bench.py --gpus 2 --backend gloo --oversubscribe --kx 256 --ky-per-gpu 64 --steps 10 --no-cpu-baseline > gpurun_out/r02a/bench_gloo2.json 2> gpurun_out/r02a/bench_gloo2.err; echo "gloo2 rc=$?"; tail -c 1200 gpurun_out/r02a/bench_gloo2.json; tail -3 gpurun_out/r02a/bench_gloo2.err
python bench.py --gpus 2 --steps 3; echo "nccl2 on one gpu rc=$? (expected 3)"
bash tools/profile_round.sh r02a > gpurun_out/r02a/profile.log 2>&1; tail -5 gpurun_out/r02a/profile.log
python3 tools/summarize_pmc.py r02a cns_N4_512x512 > gpurun_out/r02a/pmc.log 2>&1; tail -5 gpurun_out/r02a/pmc.log
cat gpurun_out/r02a/counters.txt | tr '\n' ' '
