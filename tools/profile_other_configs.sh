#!/bin/bash
# rocprofv3 passes + default bench for the Euler (cfg2) and hex (cfg5 per GPU) workloads:  bash tools/profile_other_configs.sh TAG
TAG=${1:-r03}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/${TAG}_euler gpurun_out/${TAG}_hex
bash tools/profile_round.sh ${TAG}_euler --formulation euler --kx 256 --ky-per-gpu 256 > gpurun_out/${TAG}_euler/profile.log 2>&1; tail -2 gpurun_out/${TAG}_euler/profile.log
python bench.py --formulation euler --kx 256 --ky-per-gpu 256 > gpurun_out/${TAG}_euler/bench_default.json 2> gpurun_out/${TAG}_euler/bench.err; tail -c 600 gpurun_out/${TAG}_euler/bench_default.json
bash tools/profile_round.sh ${TAG}_hex --formulation hex > gpurun_out/${TAG}_hex/profile.log 2>&1; tail -2 gpurun_out/${TAG}_hex/profile.log
python bench.py --formulation hex > gpurun_out/${TAG}_hex/bench_default.json 2> gpurun_out/${TAG}_hex/bench.err; tail -c 600 gpurun_out/${TAG}_hex/bench_default.json
