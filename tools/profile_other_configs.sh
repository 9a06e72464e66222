cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02h_euler gpurun_out/r02h_hex
bash tools/profile_round.sh r02h_euler --formulation euler --kx 256 --ky-per-gpu 256 > gpurun_out/r02h_euler/profile.log 2>&1; tail -2 gpurun_out/r02h_euler/profile.log
python bench.py --formulation euler --kx 256 --ky-per-gpu 256 > gpurun_out/r02h_euler/bench_default.json 2> gpurun_out/r02h_euler/bench.err; tail -c 600 gpurun_out/r02h_euler/bench_default.json
bash tools/profile_round.sh r02h_hex --formulation hex > gpurun_out/r02h_hex/profile.log 2>&1; tail -2 gpurun_out/r02h_hex/profile.log
python bench.py --formulation hex > gpurun_out/r02h_hex/bench_default.json 2> gpurun_out/r02h_hex/bench.err; tail -c 600 gpurun_out/r02h_hex/bench_default.json
