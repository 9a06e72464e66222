cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02c
python -m pytest tests -m gpu -q -x > gpurun_out/r02c/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02c/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02c/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r02c/smoke.log
bash tools/run_round.sh r02c
