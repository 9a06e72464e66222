#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against (run on the GPU box):
#   1. kernel trace + stats of the default bench command
#   2. separate PMC passes: FETCH_SIZE, WRITE_SIZE (cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"),
#      fp64 VALU instruction counts (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 + SQ_INSTS_VALU), no trace domains beside --kernel-trace
#   bash tools/profile_round.sh TAG [bench.py args...];  then  python3 tools/summarize_pmc.py TAG KEY
set -e
TAG=${1:-r02}
EXTRA="${@:2}"   # extra bench.py arguments, e.g. --formulation hex
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
# the hash of the kernel sources these passes measure (summarize_pmc.py stamps it into pmc_traffic.json; bench.py drops the
# PMC-derived fields when the sources have changed since)
python3 -c "import bench; print(bench.kernel_source_hash())" > $OUT/kernel_src_sha.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-rough-state $EXTRA > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rough-state $EXTRA > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rough-state $EXTRA > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/valu -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rough-state $EXTRA > $OUT/valu.log 2>&1 || echo "fp64 VALU counters unavailable"
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/fetch -name "*counter_collection.csv" -exec cp {} $OUT/fetch_counters.csv \;
find $OUT/write -name "*counter_collection.csv" -exec cp {} $OUT/write_counters.csv \;
find $OUT/valu -name "*counter_collection.csv" -exec cp {} $OUT/valu_counters.csv \; 2>/dev/null || true
tail -2 $OUT/trace.log
ls $OUT
