#!/bin/bash
# SQ instruction-mix / LDS counters of the bench kernels (per-wave averages printed by tools/pmc_sq.py).
#   bash tools/pmc_sq.sh TAG [bench.py args...]
set -e
TAG=${1:-sq}
EXTRA="${@:2}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rough-state $EXTRA > $OUT/g$i.log 2>&1
  find $OUT/g$i -name "*counter_collection.csv" -exec cp {} $OUT/g$i.csv \;
done
python3 tools/pmc_sq.py $OUT
