#!/bin/bash
# PMC wave-state / instruction-mix counters of tools/bench_attn.py (two SQ passes of 8 counters), reduced by tools/prof_summarize.py.
#   SPA3D_ATTN_BWD_MODE=4 NOMASK=1 bash tools/prof_attn.sh tag
set -e
TAG=${1:-attn}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 "$@" -d $OUT/$name -o $name -- python3 $REPO/tools/bench_attn.py > $OUT/$name.log 2>&1
  local db=$(find $OUT/$name -name '*.db' | head -1)
  python3 $REPO/tools/prof_summarize.py $db $OUT/$name
  rm -rf $OUT/$name
}
run state --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
run insts --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
cat $OUT/state_pmc.csv $OUT/insts_pmc.csv | grep "attn_bwd" > $REPO/gpurun_out/${TAG}_pmc.csv
cat $REPO/gpurun_out/${TAG}_pmc.csv
