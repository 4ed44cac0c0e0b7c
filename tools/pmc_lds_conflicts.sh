#!/bin/bash
# LDS bank-conflict cycles of the two large-register-tile GEMM kernels against their LDS-active cycles (one --pmc pass, no trace, per the pool's rules):
#   bash tools/pmc_lds_conflicts.sh        -> gpurun_out/lds_conflicts_pmc.csv
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $OUT/ntb -o ntb -- python3 $REPO/tools/bench_ntb.py 5 > $OUT/ntb.log 2>&1
python3 $REPO/tools/prof_summarize.py $(find $OUT/ntb -name '*.db' | head -1) $OUT/ntb
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $OUT/tnb -o tnb -- python3 $REPO/tools/bench_tnb.py 3 > $OUT/tnb.log 2>&1
python3 $REPO/tools/prof_summarize.py $(find $OUT/tnb -name '*.db' | head -1) $OUT/tnb
rm -rf $OUT/ntb $OUT/tnb
cat $OUT/ntb_pmc.csv $OUT/tnb_pmc.csv | grep -v "at::\|elementwise\|Cijk\|pack_kernel\|transpose\|zero\|fill" > $REPO/gpurun_out/lds_conflicts_pmc.csv
cat $REPO/gpurun_out/lds_conflicts_pmc.csv
