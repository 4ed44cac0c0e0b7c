#!/bin/bash
# Profiles of one round, on the GPU box:  bash tools/profile_round.sh r02
#   kernel trace + stats of `bench.py --steps 2`, then three SEPARATE --pmc passes of `bench.py --steps 1` (FETCH_SIZE / WRITE_SIZE /
#   SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE: TCC and SQ counters never share a pass with a trace, per the pool's rules), reduced by
#   tools/prof_summarize.py to the small CSVs that are committed under profiles/.
set -e
TAG=${1:-r02}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT $REPO/profiles
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof args...
  local name=$1; shift
  rocprofv3 "$@" -d $OUT/$name -o $name -- python3 $REPO/bench.py --steps ${STEPS:-1} --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1
  local db=$(find $OUT/$name -name '*.db' | head -1)
  python3 $REPO/tools/prof_summarize.py $db $OUT/$name
  rm -rf $OUT/$name   # the rocpd databases are hundreds of MB: only the summaries travel back
  echo "$name done"; tail -c 300 $OUT/$name.log | head -c 300; echo
}
STEPS=2 run trace --kernel-trace --stats
cp $OUT/trace_kernel_stats.csv $REPO/profiles/${TAG}_bench_b64_steps2_kernel_stats.csv
run fetch --pmc FETCH_SIZE
cp $OUT/fetch_pmc.csv $REPO/profiles/${TAG}_bench_b64_pmc_fetch.csv
run write --pmc WRITE_SIZE
cp $OUT/write_pmc.csv $REPO/profiles/${TAG}_bench_b64_pmc_write.csv
run mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
cp $OUT/mfma_pmc.csv $REPO/profiles/${TAG}_bench_b64_pmc_mfma.csv
cp $REPO/profiles/${TAG}_*.csv $REPO/gpurun_out/ 2>/dev/null || true
