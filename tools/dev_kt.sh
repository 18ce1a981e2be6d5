# Developer: rocprofv3 kernel-trace stats of library variants on the serial tick.  bash tools/dev_kt.sh <config> <steps> <lib.so>...
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CFG=$1; STEPS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out
for LIB in "$@"; do
  TAG=$(basename $LIB .so)
  export SMX_LIBRARY=$(realpath $LIB)
  rm -rf $OUT/kt_$TAG
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt_$TAG -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dev_kernel_times.py $CFG $STEPS > $OUT/kt_$TAG.txt 2>&1)
  cp $(find $OUT/kt_$TAG -name "*kernel_stats.csv" | head -1) $OUT/kt_${TAG}.csv
  rm -rf $OUT/kt_$TAG
  echo "== $TAG"; head -12 $OUT/kt_${TAG}.csv | cut -d, -f1,2,4 | sed 's/"//g' | column -s, -t | cut -c1-100
done
