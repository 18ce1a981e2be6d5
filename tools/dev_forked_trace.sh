# Developer: kernel trace of the forked tick (bench loop, no per-kernel timing).  bash tools/dev_forked_trace.sh <tag> [lib.so] [config]
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; [ -n "$2" ] && [ "$2" != "-" ] && export SMX_LIBRARY=$(realpath $2)
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/ft_$TAG
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/ft_$TAG -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config ${3:-c4} --steps 40 --warmup 10 --repeats 1 --no-cpu-baseline --phase-steps 0 > $OUT/ft_$TAG.json 2> $OUT/ft_$TAG.err)
cp $(find $OUT/ft_$TAG -name "*kernel_trace.csv" | head -1) $OUT/ft_${TAG}_trace.csv
rm -rf $OUT/ft_$TAG
python tools/dev_timeline.py $OUT/ft_${TAG}_trace.csv 5
