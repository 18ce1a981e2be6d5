# Developer: rocprofv3 passes of a library variant on the serial C4 tick (tools/dev_kernel_times.py: every kernel on the
# caller's stream).   bash tools/dev_profile.sh <tag> <library.so> [config] [steps]
# Leaves gpurun_out/<tag>_kernel_stats.csv and gpurun_out/<tag>_sq.txt.
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; export SMX_LIBRARY=$(realpath $2); CFG=${3:-c4}; STEPS=${4:-60}
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/prof_$TAG; mkdir -p $OUT/prof_$TAG
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG/kt -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dev_kernel_times.py $CFG $STEPS > $OUT/prof_$TAG/kt.txt 2> $OUT/prof_$TAG/kt.err; echo "kernel-trace rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/prof_$TAG/pmc_a -- python3 $GRAFT_REPO_ROOT/tools/dev_kernel_times.py $CFG $STEPS > /dev/null 2> $OUT/prof_$TAG/pmc_a.err; echo "pmc a rc=$?"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/prof_$TAG/pmc_b -- python3 $GRAFT_REPO_ROOT/tools/dev_kernel_times.py $CFG $STEPS > /dev/null 2> $OUT/prof_$TAG/pmc_b.err; echo "pmc b rc=$?"
cd $GRAFT_REPO_ROOT
cp $(find $OUT/prof_$TAG/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python tools/pmc_summary.py $OUT/prof_$TAG/pmc_a $OUT/prof_$TAG/pmc_b > $OUT/${TAG}_sq.txt
rm -rf $OUT/prof_$TAG/pmc_a $OUT/prof_$TAG/pmc_b $OUT/prof_$TAG/kt
head -14 $OUT/${TAG}_kernel_stats.csv | cut -c1-120
