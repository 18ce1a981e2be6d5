# Round-end measurement on the GPU box (gpurun), in parts so that a call stays within its limit:
#   bash tools/final_profile.sh <round> tests|bench|stats|pmc|pmc35|sq
# Summaries land in gpurun_out/<round>_*; the ones to be judged are copied into profiles/ afterwards.
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-r03}; PART=${2:-bench}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
pmc_pair() {  # <config> <workload key> <steps>
  local C=$1 KEY=$2 STEPS=$3
  rm -rf $OUT/pmc_fetch_$C $OUT/pmc_write_$C
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$C -- python3 $GRAFT_REPO_ROOT/bench.py --config $C --no-cpu-baseline --phase-steps 0 --steps $STEPS --warmup 5 --repeats 1 > $OUT/pmc_fetch_$C.json 2> $OUT/pmc_fetch_$C.err); echo "pmc fetch $C rc=$?"
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$C -- python3 $GRAFT_REPO_ROOT/bench.py --config $C --no-cpu-baseline --phase-steps 0 --steps $STEPS --warmup 5 --repeats 1 > $OUT/pmc_write_$C.json 2> $OUT/pmc_write_$C.err); echo "pmc write $C rc=$?"
  python tools/collect_hbm_traffic.py --fetch $OUT/pmc_fetch_$C --write $OUT/pmc_write_$C --workload-key $KEY --bench-json $OUT/pmc_write_$C.json --out $OUT/${R}_${C}_hbm_traffic.json; echo "collect $C rc=$?"
  rm -rf $OUT/pmc_fetch_$C $OUT/pmc_write_$C
}
case $PART in
tests)
  python -m pytest tests -m gpu -x -q > $OUT/${R}_t_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/${R}_t_gpu.log
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 ;;
bench)
  python bench.py > $OUT/${R}_bench_c4.json 2> $OUT/bench_c4.err; echo "bench c4 rc=$?"
  python bench.py --steps 20 --warmup 5 > $OUT/${R}_bench_c4_driver.json 2> $OUT/bench_c4d.err; echo "bench c4 (driver's flags) rc=$?"
  python bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline > $OUT/${R}_bench_c4_first220.json 2> $OUT/bench_c4e.err; echo "bench c4 (ticks 20-220) rc=$?"
  for c in c2 c3 c5; do python bench.py --config $c --no-cpu-baseline > $OUT/${R}_bench_$c.json 2> $OUT/bench_$c.err; echo "bench $c rc=$?"; done
  for E in 512 1024 2048; do python bench.py --envs-per-gpu $E --no-cpu-baseline --phase-steps 0 > $OUT/${R}_bench_c4_shard_$E.json 2> $OUT/bench_shard_$E.err; echo "shard $E rc=$?"; done ;;
stats)
  for c in ${STATS_CONFIGS:-c4 c3 c5}; do
    rm -rf $OUT/prof_final $OUT/prof_serial
    (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_final -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu-baseline --phase-steps 0 > $OUT/prof_bench_$c.json 2> $OUT/prof_bench_$c.err); echo "rocprof $c rc=$?"
    cp $(find $OUT/prof_final -name "*kernel_stats.csv" | head -1) $OUT/${R}_kernel_stats_$c.csv
    cp $(find $OUT/prof_final -name "*kernel_trace.csv" | head -1) $OUT/ft_${R}_$c.csv
    python tools/dev_timeline.py $OUT/ft_${R}_$c.csv 5 > $OUT/${R}_timeline_$c.txt 2>&1
    (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_serial -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dev_kernel_times.py $c 60 > $OUT/prof_serial_$c.txt 2> $OUT/prof_serial_$c.err); echo "rocprof serial $c rc=$?"
    cp $(find $OUT/prof_serial -name "*kernel_stats.csv" | head -1) $OUT/${R}_kernel_stats_${c}_serial.csv
    rm -rf $OUT/prof_final $OUT/prof_serial $OUT/ft_${R}_$c.csv
    python tools/dev_slow_counts.py $c 40 > $OUT/${R}_slow_lists_$c.txt 2>&1
  done ;;
pmc) pmc_pair c4 c4:loop:4096x32 60 ;;
pmc35)
  pmc_pair c3 c3:intersections/4lane:2048x16 60
  pmc_pair c5 c5:minicity:4096x64 40 ;;
sq)
  bash tools/dev_profile.sh ${R}sq smarts_amd/libsmarts_mi355x.so c4 60 > $OUT/${R}_sq_profile.log 2>&1
  python tools/sq_table.py $OUT/${R}sq_sq.txt > $OUT/${R}_sq_counters_c4.txt 2>&1
  python tools/dev_spans.py c4 serial > $OUT/${R}_wave_spans_c4.txt 2>&1
  python tools/dev_spans.py c4 forked >> $OUT/${R}_wave_spans_c4.txt 2>&1 ;;
esac
