# Round-end measurement on the GPU box (gpurun): tests, smoke, the bench lines and the rocprofv3 passes whose
# summaries are copied into profiles/ afterwards (tools/collect_hbm_traffic.py turns the PMC passes into JSON).
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=${1:-r02}
python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/t_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/${R}_bench_c4.json 2> gpurun_out/bench_c4.err; echo "bench c4 rc=$?"
python bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline > gpurun_out/${R}_bench_c4_first220.json 2> gpurun_out/bench_c4e.err; echo "bench c4 (ticks 20-220) rc=$?"
python tools/dev_feature_cost.py > gpurun_out/${R}_feature_cost.txt 2>&1; echo "feature cost rc=$?"
for c in c2 c3 c5; do python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${R}_bench_$c.json 2> gpurun_out/bench_$c.err; echo "bench $c rc=$?"; done
rm -rf gpurun_out/prof_final gpurun_out/prof_serial gpurun_out/pmc_fetch gpurun_out/pmc_write
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --phase-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err; echo "rocprof rc=$?"
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_serial -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dev_kernel_times.py c4 200 > $GRAFT_REPO_ROOT/gpurun_out/prof_serial.txt 2> $GRAFT_REPO_ROOT/gpurun_out/prof_serial.err; echo "rocprof serial rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --phase-steps 0 --steps 100 --warmup 20 --repeats 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.err; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --phase-steps 0 --steps 100 --warmup 20 --repeats 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_write.err; echo "pmc write rc=$?"
cd $GRAFT_REPO_ROOT
python tools/collect_hbm_traffic.py --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write --workload-key c4:loop:4096x32 --out gpurun_out/${R}_c4_hbm_traffic.json; echo "collect rc=$?"
find gpurun_out/prof_final gpurun_out/prof_serial -name "*kernel_stats.csv" | head -4
