set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/t_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "bench rc=$?"
for c in c3 c4 c5; do python bench.py --config $c --steps 100 --warmup 20 > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; echo "bench $c rc=$?"; done
rm -rf gpurun_out/prof_final gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -o runc --output-format csv -- python3 bench.py --no-cpu-baseline --phase-steps 0 > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err; echo "rocprof rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --no-cpu-baseline --phase-steps 0 --steps 100 --warmup 20 > /dev/null 2> gpurun_out/pmc_fetch.err; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --no-cpu-baseline --phase-steps 0 --steps 100 --warmup 20 > /dev/null 2> gpurun_out/pmc_write.err; echo "pmc write rc=$?"
python tools/collect_hbm_traffic.py --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write --workload-key c2:loop:1024x8 --out gpurun_out/hbm_traffic.json; echo "collect rc=$?"
find gpurun_out/prof_final -name "*kernel_stats.csv" | head -2
