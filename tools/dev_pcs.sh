cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pcs
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 1 --kernel-trace --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/tools/dev_kernel_times.py c4 12 > $OUT/log.txt 2>&1
echo "rc=$?"
ls -la $OUT | head; find $OUT -type f | head -20; tail -5 $OUT/log.txt
for f in $(find $OUT -name "*pc_sampling*csv" | head -2); do echo $f; head -5 $f; wc -l $f; done
