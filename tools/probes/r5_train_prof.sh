cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_r05_train; mkdir -p $R/gpurun_out/prof_r05_train
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r05_train -o t --output-format csv -- python $R/bench.py --task train --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $R/gpurun_out/prof_r05_train/log.txt 2>&1
f=$(find $R/gpurun_out/prof_r05_train -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r05_train_kernel_stats.csv
head -40 $f | cut -c1-200
find $R/gpurun_out/prof_r05_train -name "*kernel_trace.csv" -size +8M -delete
