cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/prof_r04_train_bf16; mkdir -p $R/gpurun_out/prof_r04_train_bf16
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04_train_bf16 -o t --output-format csv -- python $R/bench.py --task train --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $R/gpurun_out/prof_r04_train_bf16/log.txt 2>&1
f=$(find $R/gpurun_out/prof_r04_train_bf16 -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r04_train_bf16_kernel_stats.csv
head -24 $f | cut -c1-150
find $R/gpurun_out/prof_r04_train_bf16 -name "*kernel_trace.csv" -size +8M -delete
