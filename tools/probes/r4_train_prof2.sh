cd /tmp && export TMPDIR=/tmp
R=/root/repo
for mode in f32 bf16; do
rm -rf $R/gpurun_out/prof_train_$mode; mkdir -p $R/gpurun_out/prof_train_$mode
extra=""; if [ $mode = bf16 ]; then extra="--dtype bf16"; fi
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train_$mode -o t --output-format csv -- python $R/bench.py --task train $extra --steps 10 --warmup 2 --no-cpu-baseline --no-configs > $R/gpurun_out/prof_train_$mode/log.txt 2>&1
f=$(find $R/gpurun_out/prof_train_$mode -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/train_${mode}_kernel_stats.csv
echo "== $mode"; head -28 $f | cut -c1-150
find $R/gpurun_out/prof_train_$mode -name "*kernel_trace.csv" -size +8M -delete
done
