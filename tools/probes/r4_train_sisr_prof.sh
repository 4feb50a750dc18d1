cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/prof_r04_train_sisr; mkdir -p $R/gpurun_out/prof_r04_train_sisr
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r04_train_sisr -o t --output-format csv -- python $R/bench.py --task train_sisr --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $R/gpurun_out/prof_r04_train_sisr/log.txt 2>&1
f=$(find $R/gpurun_out/prof_r04_train_sisr -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r04_train_sisr_kernel_stats.csv
awk -F'",' 'NR>1{n=$1; gsub(/"/,"",n); split($2,a,","); printf "%-100s %6d %10.3f\n", substr(n,1,100), a[1]/10, a[2]/1e6/10}' $f | head -45
find $R/gpurun_out/prof_r04_train_sisr -name "*kernel_trace.csv" -size +8M -delete
