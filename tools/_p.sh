python bench.py --task train --no-cpu-baseline 2>&1 | tail -1
python bench.py --task train --dtype bf16 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
python bench.py --task train_sisr --no-cpu-baseline 2>&1 | tail -2 | cut -c1-300
