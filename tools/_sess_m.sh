mkdir -p gpurun_out/r06_m; O=gpurun_out/r06_m
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q --timeout=900 --timeout-method=thread -k "comb_top_tables or full_size or comb_table_shapes or lookup_tables" > $O/pytest_top.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_top.log
F="--no-cpu-baseline --no-microbench --no-fallbacks --no-end-to-end --no-configs --no-latency --steps 10"
for i in 1 2; do
  timeout 400 python bench.py $F --detail $O/d100_$i.json > $O/b100_$i.json 2> $O/b100_$i.err
  timeout 400 python bench.py $F --lookup-budget-gb 180 --detail $O/d180_$i.json > $O/b180_$i.json 2> $O/b180_$i.err
done
for f in $O/b1*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config']['msm_method'], d['config']['msm_table_bytes'], d['roofline']['avg_launch_us'], d['roofline']['sclk_mhz'])"; done; tail -3 $O/b180_1.err
