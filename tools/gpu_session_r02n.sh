#!/bin/bash
# Why is the cache-hit probe of the lookup kernel 12 % faster?  (a) region probes: gathers stay random inside 4 MB
# regions but span 64 / 2048 / 8192 / all 30720 regions (address-translation reach); (b) clocks and power sampled
# with rocm-smi while the default bench runs (sustained-clock hypothesis).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # lib streams
  if [ $1 = main ]; then unset PLONK_HIP_LIB; else export PLONK_HIP_LIB=$R/plonkathon_amd/libplonk_hip_$1.so; fi
  timeout 400 python bench.py --steps 6 --streams $2 --no-cpu-baseline --no-microbench --no-fallbacks > gpurun_out/n_$1_$2.json 2> gpurun_out/n_$1_$2.err
  echo "$1 streams=$2 rc=$? $(grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*' gpurun_out/n_$1_$2.json | head -2 | tr '\n' ' ')"
}
( while true; do echo "T $(date +%s.%N)"; rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" ; sleep 0.4; done ) > gpurun_out/n_smi.log 2>&1 &
SMI=$!
sleep 2
echo "MARK main1 $(date +%s.%N)" >> gpurun_out/n_marks.log; run main 1
echo "MARK r64 $(date +%s.%N)" >> gpurun_out/n_marks.log; run r64 1
echo "MARK r2k $(date +%s.%N)" >> gpurun_out/n_marks.log; run r2k 1
echo "MARK r8k $(date +%s.%N)" >> gpurun_out/n_marks.log; run r8k 1
echo "MARK main2 $(date +%s.%N)" >> gpurun_out/n_marks.log; run main 2
echo "MARK end $(date +%s.%N)" >> gpurun_out/n_marks.log
kill $SMI
wc -l gpurun_out/n_smi.log
grep -E "sclk" gpurun_out/n_smi.log | sort | uniq -c | sort -rn | head -12
grep -E "Power" gpurun_out/n_smi.log | awk '{print $NF}' | sort -n | awk '{a[NR]=$1} END {print "power min/med/max", a[1], a[int(NR/2)], a[NR]}'
