#!/bin/bash
# One GPU session = a list of named steps, run in order on the gpurun box; every step writes under gpurun_out/<tag>/.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_session.sh r03a tests ntt_sweep bench'
# Steps: tests[:<pytest -k expr>]  ntt_sweep[:<args>]  ntt_ab  bench[:<args>]  env_bench:<VAR=VALUE args>  rocprof  ntt_trace  pmc  ubench  smoke  latency
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "=== $name $arg"
  case $name in
    tests)
      # (--timeout-method=thread: a test stuck inside a HIP / RCCL call is ended by os._exit, not left to the session's own limit)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -s --timeout=900 --timeout-method=thread -k "$arg" > "$out/pytest_gpu.log" 2>&1
      else timeout 1800 python -m pytest tests -m gpu -x -q --timeout=900 --timeout-method=thread > "$out/pytest_gpu.log" 2>&1; fi
      echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -4 "$out/pytest_gpu.log" ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$out/smoke.log" ;;
    ntt_sweep) timeout 600 python tools/ntt_sweep.py $arg >> "$out/ntt_sweep.jsonl" 2>> "$out/ntt_sweep.err"; echo "rc=$?"; tail -3 "$out/ntt_sweep.jsonl" ;;
    ntt_ab)   # every alternative build present as plonkathon_amd/libplonk_hip_<name>.so, same shapes
      for so in plonkathon_amd/libplonk_hip_*.so; do
        t=$(basename "$so" .so); t=${t#libplonk_hip_}
        PLONK_HIP_LIB=$so timeout 300 python tools/ntt_sweep.py --tag "$t" $arg >> "$out/ntt_ab.jsonl" 2>> "$out/ntt_ab.err"
      done
      timeout 300 python tools/ntt_sweep.py --tag default $arg >> "$out/ntt_ab.jsonl" 2>> "$out/ntt_ab.err"; echo "rc=$?" ;;
    bench) timeout 900 python bench.py --detail "$out/bench_detail.json" $arg > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"; wc -c "$out/bench.json"; cat "$out/bench.json"; echo ;;
    rocprof)
      ( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/rocprof" -o trace -- python "$OLDPWD/bench.py" $arg > "$OLDPWD/$out/bench_under_rocprof.json" 2> "$OLDPWD/$out/rocprof.err" ); echo "rocprof rc=$?"
      find "$out/rocprof" -name "*kernel_stats.csv" | head -1 | xargs -r head -12 ;;
    ntt_trace)   # rocprofv3 --kernel-trace --stats of the lone transforms of configs[3]: per-pass kernel names and durations
      ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$out/ntt_trace" -o t -- python "$OLDPWD/tools/ntt_only.py" ${arg:-16,18,20,22,24} 7 > "$OLDPWD/$out/ntt_trace.log" 2> "$OLDPWD/$out/ntt_trace.err" ); echo "ntt_trace rc=$?"
      python tools/ntt_trace_summary.py "$out/ntt_trace" "$out/ntt_trace.log" > "$out/ntt_kernel_stats.txt" 2>> "$out/ntt_trace.err"; cat "$out/ntt_kernel_stats.txt"
      find "$out/ntt_trace" -name "*kernel_stats.csv" | head -1 | xargs -r head -12 ;;
    pmc) bash tools/pmc_collect.sh "$out" $arg ;;
    pmcu) bash tools/pmc_ntt_util.sh "$out/pmcu" ;;
    valu) bash tools/pmc_valu.sh "$out/valu" ;;
    stepvalu) bash tools/pmc_step.sh "$out/stepvalu" $arg ;;
    msm_sweep) timeout 600 python tools/msm_sweep.py $arg > "$out/msm_sweep.jsonl" 2> "$out/msm_sweep.err"; echo "rc=$?"; cat "$out/msm_sweep.jsonl" ;;
    env_sweep)   # "VAR=VALUE <ntt_sweep args>": the sweep under one environment setting, rows tagged VAR=VALUE
      kv=${arg%% *}; rest=${arg#* }
      env "$kv" timeout 300 python tools/ntt_sweep.py --tag "$kv" $rest >> "$out/ntt_sweep.jsonl" 2>> "$out/ntt_sweep.err"; echo "rc=$?" ;;
    rocprof_named)   # "<name> <bench args>": kernel trace + stats of one bench configuration under profiles-style names
      nm=${arg%% *}; rest=${arg#* }
      ( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OLDPWD/$out/rocprof_$nm" -o trace -- python "$OLDPWD/bench.py" $rest > "$OLDPWD/$out/bench_under_rocprof_$nm.json" 2> "$OLDPWD/$out/rocprof_$nm.err" ); echo "rocprof rc=$?"
      find "$out/rocprof_$nm" -name "*kernel_stats.csv" | head -1 | xargs -r head -14 ;;
    env_bench)   # "VAR=VALUE <bench args>": bench.py under one environment setting -> bench_VAR=VALUE.json
      kv=${arg%% *}; rest=${arg#* }
      env "$kv" timeout 600 python bench.py $rest > "$out/bench_$kv.json" 2> "$out/bench_$kv.err"; echo "bench $kv rc=$?"; cut -c1-200 "$out/bench_$kv.json"; echo ;;
    latency) timeout 600 python tools/latency.py $arg > "$out/latency.json" 2> "$out/latency.err"; echo "rc=$?"; cat "$out/latency.json" ;;
    lat_trace)   # kernel trace of batches of one through the lock-step prover: where a single proof's 1.9 ms go
      ( cd /tmp && LATENCY_ONLY=b1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$out/lat_trace" -o t -- python "$OLDPWD/tools/latency.py" 11 20 > "$OLDPWD/$out/lat_trace.log" 2> "$OLDPWD/$out/lat_trace.err" ); echo "lat_trace rc=$?"; cat "$out/lat_trace.log"
      find "$out/lat_trace" -name "*kernel_stats.csv" | head -1 | xargs -r head -40 ;;
    ubench) for b in tools/ubench/*.bin; do timeout 120 "$b" > "$out/$(basename $b .bin).json" 2>&1; done ;;
    *) echo "unknown step $name" ;;
  esac
done
