mkdir -p gpurun_out
for lpm in 1 4 16; do PLONK_MSM_COMB_LPM=$lpm timeout 300 python tools/msm_sweep.py 1152 comb20 2>/dev/null | cut -c1-420 | sed "s/^/lpm=$lpm /"; done
for M in 1 9 512 1536; do timeout 300 python tools/msm_sweep.py $M comb20 2>/dev/null | cut -c1-300 | sed "s/^/M=$M /"; done
