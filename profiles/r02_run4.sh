set -e
mkdir -p gpurun_out/r02
bash profiles/pmc_wave.sh vreg_nodrain_1w profiles/ab/vreg_nodrain.so sac_gail 65536 > gpurun_out/r02/pmcw_vreg_nodrain_1w.txt
bash profiles/pmc_wave.sh vreg_nodrain_4w profiles/ab/vreg_nodrain.so sac_gail 262144 > gpurun_out/r02/pmcw_vreg_nodrain_4w.txt
cat gpurun_out/r02/pmcw_vreg_nodrain_1w.txt gpurun_out/r02/pmcw_vreg_nodrain_4w.txt
