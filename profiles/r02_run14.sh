set -e
mkdir -p gpurun_out/r02
bash profiles/profile.sh r02_f_final > gpurun_out/r02/profile_f_final.log 2>&1
tail -1 gpurun_out/r02/profile_f_final.log
bash profiles/profile.sh r02_f_sacgail --preset sac_gail > gpurun_out/r02/profile_f_sacgail.log 2>&1
tail -1 gpurun_out/r02/profile_f_sacgail.log
