set -e
mkdir -p gpurun_out/r02
bash profiles/profile.sh r02_h_sacgail --preset sac_gail > gpurun_out/r02/profile_h_sacgail.log 2>&1
tail -1 gpurun_out/r02/profile_h_sacgail.log | cut -c1-300
bash profiles/profile.sh r02_h_final > gpurun_out/r02/profile_h_final.log 2>&1
tail -1 gpurun_out/r02/profile_h_final.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()"
