bash scripts/run_profiles.sh r06 > gpurun_out/run_profiles_r06.log 2>&1
tail -3 gpurun_out/run_profiles_r06.log
bash scripts/run_profiles.sh r06_bs2 --global-batch 2 --steps 30 --warmup 5 > gpurun_out/run_profiles_r06_bs2.log 2>&1
tail -3 gpurun_out/run_profiles_r06_bs2.log
