bash scripts/run_profiles.sh r03 2>&1 | tail -8
bash scripts/run_profiles_psa.sh r03 2>&1 | tail -6
