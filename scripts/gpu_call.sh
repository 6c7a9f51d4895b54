cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_full.log 2>&1; grep -E "passed|failed" gpurun_out/r03_pytest_full.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r03_pytest_full.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/run_profiles.sh r03 2>&1 | tail -3
bash scripts/run_profiles_psa.sh r03 2>&1 | tail -2
out=gpurun_out
rm -rf $out/q_bs2
SEMSEG_SIDE_WGRAD=0 SEMSEG_HIPRI_MAIN=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/q_bs2 -o bench -- python bench.py --global-batch 2 --steps 10 --warmup 3 --no-cpu-baseline --module-steps 0 --no-kernel-timing > $out/q_bs2.log 2>&1
f=$(find $out/q_bs2 -name "*kernel_stats.csv" | head -1); cp "$f" $out/q_bs2_kernel_stats.csv
grep '^{"metric"' $out/q_bs2.log | cut -c1-160
