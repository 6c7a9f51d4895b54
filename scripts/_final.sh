cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06_final
bash scripts/run_profiles.sh r06 > gpurun_out/run_profiles_r06.log 2>&1; tail -2 gpurun_out/run_profiles_r06.log | cut -c1-200
cd "$GRAFT_REPO_ROOT"
(time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/r06_final/bench_driver_cmd.log 2>&1; grep '^{"metric"' gpurun_out/r06_final/bench_driver_cmd.log | tail -1 > gpurun_out/r06_final/bench_driver_cmd.json; cut -c1-200 gpurun_out/r06_final/bench_driver_cmd.json
(time timeout 3300 python -m pytest tests/ -x -q -m gpu) > gpurun_out/r06_final/suite.log 2>&1
grep -E "passed|failed" gpurun_out/r06_final/suite.log | tail -2
cp gpurun_out/parity_report.txt gpurun_out/r06_final/parity_report.txt 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
