cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_final; mkdir -p $out
(time python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')") > $out/smoke.log 2>&1; tail -3 $out/smoke.log
(time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5) > $out/bench_driver_cmd.log 2>&1; grep '^{"metric"' $out/bench_driver_cmd.log | tail -1 > $out/bench_driver_cmd.json; cut -c1-330 $out/bench_driver_cmd.json; tail -4 $out/bench_driver_cmd.log | grep real
(time timeout 3300 python -m pytest tests/ -x -q -m gpu) > $out/suite.log 2>&1
tail -8 $out/suite.log
cp gpurun_out/parity_report.txt $out/parity_report.txt 2>/dev/null
