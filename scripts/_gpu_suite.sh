cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_suite; mkdir -p $out
(time timeout 3300 python -m pytest tests/ -x -q -m gpu) > $out/suite.log 2>&1
tail -15 $out/suite.log
cp gpurun_out/parity_report.txt $out/parity_report.txt 2>/dev/null
