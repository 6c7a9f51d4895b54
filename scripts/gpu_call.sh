cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_full.log 2>&1; grep -E "passed|failed" gpurun_out/r03_pytest_full.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r03_pytest_full.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
