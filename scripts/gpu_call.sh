cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mg in 0 1; do echo "== mask-group $mg"; SEMSEG_PSAMASK_MG=$mg timeout 120 python scripts/psamask_bench.py 2>/dev/null | grep "C-ABI"; done
SEMSEG_PSAMASK_MG=1 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "psamask_vs_oracle" 2>&1 | tail -2
