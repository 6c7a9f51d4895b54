cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for f in 0 1; do echo "== fill $f"; SEMSEG_PSAMASK_FILL=$f timeout 120 python scripts/psamask_bench.py 2>/dev/null | grep "C-ABI" | grep bwd; done
SEMSEG_PSAMASK_FILL=1 timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_psa_binding.py -m gpu -x -q -k "psamask or psa" 2>&1 | tail -2
