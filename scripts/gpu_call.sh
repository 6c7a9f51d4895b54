cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "winograd or psa_ops or gemm" 2>&1 | tail -2
for pz in 0 1; do echo "== SEMSEG_GEMM_PIPE=$pz"; SEMSEG_GEMM_PIPE=$pz python scripts/conv_bench.py 2>&1 | grep "l3 conv3\|l4 conv3"; SEMSEG_GEMM_PIPE=$pz python scripts/wino_bench.py 16 2>&1 | grep "l3 conv2\|l4 conv2" | cut -c1-250; done
