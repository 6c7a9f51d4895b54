cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv or gemm or winograd or psa" 2>&1 | tail -3
for pz in 0 1; do echo "== SEMSEG_GEMM_PERSIST=$pz"; SEMSEG_GEMM_PERSIST=$pz python scripts/conv_bench.py 2>&1 | grep "1x1\|weighted"; SEMSEG_GEMM_PERSIST=$pz python scripts/wino_bench.py 16 2>&1 | grep "l3 conv2\|l4 conv2\|aux.0\|cls.0\|per step"; done
