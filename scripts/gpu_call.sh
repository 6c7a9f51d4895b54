cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SEMSEG_RUN_COMPARATOR=1 COMPARATOR_FIND=0 timeout 330 python -m pytest tests/test_comparator_gpu.py -m gpu -q -s 2>&1 | tail -5
grep -i "comparator" gpurun_out/parity_report.txt | tail -2
