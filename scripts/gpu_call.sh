cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "label or ce_head or ignored or golden" 2>&1 | tail -4
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --module-steps 0 --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bs16', d['ms_per_step'], d['value'])"
