cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "winograd" 2>&1 | tail -2
run() { tag=$1; shift; "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
for b in 16 2; do run "bs$b" timeout 300 python bench.py --global-batch $b --no-cpu-baseline --steps 10 --warmup 3 --module-steps 0 --no-kernel-timing; done
