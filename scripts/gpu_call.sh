cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python scripts/wino_bench.py 16 2>&1 | grep -v amdgpu.ids
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --module-steps 0 > gpurun_out/r03_bench4.json 2> gpurun_out/r03_bench4.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_bench4.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["final_main_loss"])
PY
timeout 300 python bench.py --global-batch 2 --no-cpu-baseline --steps 20 --warmup 5 --module-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bs2', d['ms_per_step'])"
