cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench5.json 2> gpurun_out/r03_bench5.err; echo "bench rc $?"; tail -3 gpurun_out/r03_bench5.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_bench5.json").read().strip().splitlines()[-1])
print({k:v for k,v in d.items() if k not in ("kernel_families","kernel_families_in_step","config","roofline")})
print(d["roofline"])
for k,v in d["kernel_families"].items(): print(k,v)
PY
