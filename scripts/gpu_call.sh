set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_psa_binding.py -m gpu -x -q -k "psamask or psa" > gpurun_out/r03_pytest_psa.log 2>&1; tail -3 gpurun_out/r03_pytest_psa.log
timeout 120 python scripts/psamask_bench.py > gpurun_out/r03_psamask_bench.log 2>&1; cat gpurun_out/r03_psamask_bench.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err; echo "bench rc $?"
SEMSEG_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --global-batch 2 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r03_bench2_bs2_dist.json 2> gpurun_out/r03_bench2_bs2_dist.err; echo "bs2 dist rc $?"
timeout 300 python bench.py --global-batch 2 --no-cpu-baseline --steps 20 --warmup 5 --module-steps 0 > gpurun_out/r03_bench2_bs2.json 2> gpurun_out/r03_bench2_bs2.err; echo "bs2 rc $?"
python - <<'PY'
import json
for f in ("r03_bench2","r03_bench2_bs2_dist","r03_bench2_bs2"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], (d.get("module_path") or {}).get("ms_per_step"), d.get("syncbn_collectives_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
