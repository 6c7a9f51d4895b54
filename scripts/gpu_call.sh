cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "tile_codes" 2>&1 | tail -3
SEMSEG_TILE_TUNE=1 timeout 900 python scripts/make_tile_table.py gpurun_out/tile_table.json 2>&1 | grep -v amdgpu | tail -10
cp gpurun_out/tile_table.json semseg_amd/tile_table.json
run() { tag=$1; shift; "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
for b in 16 8 4 2; do run "bs$b" timeout 300 python bench.py --global-batch $b --no-cpu-baseline --steps 10 --warmup 3 --module-steps 0 --no-kernel-timing; done
SEMSEG_FORCE_DIST=1 run "bs2-forced-rccl" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --global-batch 2 --no-cpu-baseline --steps 20 --warmup 5 --no-kernel-timing --module-steps 0
run "cfg3-713-bs2" timeout 300 python bench.py --size 713 --classes 19 --global-batch 2 --no-cpu-baseline --steps 10 --warmup 3 --module-steps 0 --no-kernel-timing
