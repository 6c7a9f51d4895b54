cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
(time python -m pytest tests/test_plan_gpu.py tests/test_abi.py -x -q -m gpu) > $out/t_plan.log 2>&1
(time python -m pytest tests/test_headline_gpu.py -x -q -m gpu -k "pspnet50 or psanet") > $out/t_headline.log 2>&1
(time python -m pytest tests/test_insitu_bwd_gpu.py -x -q -m gpu -k "default_subset") > $out/t_insitu.log 2>&1
timeout 600 python bench.py --layers 50 --steps 20 --warmup 5 > $out/r06_pspnet50_b16.log 2>&1; grep "^{\"metric\"" $out/r06_pspnet50_b16.log | tail -1 > $out/r06_pspnet50_b16.json
tail -5 $out/t_plan.log $out/t_headline.log $out/t_insitu.log; cut -c1-300 $out/r06_pspnet50_b16.json
