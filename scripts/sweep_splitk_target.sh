for b in 2 4; do for t in 320 384 448 512 640 768; do
  r=$(SEMSEG_SPLITK_TARGET=$t timeout 200 python bench.py --global-batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; print(json.load(sys.stdin)['ms_per_step'])")
  echo "batch=$b splitk_target=$t ms_per_step=$r"
done; done
r=$(SEMSEG_SIDE_WGRAD=0 timeout 200 python bench.py --global-batch 2 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; print(json.load(sys.stdin)['ms_per_step'])"); echo "batch=2 side_wgrad=0 ms_per_step=$r"
