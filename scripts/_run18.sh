cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06ah; mkdir -p $out; rm -f $out/ab.txt
V=$PWD/gpurun_variants/libsemseg_oneacc.so
echo "== two accumulator sets"; python scripts/wgrad_noise.py 16 59 512 256 2>&1 | grep -v amdgpu; python scripts/wgrad_noise.py 16 59 1024 256 2>&1 | grep -v amdgpu | head -5
echo "== one set"; SEMSEG_HIP_LIB=$V python scripts/wgrad_noise.py 16 59 512 256 2>&1 | grep -v amdgpu | head -5
(timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wgrad") > $out/t.log 2>&1; tail -2 $out/t.log
for r in 1 2 3; do
  echo "bs16 twoacc $(python scripts/step_time.py 16 10 2>&1 | tail -1 | cut -c1-40)" >> $out/ab.txt
  echo "bs16 oneacc $(SEMSEG_HIP_LIB=$V python scripts/step_time.py 16 10 2>&1 | tail -1 | cut -c1-40)" >> $out/ab.txt
done
echo "bs2 twoacc $(python scripts/step_time.py 2 30 2>&1 | tail -1 | cut -c1-40)" >> $out/ab.txt
echo "bs2 oneacc $(SEMSEG_HIP_LIB=$V python scripts/step_time.py 2 30 2>&1 | tail -1 | cut -c1-40)" >> $out/ab.txt
cat $out/ab.txt
