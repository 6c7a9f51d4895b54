cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06y; mkdir -p $out; rm -f $out/ab.txt
(timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split_k_reduced or stem or conv_dgrad_fused or test_conv_fwd or test_conv_dgrad") > $out/t_split.log 2>&1
tail -12 $out/t_split.log
for r in 1 2 3; do
 for cfg in "fused8:SEMSEG_FUSED_SPLIT=1" "sep:SEMSEG_FUSED_SPLIT=0" "fused16:SEMSEG_FUSED_SPLIT_MAX=16" "fused4:SEMSEG_FUSED_SPLIT_MAX=4"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "$name $(env $envs timeout 200 python scripts/step_time.py 2 30 2>&1 | tail -1 | cut -c1-40)" >> $out/ab.txt
 done
done
cat $out/ab.txt
for cfg in "fused8:SEMSEG_FUSED_SPLIT=1" "sep:SEMSEG_FUSED_SPLIT=0" "fused16:SEMSEG_FUSED_SPLIT_MAX=16"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "bs16 $name $(env $envs timeout 300 python scripts/step_time.py 16 10 2>&1 | tail -1 | cut -c1-40)"
done
