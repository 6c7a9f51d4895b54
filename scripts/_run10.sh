cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06z; mkdir -p $out; rm -f $out/ab.txt
for r in 1 2 3; do
 for cfg in "old:X=1" "new:SEMSEG_TILE_TABLE_SP=$PWD/gpurun_variants/tile_table_sp_new.json"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "bs2 $name $(env $envs timeout 200 python scripts/step_time.py 2 30 2>&1 | tail -1 | cut -c1-40)" >> $out/ab.txt
 done
done
for r in 1 2; do
 for cfg in "old:X=1" "new:SEMSEG_TILE_TABLE_SP=$PWD/gpurun_variants/tile_table_sp_new.json"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "bs16 $name $(env $envs timeout 300 python scripts/step_time.py 16 10 2>&1 | tail -1 | cut -c1-40)" >> $out/ab.txt
  echo "bs4 $name $(env $envs timeout 300 python scripts/step_time.py 4 20 2>&1 | tail -1 | cut -c1-40)" >> $out/ab.txt
 done
done
cat $out/ab.txt
timeout 1500 python scripts/make_tile_table.py $out/tile_table.json > $out/tile_table_f32.log 2>&1
tail -3 $out/tile_table_f32.log
