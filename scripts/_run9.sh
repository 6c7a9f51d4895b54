cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06z; mkdir -p $out
(timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split_k_reduced") > $out/t_split.log 2>&1
tail -3 $out/t_split.log
timeout 1500 python scripts/make_tile_table.py --split $out/tile_table_sp.json > $out/tile_table.log 2>&1
tail -12 $out/tile_table.log
