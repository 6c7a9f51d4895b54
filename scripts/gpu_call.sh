cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python scripts/tune_conv.py run 2>&1
