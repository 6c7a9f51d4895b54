cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06ae; mkdir -p $out
timeout 300 python scripts/host_issue_time.py 2 $out/host_issue_b2.json > $out/host_issue_b2.log 2>&1; tail -5 $out/host_issue_b2.log
timeout 300 python scripts/host_issue_time.py 16 $out/host_issue_b16.json > $out/host_issue_b16.log 2>&1; tail -5 $out/host_issue_b16.log
rm -rf /tmp/pm
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/pm -o pmc -- python scripts/step_time.py 16 2 > $out/pmc_sq.log 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python scripts/pmc_digest.py $f > $out/pmc_sq_b16.txt 2>&1; head -50 $out/pmc_sq_b16.txt
