#!/bin/bash
# Run on the GPU box (gpurun): the four rocprofv3 passes behind profiles/<tag>_*; digest with make_profiles.py.
# PMC passes are separate from each other and carry --kernel-trace only (no sys/runtime/hip/hsa traces).
tag=${1:-r01}
shift
X="$@"     # extra bench.py arguments, e.g. --global-batch 2 --steps 30 (profiles of the per-GPU batch an 8-GPU job runs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write $out/${tag}_mfma
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o bench -- python bench.py $X > $out/bench_${tag}_n1.log 2>&1
grep "^{\"metric\"" $out/bench_${tag}_n1.log | tail -1 > $out/bench_${tag}_n1.json
# the same command with every kernel on one stream: per-kernel durations without the side-stream concurrency (what the
# bench line's roofline / kernel_families are measured on)
rm -rf $out/${tag}_serial
SEMSEG_DEBUG=side_wgrad=0,hipri_main=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_serial -o bench -- python bench.py $X --no-cpu-baseline --no-exact --module-steps 0 > $out/bench_${tag}_serial.log 2>&1
f=$(find $out/${tag}_serial -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && [ "$f" != "$out/${tag}_serial/bench_kernel_stats.csv" ] && cp "$f" $out/${tag}_serial/bench_kernel_stats.csv
B="python bench.py $X --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-exact --module-steps 0"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -o pmc -- $B > $out/${tag}_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -o pmc -- $B > $out/${tag}_write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_mfma -o pmc -- $B > $out/${tag}_mfma.log 2>&1
# rocprofv3 may nest its output under <hostname>/: flatten to the names make_profiles.py reads
f=$(find $out/${tag}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && [ "$f" != "$out/${tag}_stats/bench_kernel_stats.csv" ] && cp "$f" $out/${tag}_stats/bench_kernel_stats.csv
for d in fetch write mfma; do
  f=$(find $out/${tag}_$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && [ "$f" != "$out/${tag}_$d/pmc_counter_collection.csv" ] && cp "$f" $out/${tag}_$d/pmc_counter_collection.csv
done
# gpurun merges at most 64 MiB back: the per-dispatch traces are not read by make_profiles.py
find $out/${tag}_stats $out/${tag}_serial $out/${tag}_fetch $out/${tag}_write $out/${tag}_mfma -name "*kernel_trace.csv" -delete
ls -la $out/${tag}_stats/bench_kernel_stats.csv $out/${tag}_fetch/pmc_counter_collection.csv $out/${tag}_write/pmc_counter_collection.csv $out/${tag}_mfma/pmc_counter_collection.csv
cat $out/bench_${tag}_n1.json | cut -c1-200
