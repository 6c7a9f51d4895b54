#!/bin/bash
# Run on the GPU box (gpurun): PSANet-101 465x465 bs16 (BASELINE.json configs[3]) bench line + rocprofv3 kernel stats,
# and the psamask kernels alone (scripts/psamask_bench.py) with kernel stats, FETCH_SIZE and WRITE_SIZE passes
# (PMC passes separate, --kernel-trace only).  Digest: scripts/make_profiles_psa.py <tag>.
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf $out/${tag}_psa_stats $out/${tag}_psamask_stats $out/${tag}_psamask_fetch $out/${tag}_psamask_write
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_psa_stats -o bench -- python bench.py --arch psa --size 465 > $out/bench_${tag}_psa.log 2>&1
grep "^{\"metric\"" $out/bench_${tag}_psa.log | tail -1 > $out/bench_${tag}_psa.json
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_psamask_stats -o pm -- python scripts/psamask_bench.py > $out/${tag}_psamask_bench.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_psamask_fetch -o pmc -- python scripts/psamask_bench.py > $out/${tag}_psamask_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_psamask_write -o pmc -- python scripts/psamask_bench.py > $out/${tag}_psamask_write.log 2>&1
for d in psa_stats psamask_stats; do
  f=$(find $out/${tag}_$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_$d.kernel_stats.csv
done
for d in psamask_fetch psamask_write; do
  f=$(find $out/${tag}_$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_$d.counters.csv
done
ls -la $out/${tag}_psa_stats.kernel_stats.csv $out/${tag}_psamask_stats.kernel_stats.csv $out/${tag}_psamask_fetch.counters.csv $out/${tag}_psamask_write.counters.csv
cut -c1-300 $out/bench_${tag}_psa.json
