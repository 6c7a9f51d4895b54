#!/bin/bash
# Run on the GPU box (gpurun): the bench lines behind the small-batch / multi-GPU-readiness statements of DESIGN.md
# (a claim without a kept bench line is not evidence).  Digest: copy gpurun_out/<tag>_*.json to profiles/.
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
line() { grep "^{\"metric\"" "$1" | tail -1; }
# per-GPU batch 2 (what each of 8 GPUs runs), single process, no collectives
timeout 300 python bench.py --global-batch 2 --steps 30 --warmup 5 --no-cpu-baseline --module-steps 0 > $out/${tag}_bs2.log 2>&1; line $out/${tag}_bs2.log > $out/${tag}_bs2.json
# the same through the N > 1 code path on one rank: 208 SyncBN all-reduces + bucketed gradient all-reduce over RCCL
SEMSEG_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --global-batch 2 --steps 30 --warmup 5 --no-cpu-baseline --no-exact --module-steps 0 > $out/${tag}_bs2_forced_rccl.log 2>&1; line $out/${tag}_bs2_forced_rccl.log > $out/${tag}_bs2_forced_rccl.json
# ... with the SyncBN statistics through the peer-memory exchange kernel instead of c10d (opt-in path)
SEMSEG_FORCE_DIST=1 SEMSEG_SYNCBN_XCHG=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --global-batch 2 --steps 30 --warmup 5 --no-cpu-baseline --no-exact --module-steps 0 > $out/${tag}_bs2_forced_xchg.log 2>&1; line $out/${tag}_bs2_forced_xchg.log > $out/${tag}_bs2_forced_xchg.json
# per-GPU batch 4 / 8
for b in 4 8; do timeout 300 python bench.py --global-batch $b --steps 16 --warmup 4 --no-cpu-baseline --no-exact --no-kernel-timing --module-steps 0 > $out/${tag}_bs$b.log 2>&1; line $out/${tag}_bs$b.log > $out/${tag}_bs$b.json; done
# BASELINE configs[2]: PSPNet-101 713^2, 19 classes, per-GPU batch 2
timeout 300 python bench.py --size 713 --classes 19 --global-batch 2 --steps 16 --warmup 4 --no-cpu-baseline --module-steps 0 > $out/${tag}_config3_713.log 2>&1; line $out/${tag}_config3_713.log > $out/${tag}_config3_713.json
# BASELINE configs[3]: PSANet-101 465^2 batch 16
timeout 400 python bench.py --arch psa --size 465 --steps 8 --warmup 2 --no-cpu-baseline --module-steps 0 > $out/${tag}_psanet.log 2>&1; line $out/${tag}_psanet.log > $out/${tag}_psanet.json
# two-stream backward vs everything on one stream, final code
timeout 300 python scripts/ab_libs.py $out/${tag}_two_stream_ab.json 16 2 two_stream one_stream::SEMSEG_DEBUG=side_wgrad=0,hipri_main=0 > $out/${tag}_two_stream_ab.log 2>&1
# host issue time of the three step drivers (launch by launch / C replay / hipGraph) at per-GPU batch 2 and 16
timeout 300 python scripts/host_issue_time.py 2 $out/${tag}_host_issue_b2.json > $out/${tag}_host_issue_b2.log 2>&1
timeout 300 python scripts/host_issue_time.py 16 $out/${tag}_host_issue_b16.json > $out/${tag}_host_issue_b16.log 2>&1
timeout 200 python scripts/psamask_bench.py > $out/${tag}_psamask_bench.log 2>&1
timeout 200 python scripts/bench_infer.py > $out/${tag}_infer.log 2>&1
for f in bs2 bs2_forced_rccl bs2_forced_xchg bs4 bs8 config3_713 psanet; do echo $f; cut -c1-260 $out/${tag}_$f.json; done
cat $out/${tag}_two_stream_ab.log; tail -12 $out/${tag}_psamask_bench.log; tail -2 $out/${tag}_infer.log
