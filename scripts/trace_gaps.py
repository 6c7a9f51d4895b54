"""Digest of a rocprofv3 --kernel-trace CSV of scripts/step_time.py: for the timed steps, the sum of kernel durations, the
span they cover and what lies between kernels of one queue.  python scripts/trace_gaps.py <kernel_trace.csv> [steps]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows))
# the timed steps are the last `steps` sgd pairs: cut at sgd_kernel launches
sgd = [i for i, k in enumerate(ks) if "sgd_kernel" in k[2]]
assert len(sgd) >= 2 * steps + 2, len(sgd)
lo, hi = sgd[-2 * steps - 1] + 1, sgd[-1] + 1
sel = ks[lo:hi]
span = (sel[-1][1] - sel[0][0]) / 1e6
busy = sum(e - s for s, e, _, _ in sel) / 1e6
# union of busy intervals (kernels of different queues overlap)
cur_s, cur_e, union = sel[0][0], sel[0][1], 0
for s, e, _, _ in sel[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
fam = collections.Counter()
for s, e, n, _ in sel:
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    fam[n.split("(")[0][:60]] += e - s
print("kernels %d over %d steps: span %.3f ms/step, sum of durations %.3f ms/step, union (GPU not idle) %.3f ms/step, idle %.3f ms/step"
      % (len(sel), steps, span / steps, busy / steps, union / 1e6 / steps, (span - union / 1e6) / steps))
gaps = []
byq = collections.defaultdict(list)
for k in sel:
    byq[k[3]].append(k)
for q, lst in byq.items():
    g = [max(0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1)]
    print("queue %s: %d kernels, gaps: sum %.3f ms/step, median %.2f us, p90 %.2f us, max %.1f us"
          % (q, len(lst), sum(g) / 1e6 / steps, sorted(g)[len(g) // 2] / 1e3, sorted(g)[int(len(g) * .9)] / 1e3, max(g) / 1e3))
for n, t in fam.most_common(12):
    print("  %-60s %.3f ms/step" % (n, t / 1e6 / steps))
