"""One train step of a rocprofv3 --kernel-trace CSV of scripts/step_time.py as a text timeline: per launch the queue, start offset in
the step, duration and the gap to the previous launch of the same queue; then per queue and per kernel name the sums.
python scripts/timeline.py <kernel_trace.csv> <out.txt> [step index from the end, default 2]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows))
sgd = [i for i, k in enumerate(ks) if "sgd_kernel" in k[2]]
# a step ends with its two sgd launches (base + head parameter groups): the step `back` from the end
hi = sgd[-1 - 2 * (back - 1)] + 1
lo = sgd[-1 - 2 * back] + 1
sel = ks[lo:hi]
t0 = sel[0][0]


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]


last = {}
out = open(sys.argv[2], "w")
out.write("step of %d launches, span %.3f ms\n" % (len(sel), (sel[-1][1] - t0) / 1e6))
qs = sorted({k[3] for k in sel})
busy, gaps, n = collections.Counter(), collections.Counter(), collections.Counter()
fam = collections.defaultdict(lambda: [0, 0, 0])
for s, e, name, q in sel:
    g = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    busy[q] += e - s
    gaps[q] += max(0.0, g)
    n[q] += 1
    f = fam[(q, short(name))]
    f[0] += 1
    f[1] += e - s
    f[2] += max(0.0, g)
    out.write("q%-2d %9.1f us  dur %7.1f  gap %7.1f  %s\n" % (qs.index(q), (s - t0) / 1e3, (e - s) / 1e3, g, short(name)))
for q in qs:
    out.write("queue %d: %d launches, busy %.3f ms, gaps %.3f ms\n" % (qs.index(q), n[q], busy[q] / 1e6, gaps[q] / 1e6))
out.write("per queue and kernel: launches, busy ms, gap-in-front ms\n")
for (q, name), (c, b, g) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2] * 1e3)):
    out.write("q%-2d %-70s %4d  %7.3f  %7.3f\n" % (qs.index(q), name, c, b / 1e6, g / 1e3))
out.close()
print(open(sys.argv[2]).read().split("per queue and kernel")[0][-400:])
