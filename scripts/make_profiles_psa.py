"""Digest gpurun_out/<tag>_psa_* / <tag>_psamask_* (scripts/run_profiles_psa.sh) into profiles/:
  <tag>_psanet_kernel_stats.csv, <tag>_psanet_bench.json, <tag>_psamask.json (per psamask kernel: launches, average
  duration from the --stats pass, HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes, GB/s)."""
import csv, collections, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
g = os.path.join(ROOT, "gpurun_out")
shutil.copy(os.path.join(g, tag + "_psa_stats.kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_psanet_kernel_stats.csv"))
shutil.copy(os.path.join(g, "bench_%s_psa.json" % tag), os.path.join(ROOT, "profiles", tag + "_psanet_bench.json"))
def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'(void )?([\w:<>, ]+?)\(', n)
    return (m.group(2) if m else n).strip()
def pmc(path, ctr):
    a = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != ctr: continue
        k = short(r['Kernel_Name'])
        a[k][0] += float(r['Counter_Value']); a[k][1].add(r['Dispatch_Id'])
    return {k: (v[0], len(v[1])) for k, v in a.items()}
f = pmc(os.path.join(g, tag + "_psamask_fetch.counters.csv"), "FETCH_SIZE")
w = pmc(os.path.join(g, tag + "_psamask_write.counters.csv"), "WRITE_SIZE")
dur = {}
for r in csv.DictReader(open(os.path.join(g, tag + "_psamask_stats.kernel_stats.csv"))):
    dur[short(r['Name'] + "(")] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
res = {}
for k in sorted(dur):
    if "psamask" not in k: continue
    n, us = dur[k]
    fb = f.get(k, (0, 1)); wb = w.get(k, (0, 1))
    fetch = fb[0] * 1024 * 2 / max(fb[1], 1); write = wb[0] * 1024 / max(wb[1], 1)
    res[k] = {"launches": n, "avg_us_all_shapes": round(us, 2), "hbm_fetch_bytes_per_launch": round(fetch),
              "hbm_write_bytes_per_launch": round(write), "hbm_TBps": round((fetch + write) / us / 1e6, 3)}
# per shape: the three shapes of the script launch different grids, so (kernel, Grid_Size) separates them in the PMC and
# trace files; durations from the kernel trace of the --stats pass
def per_grid(path, ctr):
    a = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != ctr: continue
        k = (short(r['Kernel_Name']), int(r.get('Grid_Size', r.get('Grid_Size_X', 0))))
        a[k][0] += float(r['Counter_Value']); a[k][1].add(r['Dispatch_Id'])
    return {k: v[0] / max(len(v[1]), 1) for k, v in a.items()}
shape_rows = {}
try:
    fg = per_grid(os.path.join(g, tag + "_psamask_fetch.counters.csv"), "FETCH_SIZE")
    wg = per_grid(os.path.join(g, tag + "_psamask_write.counters.csv"), "WRITE_SIZE")
    tr = None
    for root, _, files in os.walk(os.path.join(g, tag + "_psamask_stats")):
        for fn in files:
            if fn.endswith("kernel_trace.csv"): tr = os.path.join(root, fn)
    durs = collections.defaultdict(list)
    for r in csv.DictReader(open(tr)):
        gs = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)))
        durs[(short(r['Kernel_Name']), gs)].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    for k in sorted(fg):
        if "psamask_plane" not in k[0] and "psamask_kernel" not in k[0]: continue
        d = sorted(durs.get(k, [0.0])); us = d[len(d) // 2] / 1e3
        fetch, write = fg[k] * 1024 * 2, wg.get(k, 0.0) * 1024
        shape_rows["%s grid %d" % k] = {"median_us": round(us, 1), "hbm_fetch_MB": round(fetch / 1e6, 1),
                                        "hbm_write_MB": round(write / 1e6, 1),
                                        "hbm_TBps_moved": round((fetch + write) / max(us, 1e-9) / 1e6, 2)}
except Exception as e:
    shape_rows = {"error": repr(e)}
json.dump({"command": "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- python scripts/psamask_bench.py (three separate passes)",
           "per_kernel_and_grid (grid 245760 = N16 30x30/59x59, 368640 = N16 45x45/89x89, 122880 = N2 30x30/59x59 for the plane-group kernel)": shape_rows,
           "corrections": "FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md HBM section); averages run over the three shapes of the script (N16 30x30/59x59, N16 45x45/89x89, N2 30x30/59x59), both psa types",
           "per_shape_timing": open(os.path.join(g, tag + "_psamask_bench.log")).read().splitlines()[-25:],
           "kernels": res}, open(os.path.join(ROOT, "profiles", tag + "_psamask.json"), "w"), indent=1)
for k, v in res.items(): print(k, v)
print(open(os.path.join(ROOT, "profiles", tag + "_psanet_bench.json")).read()[:600])
