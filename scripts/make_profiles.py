"""Digest gpurun_out/r01_{stats,fetch,write,mfma} (rocprofv3 CSVs of bench.py) into profiles/."""
import csv, collections, json, re, shutil, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'(void )?([\w:<>, ]+?)\(', n)
    return (m.group(2) if m else n).strip()
def family(k):
    """Template instantiations that bench.py's KernelTimer reports as one family: the weight-gradient kernel's
    addressing MODE and the forward/data-gradient kernel's two-level-accumulation flag are dropped; the SP = 3
    (SEMSEG_ARITH_BF16X3) instances keep an SP3 tag, as in the KernelTimer labels."""
    if k.startswith('conv_wgrad_dma_wide_kernel<'): return 'conv_wgrad_dma_wide_kernel<128x256,SP3>'
    m = re.match(r'conv_wgrad_dma_kernel<\d+, \d+, \d+, \d+, (?:true|false)(?:, (\d))?>', k)
    if m: return 'conv_wgrad_dma_kernel<128x128%s>' % (',SP3' if m.group(1) == '3' else '')
    m = re.match(r'conv_wgrad_kernel<(\d+), (\d+), \d+(?:, (\d))?>', k)
    if m: return 'conv_wgrad_kernel<%s,%s%s>' % (m.group(1), m.group(2), ',SP3' if m.group(3) == '3' else '')
    m = re.match(r'conv_igemm_kernel<(\d+), (\d+), (true|false), (\d+), (?:true|false)(?:, (\d))?>', k)
    if m: return 'conv_igemm_kernel<%s,%s,%s,%s%s>' % (m.group(1), m.group(2), m.group(3), m.group(4), ',SP3' if m.group(5) == '3' else '')
    m = re.match(r'gemm_rows_bf16split_kernel<(\d+), (\d+)(?:, (\d+))?(?:, (\d+))?>', k)      # third parameter: epilogue (0 plain, 1 statistics, 2 data gradient); fourth: rows per tile (256 | 128)
    if m:
        epi = '' if m.group(3) in (None, '0') else ',' + m.group(3)
        bm = ',128' if m.group(4) == '128' else ''
        if bm and not epi: epi = ',0'
        return 'gemm_rows_bf16split_kernel<%s,%s%s%s>' % (m.group(1), m.group(2), epi, bm)
    return k
def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        k = family(short(r['Kernel_Name']))
        a[k][r['Counter_Name']] += float(r['Counter_Value'])
        seen[k][r['Dispatch_Id']] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    return {k: dict(c, launches=len(seen[k]), dur_ns=sum(seen[k].values())) for k, c in a.items()}
g = os.path.join(ROOT, "gpurun_out")
f = agg(os.path.join(g, tag + "_fetch", "pmc_counter_collection.csv"))
w = agg(os.path.join(g, tag + "_write", "pmc_counter_collection.csv"))
m = agg(os.path.join(g, tag + "_mfma", "pmc_counter_collection.csv"))
res = {}
for k in f:
    if not any(t in k for t in ("conv_igemm", "conv_wgrad", "gemm_rows", "bn_", "splitk", "wino_")): continue
    n = f[k]['launches']
    fetch = f[k]['FETCH_SIZE'] * 1024 * 2 / n
    write = w.get(k, {}).get('WRITE_SIZE', 0) * 1024 / max(w.get(k, {}).get('launches', 1), 1)
    mm = m.get(k, {}); util = clk = None
    if mm.get('GRBM_GUI_ACTIVE'):
        util = mm.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (mm['GRBM_GUI_ACTIVE'] / 8 * 1024)
        # GRBM_GUI_ACTIVE / wall time is a clock only where the kernel IS the wall time: for launches under 20 us the counter
        # includes the command processor's start / drain around the dispatch and the quotient comes out at 5-7 "GHz" (round 4's
        # file) — masked
        clk = mm['GRBM_GUI_ACTIVE'] / 8 / mm['dur_ns'] if mm['dur_ns'] / max(mm['launches'], 1) >= 20e3 else None
    res[k] = {"launches_in_the_pmc_run": n, "hbm_fetch_bytes_per_launch": round(fetch), "hbm_write_bytes_per_launch": round(write),
              "hbm_bytes_per_launch": round(fetch + write), "mfma_busy_frac": None if util is None else round(util, 4),
              "clock_ghz": None if clk is None else round(clk, 3)}
_line = json.load(open(os.path.join(g, "bench_%s_n1.json" % tag)))
json.dump({"workload": _line["config"]["workload"], "arch": "gfx950",
           "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE> (three separate passes) --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing",
           "corrections": "FETCH_SIZE and WRITE_SIZE are KB; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)",
           "kernels": res}, open(os.path.join(ROOT, "profiles", tag + "_pmc_per_kernel.json"), "w"), indent=1)
shutil.copy(os.path.join(g, tag + "_stats", "bench_kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_bench_kernel_stats.csv"))
shutil.copy(os.path.join(g, "bench_%s_n1.json" % tag), os.path.join(ROOT, "profiles", tag + "_bench_n1.json"))
for k, v in sorted(res.items(), key=lambda kv: -(kv[1]['mfma_busy_frac'] or 0))[:8]: print(k, v['mfma_busy_frac'], v['hbm_bytes_per_launch'])
# bench.py reads the PMC file that was committed when it ran; refresh the two PMC-derived roofline fields of
# the copied line from THIS round's passes (everything else in the line is as bench.py printed it)
bp = os.path.join(ROOT, "profiles", tag + "_bench_n1.json")
d = json.load(open(bp))
key = d["roofline"]["kernel"].split("+")[0].split("(")[0].replace(" ", "")
if key in res:
    d["roofline"]["traffic"] = res[key]["hbm_bytes_per_launch"]
    d["roofline"]["mfma_busy_frac_pmc"] = res[key]["mfma_busy_frac"]
    d["roofline"]["traffic_note"] = "traffic / mfma_busy_frac_pmc re-read from profiles/%s_pmc_per_kernel.json after the PMC passes of the same build" % tag
    json.dump(d, open(bp, "w"))
d = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_n1.json")))
print(d['value'], d['ms_per_step'], d.get('whole_step_frac_of_f32_mfma_peak')); print(d['roofline'])
for k, v in d['kernel_families'].items(): print(' ', k, v)
rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_bench_kernel_stats.csv"))))
for r in rows[:6]: print(short(r['Name']), r['Calls'], "avg us %.1f" % (float(r['AverageNs']) / 1e3))
# family-level durations of the same rocprofv3 --stats run, comparable with bench.py's kernel_families
fam = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = family(short(r['Name'] + "("))
    fam[k][0] += int(r['Calls']); fam[k][1] += float(r['TotalDurationNs'])
with open(os.path.join(ROOT, "profiles", tag + "_bench_family_stats.csv"), "w") as fo:
    fo.write("family,calls,total_ms,avg_us\n")
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        fo.write('"%s",%d,%.3f,%.2f\n' % (k, c, t / 1e6, t / c / 1e3))
# serialized run (SEMSEG_DEBUG=side_wgrad=0,hipri_main=0): the durations the bench line's roofline is quoted on
sp = os.path.join(g, tag + "_serial", "bench_kernel_stats.csv")
if os.path.exists(sp):
    shutil.copy(sp, os.path.join(ROOT, "profiles", tag + "_serial_kernel_stats.csv"))
    fs = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(sp)):
        k = family(short(r['Name'] + "("))
        fs[k][0] += int(r['Calls']); fs[k][1] += float(r['TotalDurationNs'])
    with open(os.path.join(ROOT, "profiles", tag + "_serial_family_stats.csv"), "w") as fo:
        fo.write("family,calls,total_ms,avg_us\n")
        for k, (c, t) in sorted(fs.items(), key=lambda kv: -kv[1][1]):
            fo.write('"%s",%d,%.3f,%.2f\n' % (k, c, t / 1e6, t / c / 1e3))
    w2 = fs.get('conv_wgrad_dma_wide_kernel<128x256,SP3>') or fs.get('conv_wgrad_dma_kernel<128x128,SP3>') or fs.get('conv_wgrad_dma_kernel<128x128>'); r2 = fs.get('wgrad_reduce_unpack_kernel')
    if w2: print("serialized rocprof: 128x128 weight-gradient kernel avg %.1f us over %d launches (+ reduce %.1f us)" % (w2[1] / w2[0] / 1e3, w2[0], r2[1] / r2[0] / 1e3 if r2 else 0))
wg = fam.get('conv_wgrad_dma_wide_kernel<128x256,SP3>') or fam.get('conv_wgrad_dma_kernel<128x128,SP3>') or fam.get('conv_wgrad_dma_kernel<128x128>'); ru = fam.get('wgrad_reduce_unpack_kernel')
if wg: print("rocprof family 128x128 weight-gradient kernel: avg %.1f us over %d launches (reduce kernel avg %.1f us)" % (wg[1] / wg[0] / 1e3, wg[0], ru[1] / ru[0] / 1e3 if ru else 0))
