"""Sum rocprofv3 --pmc counters per kernel template: python scripts/pmc_digest.py <counter_collection.csv>"""
import csv, collections, re, sys
def short(n):
    n = n.replace('(anonymous namespace)::', '')
    m = re.match(r'(void )?([\w:<>, ]+?)\(', n)
    return (m.group(2) if m else n).strip()
a = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r['Kernel_Name']); a[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
names = sorted({c for v in a.values() for c in v})
print("%-52s %5s " % ("kernel", "n") + " ".join("%14s" % c[-14:] for c in names))
for k, v in sorted(a.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    print("%-52s %5d " % (k[:52], len(n[k])) + " ".join("%14.4g" % (v.get(c, 0) / len(n[k])) for c in names))
    if 'SQ_WAVE_CYCLES' in v:
        wc = v['SQ_WAVE_CYCLES']
        print("%-52s       " % "   (fraction of SQ_WAVE_CYCLES)" + " ".join("%14.3f" % (v.get(c, 0) / wc) for c in names))
