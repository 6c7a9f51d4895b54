"""Process-by-process A/B of prebuilt kernel libraries and environment switches on the whole train step, interleaved over
ROUNDS rounds.   python scripts/ab_libs.py out.json [batch] [rounds] name[:lib.so][:KEY=VAL+KEY=VAL] ...
Each configuration runs scripts/step_time.py in a fresh process (SEMSEG_HIP_LIB = the library; 'default' = the in-tree one)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path, batch, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3])
cfgs = []
for c in sys.argv[4:]:
    f = c.split(":")
    cfgs.append((f[0], f[1] if len(f) > 1 and f[1] else "default", f[2] if len(f) > 2 else ""))
res = {c[0]: {"lib": c[1], "env": c[2], "ms": [], "families": None} for c in cfgs}
for r in range(rounds):
    for name, lib, env in cfgs:
        e = dict(os.environ)
        if lib != "default":
            e["SEMSEG_HIP_LIB"] = os.path.join(ROOT, lib)
        for kv in [x for x in env.split("+") if x]:
            e[kv.split("=")[0]] = kv.split("=", 1)[1]
        e["FAMILIES"] = "1" if r == 0 else "0"
        args = e.pop("STEP_ARGS", "").split()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "step_time.py"), batch] + args, env=e,
                           capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            res[name].setdefault("errors", []).append((p.stderr or "")[-400:])
            continue
        d = json.loads(line[-1])
        res[name]["ms"].append(d["ms"])
        res[name]["loss"] = d["loss"]
        if "families" in d:
            res[name]["families"], res[name]["serial_sum_ms"] = d["families"], d["serial_sum_ms"]
    json.dump(res, open(out_path, "w"), indent=1)
for name in res:
    print("%-14s %s  min %s" % (name, res[name]["ms"], min(res[name]["ms"]) if res[name]["ms"] else None), flush=True)
