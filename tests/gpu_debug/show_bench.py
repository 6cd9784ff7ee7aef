"""Print the per-stage / per-launch numbers of a bench.py JSON line. usage: show_bench.py file.json [...]"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e); continue
    r = d.get("roofline") or {}
    print("%s: value %.1f  ms/step %.4f  tree %.4f  chirp %.4f  frac %.3f" % (f, d["value"], d["ms_per_step"], r.get("tree_ms", 0), r.get("chirpz_epilogue_ms", 0), r.get("frac", 0)))
    for s in r.get("stages") or []:
        print("   %-34s %7.1f us  %s" % (s["stage"][:34], s["us"], s["launches"]))
    L = r.get("launches_us") or []
    print("   launches: " + "  ".join("%s %.1f" % (n.replace("KColBridge2", "Br2").replace("KMidSym", "Mid"), u) for n, u in L))
