"""Turn a `rocprofv3 --kernel-trace --stats` kernel_stats CSV into the markdown table committed beside it,
with the per-launch HIP-event averages of a bench line (roofline.stages) next to it when given.

    python profiles/summarize.py profiles/r01_kernel_stats.csv [profiles/r01_bench.json]
"""
import csv
import json
import re
import sys


def short(name):
    m = re.match(r"void kernel_entry<(.*)>\(", name)
    return m.group(1).strip() if m else name


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    print("| kernel | calls | avg us | total % |")
    print("|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    if len(sys.argv) > 2:
        line = json.load(open(sys.argv[2]))
        # bench.py's launch names leave trailing default template arguments out
        avg = {re.sub(r"(, false)+>$", ">", short(r["Name"])): (float(r["AverageNs"]) / 1e3) for r in rows}
        print()
        print("| stage (bench.py, HIP events) | levels | launches | us (events) | us (rocprof avg x launches) | "
              "algorithmic MB | GB/s | frac of 8 TB/s |")
        print("|---|---|---|---|---|---|---|---|")
        for s in line["roofline"]["stages"]:
            rp = sum(avg.get(k, float("nan")) * c for k, c in s["launches"].items())
            lv = s["levels"] or ["-", "-"]
            print("| %s | %s-%s | %s | %.1f | %.1f | %.1f | %.0f | %.3f |" % (
                s["stage"], lv[0], lv[1],
                ", ".join("%s x%d" % kc for kc in s["launches"].items()), s["us"], rp,
                s["algorithmic_bytes"] / 1e6, s.get("model_GB/s", s.get("GB/s")), s.get("model_frac", s.get("frac"))))


if __name__ == "__main__":
    main()
