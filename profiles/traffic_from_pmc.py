"""HBM bytes of the tree launches of ONE step from the PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
rocprofv3 --pmc runs of bench.py, tests/gpu_debug/pmc_passes.sh), merged into profiles/tree_traffic.json under the
workload key bench.py looks up.

    python profiles/traffic_from_pmc.py gpurun_out/pmc_TAG  [more dirs ...]

Per kernel: average over its dispatches of 2 x FETCH_SIZE (the gfx950 correction of MI355X_MICROARCH.md, HBM section:
the counter tallies 128-byte requests at 64 bytes) + WRITE_SIZE, counter unit KiB; times the number of launches of that
kernel in one step (roofline.stages[*].launches of the bench line printed under the profiler, p3.json)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.match(r"void kernel_entry<(.*)>\(", name)
    k = m.group(1).strip() if m else name[:40]
    return re.sub(r"(, false)+>$", ">", k)   # bench.py's launch names leave trailing default arguments out


def avg_counter(d, counter):
    per = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[(r["Dispatch_Id"], short(r["Kernel_Name"]))] += float(r["Counter_Value"])
    acc = collections.defaultdict(list)
    for (_, k), v in per.items():
        acc[k].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    path = os.path.join(ROOT, "profiles", "tree_traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    for d in sys.argv[1:]:
        line = json.loads(open(os.path.join(d, "p3.json")).read().strip().splitlines()[-1])
        wl = line["config"]["workload"]
        fetch, write = avg_counter(d, "FETCH_SIZE"), avg_counter(d, "WRITE_SIZE")
        total, chirp, per_kernel = 0.0, 0.0, {}
        for st in line["roofline"]["stages"]:
            for k, cnt in st["launches"].items():
                b = (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0
                per_kernel[k] = {"launches": cnt, "fetch_MB": round(2 * fetch.get(k, 0.0) * 1024 / 1e6, 1),
                                 "write_MB": round(write.get(k, 0.0) * 1024 / 1e6, 1)}
                if st.get("levels") is None and st["stage"].startswith("chirp"):
                    chirp += b * cnt      # the evaluation stage is not part of roofline.traffic (the tree's launches)
                else:
                    total += b * cnt
        m = re.search(r"D=M=2\^(\d+).*?, (\w+), (\d+) signal", wl)
        wk = "cfg5" if "kdvv" in wl else ("cfg3" if int(m.group(3)) > 1 else "cfg2")
        key = "%s/D=2^%s/%s/B=%s" % (wk, m.group(1), m.group(2), m.group(3))
        table[key] = {"tree_hbm_bytes_per_step": int(total), "algorithmic_bytes": line["roofline"]["algorithmic_bytes"],
                      "tree_ms_under_profiler": line["roofline"]["tree_ms"], "per_kernel": per_kernel, "source": os.path.basename(d),
                      "build_id": line.get("build_id"), "chirp_hbm_bytes_per_step": int(chirp)}
        print(key, "%.3f GB per step" % (total / 1e9))
    json.dump(table, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
