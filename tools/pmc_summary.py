"""Per-kernel averages of a rocprofv3 --pmc pass (counter_collection.csv): counter sums per launch for every kernel
whose name contains one of the given substrings.
usage: pmc_summary.py <counter_collection.csv> <out.json> [substr ...]"""
import csv
import json
import sys


def main():
    by_grid = "--by-grid" in sys.argv
    argv = [a for a in sys.argv if a != "--by-grid"]
    path, outp, pats = argv[1], argv[2], argv[3:] or ["k_accumulate", "k_ntt_pass", "bench"]
    acc = {}
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not any(p in k for p in pats):
            continue
        name = k.replace("zg::", "").replace("void ", "").split("(")[0][:70]
        if by_grid:
            name += " grid=%s" % r.get("Grid_Size", "?")
        a = acc.setdefault(name, {"launches": 0})
        did = (r.get("Dispatch_Id"), name)
        if did not in seen:
            seen.add(did)
            a["launches"] += 1
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = {}
    for name, a in acc.items():
        n = max(a["launches"], 1)
        out[name] = {"launches": a["launches"], **{c: v / n for c, v in a.items() if c != "launches"}}
    json.dump(out, open(outp, "w"), indent=1)
    for name, a in out.items():
        print(name, json.dumps(a))


if __name__ == "__main__":
    main()
