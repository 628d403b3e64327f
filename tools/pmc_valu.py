"""VALU instructions per wave-level mixed addition of k_accumulate<Fq>, from a rocprofv3 --pmc SQ_INSTS_VALU pass.
The h multiexp's accumulation launches are the ones with the largest grid; each adds `adds_per_launch` points (lane
level), i.e. adds_per_launch / 64 wave-level additions.
usage: pmc_valu.py <counter_collection.csv> <out.json> <adds_per_launch>"""
import csv
import json
import sys


def main():
    path, outp, adds = sys.argv[1], sys.argv[2], float(sys.argv[3])
    by_grid = {}
    for r in csv.DictReader(open(path)):
        if "k_accumulate<zg::Fp" not in r["Kernel_Name"] or r["Counter_Name"] != "SQ_INSTS_VALU":
            continue
        g = int(r["Grid_Size"])
        a = by_grid.setdefault(g, [0.0, set()])
        a[0] += float(r["Counter_Value"])
        a[1].add(r["Dispatch_Id"])
    g = max(by_grid)
    total, launches = by_grid[g][0], len(by_grid[g][1])
    per_launch = total / launches
    out = {"note": "rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -- python3 bench.py ...; the %d launches of k_accumulate<Fq> "
                   "with the largest grid (%d threads: the parts of the h multiexp), %.0f mixed additions each" % (launches, g, adds),
           "sq_insts_valu_per_launch": per_launch, "mixed_adds_per_launch": adds,
           "k_accumulate_g1_valu_instructions_per_wave_mixed_add": per_launch / (adds / 64.0)}
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
