"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into
per-launch HBM traffic of the dominant kernel, with the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md section HBM: counters are in KiB; FETCH_SIZE under-reports wide
(16 B/lane) coalesced reads by exactly 2x -- k_accumulate reads its bases with dwordx4 loads, so the
fetch side is doubled; WRITE_SIZE is uncalibrated and taken as is.
usage: pmc_traffic.py <fetch counter csv> <write counter csv> <out json>"""
import csv
import json
import sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        a = acc.setdefault(k, [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"note": "bytes per launch; FETCH_SIZE*1024*2 (gfx950 wide-read correction) + WRITE_SIZE*1024",
           "kernels": {}}
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        f, nf = fetch[k]
        w, nw = write.get(k, [0.0, 1])
        out["kernels"][k[:90]] = {"launches": nf, "fetch_bytes_per_launch": 2 * 1024 * f / nf,
                                  "write_bytes_per_launch": 1024 * w / max(nw, 1)}
    acc = [k for k in fetch if "k_accumulate" in k and "FqParams" in k]
    if acc:
        k = acc[0]
        f, nf = fetch[k]
        w, nw = write.get(k, [0.0, 1])
        out["msm_accumulate_g1_bytes_per_launch"] = 2 * 1024 * f / nf + 1024 * w / max(nw, 1)
        out["msm_accumulate_g1_launches"] = nf
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
    for k, v in list(out["kernels"].items())[:8]:
        print("%-90s %s" % (k, v))


if __name__ == "__main__":
    main()
