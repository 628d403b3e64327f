"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into
per-launch HBM traffic of the dominant kernel.  Counters are in KiB.  /opt/skills/guides/MI355X_MICROARCH.md (HBM
section): FETCH_SIZE reports exactly half the bytes of a wide COALESCED streaming read and is uncalibrated for other
patterns.  k_accumulate's traffic is one 64-byte random gather per lane, so the factor was calibrated on exactly that
pattern with a known byte count (tools/ubench_gather.hip, profiles/r02_pmc_calibration.json): FETCH_SIZE * 1024 =
1.02 x the gathered bytes (factor 1, not 2; the streaming kernel of the same run confirms the guide's 2.0), and
WRITE_SIZE * 1024 = the written bytes exactly.
usage: pmc_traffic.py <fetch counter csv> <write counter csv> <out json> [fetch factor, default 1.0]"""
import csv
import json
import sys


def per_kernel(path, counter, by_grid=False):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if by_grid:
            k = (k, int(r.get("Grid_Size", 0) or 0))
        a = acc.setdefault(k, [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    ff = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    out = {"note": "bytes per launch; FETCH_SIZE*1024*%.2f (factor calibrated on 64-byte random gathers, see "
                   "r02_pmc_calibration.json) + WRITE_SIZE*1024" % ff,
           "fetch_factor": ff, "kernels": {}}
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        f, nf = fetch[k]
        w, nw = write.get(k, [0.0, 1])
        out["kernels"][k[:90]] = {"launches": nf, "fetch_bytes_per_launch": ff * 1024 * f / nf,
                                  "write_bytes_per_launch": 1024 * w / max(nw, 1)}
    acc = [k for k in fetch if "k_accumulate" in k and "FqParams" in k]
    if acc:
        k = acc[0]
        f, nf = fetch[k]
        w, nw = write.get(k, [0.0, 1])
        out["msm_accumulate_g1_bytes_per_launch"] = ff * 1024 * f / nf + 1024 * w / max(nw, 1)
        out["msm_accumulate_g1_launches"] = nf
    # the h multiexp's launches of k_accumulate<Fq> are the ones with the largest grid (every point reaches the kernel)
    fg, wg = per_kernel(sys.argv[1], "FETCH_SIZE", True), per_kernel(sys.argv[2], "WRITE_SIZE", True)
    hk = [k for k in fg if "k_accumulate" in k[0] and "FqParams" in k[0]]
    if hk:
        top = max(g for _, g in hk)
        big = [k for k in hk if k[1] >= 0.98 * top]
        f = sum(fg[k][0] for k in big)
        nf = sum(fg[k][1] for k in big)
        w = sum(wg.get(k, [0.0, 0])[0] for k in big)
        nw = max(sum(wg.get(k, [0.0, 0])[1] for k in big), 1)
        out["msm_accumulate_g1_dense_bytes_per_launch"] = ff * 1024 * f / nf + 1024 * w / nw
        out["msm_accumulate_g1_dense_launches"] = nf
    # the H pipeline (all transform kernels) per proof: one k_ntt_tail3 launch per proof in evaluation form
    ntt = [k for k in fetch if "k_ntt" in k]
    proofs = max([fetch[k][1] for k in ntt if "tail3" in k] or [0])
    if ntt and proofs:
        tot = sum(ff * 1024 * fetch[k][0] + 1024 * write.get(k, [0.0, 0])[0] for k in ntt)
        out["h_pipeline_bytes_per_proof"] = tot / proofs
        out["h_pipeline_launches_per_proof"] = sum(fetch[k][1] for k in ntt) / proofs
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
    for k, v in list(out["kernels"].items())[:8]:
        print("%-90s %s" % (k, v))


if __name__ == "__main__":
    main()
