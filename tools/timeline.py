"""Compact per-stream timeline of the LAST prove step from a rocprofv3 kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: find the last k_r1cs_eval group
idx = [i for i, r in enumerate(rows) if "k_r1cs_eval" in r["Kernel_Name"]]
start_i = idx[-3] if len(idx) >= 3 else 0
rows = rows[start_i:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("zg::", "").replace("void ", "")[:44]
    print("%9.3f %9.3f %8.3f  q%-3s %s" % (s / 1e6, e / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?"), name))
