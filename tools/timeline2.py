"""Per-queue timeline of a window of a rocprofv3 kernel-trace CSV.
usage: timeline2.py <kernel_trace.csv> [n_last_ms]   -- prints the kernels of the last n ms (default 40)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
span = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
tend = int(rows[-1]["End_Timestamp"])
rows = [r for r in rows if int(r["Start_Timestamp"]) >= tend - span * 1e6]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("zg::", "").replace("void ", "")
    name = name.split("(")[0][:40]
    grid = int(r.get("Grid_Size", 0) or 0) // max(int(r.get("Workgroup_Size", 1) or 1), 1)
    print("%9.3f %9.3f %8.3f  q%-3s wg%-7d %s" % (s / 1e6, e / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?"), grid, name))
