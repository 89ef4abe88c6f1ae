#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite database (rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd) into the
kernel-stats CSV committed under profiles/ (same columns as rocprofv3's kernel_stats.csv).
usage: rocpd_stats.py results.db out.csv [name-filter]"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    flt = sys.argv[3] if len(sys.argv) > 3 else None
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name").fetchall()
    total = sum(r[2] for r in rows)
    rows.sort(key=lambda r: -r[2])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPRs", "SGPRs", "LDSBytes",
                    "GridX", "WorkgroupX"])
        for r in rows:
            name = r[0].split("(")[0]
            if flt and flt not in name:
                continue
            w.writerow([name, r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / total, 3)] + list(r[6:]))


if __name__ == "__main__":
    main()
