"""Reads a rocprofv3 --kernel-trace CSV and reports dispatches of ONE queue / stream whose execution overlaps the
previous dispatch of the same queue / stream (in-order streams must never show any)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
print("%d dispatches, columns: %s" % (len(rows), list(rows[0].keys())))
for keyname in ("Queue_Id", "Stream_Id"):
    if keyname not in rows[0]:
        continue
    by = defaultdict(list)
    for r in rows:
        by[r[keyname]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], int(r.get("Dispatch_Id", 0))))
    for k, v in sorted(by.items()):
        v.sort(key=lambda t: t[3])
        n_ov = 0
        for a, b in zip(v, v[1:]):
            if b[0] < a[1]:
                n_ov += 1
                if n_ov <= 5:
                    print("  %s %s: dispatch %d (%s) starts %d ns BEFORE dispatch %d (%s) ends" % (keyname, k, b[3], b[2], a[1] - b[0], a[3], a[2]))
        print("%s %s: %d dispatches, %d overlap their predecessor" % (keyname, k, len(v), n_ov))
