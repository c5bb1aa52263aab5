"""Who issues the __amd_rocclr_copyBuffer dispatches of a bench run, and are they on the critical path?

    cd /tmp && rocprofv3 --kernel-trace -d <dir> -o t --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --secondary none --no-cpu-baseline --no-roofline
    python scripts/debug/copybuffer_trace.py <dir>

Reads the kernel trace (start / end time stamp of every dispatch), sorts it by start time and prints, for the dispatches whose name
contains `copyBuffer` (or argv[2]): how many, their durations, which kernels run right before and right after them, the idle time
of the device around them, and how many fall between two plan runs (before a `range_clear_kernel` / `prep_*` kernel) against
inside one."""
import csv
import glob
import os
import sys
from collections import Counter


def main():
    d = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no *kernel_trace.csv under %s" % d
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = {a.lower(): b for a, b in r.items()}
                rows.append((int(k["start_timestamp"]), int(k["end_timestamp"]), k["kernel_name"]))
    rows.sort()
    short = lambda n: n.split("(")[0].split("<")[0][-40:]
    print("%d dispatches, %d matching '%s'" % (len(rows), sum(pat in r[2] for r in rows), pat))
    before, after = Counter(), Counter()
    dur = gap_b = gap_a = 0.0
    n = 0
    runs = []
    for i, (s, e, name) in enumerate(rows):
        if pat not in name:
            continue
        n += 1
        dur += (e - s) / 1e3
        if i > 0:
            before[short(rows[i - 1][2])] += 1
            gap_b += max(0, s - rows[i - 1][1]) / 1e3
        if i + 1 < len(rows):
            after[short(rows[i + 1][2])] += 1
            gap_a += max(0, rows[i + 1][0] - e) / 1e3
        runs.append(i)
    if not n:
        return
    print("total %.1f us in the copies themselves (%.2f us each); idle before them %.1f us, after them %.1f us (sum over all)" % (dur, dur / n, gap_b, gap_a))
    print("kernel right BEFORE:", before.most_common(8))
    print("kernel right AFTER :", after.most_common(8))
    # consecutive copies form groups: how long is a group, how many groups
    groups, cur = [], [runs[0]]
    for a, b in zip(runs, runs[1:]):
        if b == a + 1:
            cur.append(b)
        else:
            groups.append(cur); cur = [b]
    groups.append(cur)
    sizes = Counter(len(g) for g in groups)
    span = sum((rows[g[-1]][1] - rows[g[0]][0]) / 1e3 for g in groups)
    print("%d groups of consecutive copies, sizes %s; wall time covered by the groups %.1f us" % (len(groups), sorted(sizes.items()), span))
    total = (rows[-1][1] - rows[0][0]) / 1e3
    busy = sum((e - s) for s, e, _ in rows) / 1e3
    print("trace spans %.1f ms, sum of dispatch durations %.1f ms" % (total / 1e3, busy / 1e3))


if __name__ == "__main__":
    main()
