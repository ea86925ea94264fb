"""Largest idle gaps between consecutive kernels in a rocprofv3 --kernel-trace CSV (companion of tools/stall_hunt.py):
prints the N largest gaps with the kernels on either side, and per-kernel statistics (count, mean, max, max/mean) so that a
single slow launch of an otherwise fast kernel stands out.

usage: python tools/trace_gaps.py <dir-or-kernel_trace.csv> [N=15]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    if os.path.isdir(path):
        c = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        if not c:
            raise SystemExit("no *kernel_trace.csv under " + path)
        path = c[0]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
    rows.sort()
    print("%d kernels, span %.2f ms, busy %.2f ms" % (len(rows), (rows[-1][1] - rows[0][0]) / 1e6,
                                                     sum(e - s for s, e, _ in rows) / 1e6))
    gaps = []
    end = rows[0][1]
    prev = rows[0][2]
    for s, e, n in rows[1:]:
        if s > end:
            gaps.append((s - end, prev, n, s))
        if e > end:
            end, prev = e, n
    gaps.sort(reverse=True)
    print("largest idle gaps (no kernel running on the device):")
    for g, a, b, at in gaps[:top]:
        print("  %9.3f ms at t=%.3f ms  after %-50s before %s" % (g / 1e6, (at - rows[0][0]) / 1e6, a, b))
    stats = {}
    for s, e, n in rows:
        stats.setdefault(n, []).append(e - s)
    print("kernels whose slowest launch is far above their mean:")
    for n, v in sorted(stats.items(), key=lambda kv: -max(kv[1])):
        m = sum(v) / len(v)
        if max(v) > 3 * m and max(v) > 2e5:
            print("  %-60s n=%5d mean %8.1f us max %9.1f us" % (n, len(v), m / 1e3, max(v) / 1e3))


if __name__ == "__main__":
    main()
