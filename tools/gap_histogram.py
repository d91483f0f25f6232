#!/usr/bin/env python3
"""Idle time between consecutive kernels of one bench step, from a rocprofv3 --kernel-trace CSV.
usage: python3 tools/gap_histogram.py <dir or *_kernel_trace.csv> [out.txt]
Kernels are taken in start order (the engine and its evaluator share one stream); gap = next start - this end.  Printed:
the histogram of the gaps, their sum against the span of the step, and the sum by which kernel FOLLOWS the gap."""
import csv, glob, os, sys
from collections import defaultdict

def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].split("<")[0][:28]

def main():
    src = sys.argv[1]
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(src)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # the step proper: from the first k_search_round to the last k_play_move
    first = next(i for i, r in enumerate(rows) if r[2].startswith("k_search_round"))
    last = max(i for i, r in enumerate(rows) if r[2].startswith("k_play_move"))
    rows = rows[first:last + 1]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    edges = [0, 2, 4, 6, 8, 10, 15, 20, 50, 100, 1000, 10 ** 9]
    hist = [0] * (len(edges) - 1); hsum = [0] * (len(edges) - 1)
    by = defaultdict(lambda: [0, 0])
    tot = 0
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = max(0, s1 - e0) / 1e3
        tot += g
        for k in range(len(edges) - 1):
            if edges[k] <= g < edges[k + 1]:
                hist[k] += 1; hsum[k] += g
                break
        by[n0 + " -> " + n1][0] += 1; by[n0 + " -> " + n1][1] += g
    out = []
    out.append("%s: %d kernels, span %.2f ms, kernels busy %.2f ms (%.2f %%), gaps %.2f ms (%.2f %% of the span)" % (
        os.path.basename(src), len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e3, 100.0 * tot * 1e3 / span))
    out.append("gap [us)        count     sum ms")
    for k in range(len(edges) - 1):
        out.append("%5d - %-7s %6d  %9.3f" % (edges[k], "inf" if edges[k + 1] >= 10 ** 9 else edges[k + 1], hist[k], hsum[k] / 1e3))
    out.append("by pair (previous -> next)                                   count   sum ms   avg us")
    for key, (n, g) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
        out.append("%-60s %6d %8.3f %8.2f" % (key, n, g / 1e3, g / n))
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")

if __name__ == "__main__":
    main()
