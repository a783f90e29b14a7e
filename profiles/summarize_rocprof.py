#!/usr/bin/env python3
"""Condenses a rocprofv3 `--kernel-trace --stats` results .db (rocpd sqlite) into a short per-kernel table.
usage: python profiles/summarize_rocprof.py gpurun_out/prof/x_results.db > profiles/rNN_name.md"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.search(r"rocprim::(wrapped_)?(\w+?)(_config)?<", name)
    if name.startswith("void rocprim") and m:
        return "rocprim " + m.group(2)
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    agg = {}
    for name, calls, total, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls; a[1] += total; a[2] += pct
    print("| kernel | calls | total ms | avg us | % of GPU time |")
    print("|---|---|---|---|---|")
    for k, (calls, total, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("| `%s` | %d | %.3f | %.1f | %.2f |" % (k, calls, total / 1e3, total / calls, pct))


if __name__ == "__main__":
    main(sys.argv[1])
