"""timeline of the kernels of the LAST bake in a rocprofv3 --kernel-trace --output-format csv run (start / duration in ms relative to the bake's first kernel)
usage: python profiles/scripts/r03_timeline.py <dir with *_kernel_trace.csv> [max lines] [bake: -1 = last, -2 = the one before ...]"""
import csv, glob, sys, re
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
# the last bake starts at the last setup_fetch
starts = [i for i, r in enumerate(rows) if "setup_fetch" in r[2]]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
rows = rows[starts[which]:(starts[which + 1] if which != -1 else len(rows))]
t0 = rows[0][0]
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("ommx::", "")
    n = re.sub(r"rocprim::.*?(\w+)(<|$).*", r"rocprim \1", n)
    return n[:60]
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 400
prev_end = None
for s, e, n, q in rows[:lim]:
    print("%9.3f  +%8.3f  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, short(n)))
