#!/usr/bin/env python3
"""Turns rocprofv3 `--pmc ... --output-format csv` runs of bench.py into (a) a per-kernel HBM traffic table and
(b) profiles/<tag>_hbm_traffic.json, which bench.py reads to fill roofline.traffic for the matching workload.

usage: python profiles/summarize_pmc.py <tag> <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ [pmc_SQ/]> "<bench command line>"

Units and corrections (MI355X_MICROARCH.md, HBM section; calibrated here on kernels with a known byte count, sat_rows and
sat_cols of the 4096^2 texture): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports HALF the bytes read
(sat_rows reads 16.8 MB -> 8207.75 KiB reported; sat_cols reads 67.1 MB -> 32778 KiB) and is doubled below; WRITE_SIZE is
exact (sat_rows writes 67.1 MB -> 65536 KiB) and is used as is.  The two counters are collected in separate passes."""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    if "rocprim::" in name:
        m = re.search(r"rocprim::(?:trampoline_kernel<rocprim::)?(?:wrapped_)?(\w+)", name)
        return "rocprim " + (m.group(1) if m else "kernel")
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return re.sub(r"<.*", "", name)


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"],)
        if key not in seen:
            seen.add(key)
            dur[k][0] += 1; dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return agg, dur


def main():
    tag, root, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch, _ = load(os.path.join(root, "pmc_FETCH_SIZE", "pmc_counter_collection.csv"))
    write, _ = load(os.path.join(root, "pmc_WRITE_SIZE", "pmc_counter_collection.csv"))
    rows = []
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, {}).get("FETCH_SIZE", [1, 0.0]); w = write.get(k, {}).get("WRITE_SIZE", [1, 0.0])
        rd = 2.0 * f[1] / max(1, f[0]) * 1024.0      # bytes per launch, gfx950 x2 correction
        wr = w[1] / max(1, w[0]) * 1024.0
        rows.append((k, max(f[0], w[0]), rd, wr))
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    out = ["# %s -- HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % tag, "",
           "command: `%s`" % cmd, "", "FETCH_SIZE x2 (gfx950), WRITE_SIZE x1, KiB -> bytes; see the docstring of profiles/summarize_pmc.py for the calibration.", "",
           "| kernel | launches | read MB / launch | written MB / launch |", "|---|---|---|---|"]
    for k, n, rd, wr in rows[:24]:
        out.append("| `%s` | %d | %.2f | %.2f |" % (k, n, rd / 1e6, wr / 1e6))
    sq_path = os.path.join(root, "pmc_SQ", "pmc_counter_collection.csv")
    summary = {}
    if os.path.exists(sq_path):
        sq, _ = load(sq_path)
        c = {n: v[1] / max(1, v[0]) for n, v in sq.get("ommx::classify_tiles", {}).items()}
        if c:
            out += ["", "## `classify_tiles` SQ counters (per launch)", "", "| counter | value |", "|---|---|"]
            out += ["| %s | %.4g |" % (n, v) for n, v in sorted(c.items())]
            if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
                # GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_ACTIVE_INST_* count quad-cycles (one wave64 VALU instruction = 1) over 1024 SIMDs
                cap = 1024.0 * (c["GRBM_GUI_ACTIVE"] / 8.0) / 4.0
                summary["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] / cap
                out += ["", "VALU issue utilisation = SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE/8 / 4) = **%.2f**" % summary["valu_busy"]]
            if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
                summary["valu_lane_util"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
                out += ["", "VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = **%.2f**" % summary["valu_lane_util"]]
            if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
                out += ["", "SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.2f" % (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"])]
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), tag + "_pmc.md"), "w").write("\n".join(out) + "\n")
    ck = [r for r in rows if r[0] == "ommx::classify_tiles"]
    js = {"command": cmd, "kernel": "classify_tiles", "read_bytes_per_launch": ck[0][2] if ck else None, "written_bytes_per_launch": ck[0][3] if ck else None,
          "traffic_bytes_per_launch": (ck[0][2] + ck[0][3]) if ck else None, "corrections": "FETCH_SIZE KiB x2 (gfx950), WRITE_SIZE KiB x1", **summary}
    here = os.path.dirname(os.path.abspath(__file__))
    json.dump(js, open(os.path.join(here, tag + "_hbm_traffic.json"), "w"), indent=1)
    json.dump(js, open(os.path.join(here, "hbm_traffic_latest.json"), "w"), indent=1)   # the one bench.py reads
    print(json.dumps(js))


if __name__ == "__main__":
    main()
