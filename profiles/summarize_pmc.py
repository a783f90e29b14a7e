#!/usr/bin/env python3
"""Turns the rocprofv3 `--pmc ... --output-format csv` passes of profiles/collect.sh into
  <dir>/<tag>_pmc.md           per-kernel HBM traffic table + instruction-class / issue-cycle breakdown of classify_tiles
  <dir>/<tag>_hbm_traffic.json the figures bench.py reports as roofline.traffic (stamped with the library they were measured on)

usage: python profiles/summarize_pmc.py <tag> <dir with pmc_*/ passes> "<bench command line>"
The two outputs are then copied into profiles/ (tracked) by hand: gpurun merges only gpurun_out/; the json also becomes
profiles/pmc_latest_<config>.json, which bench.py reads for `roofline` (VALU wave-instructions per launch, shader clock, HBM traffic).

Units and corrections (MI355X_MICROARCH.md, HBM section; calibrated here on kernels with a known byte count, sat_rows and
sat_cols of the 4096^2 texture): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports HALF the bytes read
(sat_rows reads 16.8 MB -> 8207.75 KiB reported; sat_cols reads 67.1 MB -> 32778 KiB) and is doubled below; WRITE_SIZE is
exact (sat_rows writes 67.1 MB -> 65536 KiB) and is used as is.  Every counter group is collected in its own pass.

Issue-cycle model: a wave64 VALU instruction occupies its SIMD for the number of cycles measured by profiles/valu_rates.hip on the same
chip (profiles/valu_rates_mi355x.json: plain fp32 / int / mov ~2.5, conversions / shifts / v_pk_* / fp64 ~4.5, transcendentals ~8.2);
instructions the SQ counters do not classify (moves, compares, selects, bit logic, lane moves) are priced as plain ones.  The scalar unit
is shared by the 4 SIMDs of a CU and issues one SALU or branch instruction per cycle (measured: 4.2 SIMD-cycles per instruction with all
four SIMDs issuing)."""
import collections
import csv
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CLOCK_HZ = 2.4e9
NUM_CU, SIMD_PER_CU = 256, 4


def short(name):
    if "classify_tiles<" in name:   # the persistent launch of the levels >= 6 is THE kernel of the roofline; the other instantiations are listed apart
        return "ommx::classify_tiles" if re.search(r"classify_tiles<\w+, true, 4096", name) else "ommx::classify_tiles (levels < 6)"
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    if "rocprim::" in name:
        m = re.search(r"rocprim::(?:trampoline_kernel<rocprim::)?(?:wrapped_)?(\w+)", name)
        return "rocprim " + (m.group(1) if m else "kernel")
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return re.sub(r"<.*", "", name)


def load(path):
    """per short kernel name: counter -> [launches, sum]; and the kernel durations of that pass"""
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    seen = set()
    if not os.path.exists(path):
        return agg, dur
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k][0] += 1; dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return agg, dur


def lib_stamp():
    """sha256 (16 hex digits) over the library's sources: bench.py prints roofline.traffic only while this still matches the tree"""
    d = os.path.join(os.path.dirname(HERE), "omm_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp", ".inc")) or f == "Makefile":
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def rates():
    """cycles per wave-instruction per SIMD by class, from the committed microbenchmark (defaults = its round-2 numbers)"""
    r = {"plain": 2.5, "half": 4.5, "trans": 8.2, "salu_per_cu": 1.05}
    p = os.path.join(HERE, "valu_rates_mi355x.json")
    if os.path.exists(p):
        c = {x["op"]: x["cycles_per_wave_instr_per_simd_at_2.4GHz"] for x in json.load(open(p))["cases"]}
        plain = [c[k] for k in ("v_mul_f32", "v_add_f32", "v_fma_f32", "v_mov_b32", "v_and_b32", "v_add_u32") if k in c]
        half = [c[k] for k in ("v_cvt_i32_f32", "v_floor_f32", "v_lshlrev_b32", "v_mul_lo_u32", "v_mul_f64", "v_add_f64", "v_div_fixup_f32") if k in c]
        trans = [c[k] for k in ("v_sqrt_f32", "v_rcp_f32", "v_rsq_f32") if k in c]
        if plain: r["plain"] = sum(plain) / len(plain)
        if half: r["half"] = sum(half) / len(half)
        if trans: r["trans"] = sum(trans) / len(trans)
        if "s_add_u32/s_xor_b32" in c: r["salu_per_cu"] = c["s_add_u32/s_xor_b32"] / SIMD_PER_CU
    return r


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

    # ---- instruction classes and issue cycles of the dominant classification kernel: the persistent classify_tiles launch, or the deferred generic
    #      pass (classify_generic) when a bake of asset-sized triangles spends longer there ----
    KERNEL = "ommx::classify_tiles"
    for p in ("C1", "C4"):
        _, dur = load(os.path.join(root, "pmc_" + p, "pmc_counter_collection.csv"))
        dt, dg = dur.get("ommx::classify_tiles"), dur.get("ommx::classify_generic")
        if dt and dg and dg[0] and dt[0] and dg[1] / dg[0] > dt[1] / dt[0]:
            KERNEL = "ommx::classify_generic"
        if dt or dg: break
    KSHORT = KERNEL.split("::")[1]
    c, dur_ms = {}, None
    for p in ("C1", "C2", "C3", "C4", "SQ"):
        agg, dur = load(os.path.join(root, "pmc_" + p, "pmc_counter_collection.csv"))
        for n, v in agg.get(KERNEL, {}).items():
            c[n] = v[1] / max(1, v[0])
        d = dur.get(KERNEL)
        if d and d[0] and dur_ms is None:
            dur_ms = d[1] / d[0]
    summary = {}
    if c:
        R = rates()
        cyc = (c["GRBM_GUI_ACTIVE"] / 8.0) if "GRBM_GUI_ACTIVE" in c else (dur_ms or 0) * 1e-3 * CLOCK_HZ   # shader cycles of one launch (GRBM sums the 8 XCDs)
        simd_cycles, cu_cycles = cyc * NUM_CU * SIMD_PER_CU, cyc * NUM_CU
        out += ["", "## `%s`: instructions per launch by class (SQ_INSTS_*, wave-instructions) and the SIMD cycles they occupy" % KSHORT, "",
                "kernel duration in the counter passes: %s ms; shader cycles per launch %.4g (GRBM_GUI_ACTIVE / 8 XCDs)" % ("%.2f" % dur_ms if dur_ms else "?", cyc), "",
                "| class | counter | wave-instructions | cycles each (profiles/valu_rates_mi355x.json) | share of the VALU pipe (1024 SIMDs x cycles) |", "|---|---|---|---|---|"]
        valu = c.get("SQ_INSTS_VALU", 0.0)
        classes = [("fp32 add", "SQ_INSTS_VALU_ADD_F32", "plain"), ("fp32 mul", "SQ_INSTS_VALU_MUL_F32", "plain"), ("fp32 fma (IEEE div / sqrt expansions)", "SQ_INSTS_VALU_FMA_F32", "plain"),
                   ("fp32 transcendental (v_rcp / v_sqrt / v_rsq)", "SQ_INSTS_VALU_TRANS_F32", "trans"), ("conversions", "SQ_INSTS_VALU_CVT", "half"),
                   ("int32 arithmetic", "SQ_INSTS_VALU_INT32", "plain"), ("int64 arithmetic", "SQ_INSTS_VALU_INT64", "half"),
                   ("fp64 (winding test)", None, "half")]
        known, busy = 0.0, 0.0
        for label, ctr, rate in classes:
            n = c.get(ctr, 0.0) if ctr else sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
            known += n; busy += n * R[rate]
            out.append("| %s | %s | %.4g | %.1f | %.3f |" % (label, ctr or "SQ_INSTS_VALU_{ADD,MUL,FMA}_F64", n, R[rate], n * R[rate] / simd_cycles if simd_cycles else 0))
        other = max(0.0, valu - known)
        busy += other * R["plain"]
        out.append("| moves, compares, selects, bit logic, lane moves (not classified by the SQ) | SQ_INSTS_VALU - sum of the above | %.4g | %.1f | %.3f |" % (other, R["plain"], other * R["plain"] / simd_cycles if simd_cycles else 0))
        out.append("| **all VALU** | SQ_INSTS_VALU | %.4g | | **%.3f** |" % (valu, busy / simd_cycles if simd_cycles else 0))
        scal = c.get("SQ_INSTS_SALU", 0.0) + c.get("SQ_INSTS_BRANCH", 0.0)
        out += ["", "| other units | counter | wave-instructions | per VALU instruction |", "|---|---|---|---|"]
        for label, ctr in (("scalar ALU", "SQ_INSTS_SALU"), ("branches", "SQ_INSTS_BRANCH"), ("scalar memory", "SQ_INSTS_SMEM"), ("LDS", "SQ_INSTS_LDS"),
                           ("vector memory reads", "SQ_INSTS_VMEM_RD"), ("vector memory writes", "SQ_INSTS_VMEM_WR")):
            if ctr in c:
                out.append("| %s | %s | %.4g | %.3f |" % (label, ctr, c[ctr], c[ctr] / valu if valu else 0))
        summary["valu_wave_instructions_per_launch"] = valu
        summary["scalar_wave_instructions_per_launch"] = scal
        summary["scalar_per_valu"] = scal / valu if valu else None
        summary["shader_cycles_per_launch"] = cyc
        summary["kernel_ms_in_counter_pass"] = dur_ms
        summary["shader_clock_hz"] = cyc / (dur_ms * 1e-3) if dur_ms else None   # effective clock of the launch: GRBM_GUI_ACTIVE / 8 XCDs / its duration in the same pass
        summary["valu_issue_utilisation"] = busy / simd_cycles if simd_cycles else None
        summary["valu_instr_per_cycle_per_simd"] = valu / simd_cycles if simd_cycles else None
        summary["scalar_issue_utilisation"] = scal * R["salu_per_cu"] / cu_cycles if cu_cycles else None
        out += ["", "VALU issue utilisation (modelled: sum of class count x measured cycles / (1024 SIMDs x kernel cycles)) = **%.2f**; raw: %.3f VALU wave-instructions per cycle per SIMD" % (summary["valu_issue_utilisation"], summary["valu_instr_per_cycle_per_simd"]),
                "", "Scalar-unit issue utilisation ((SALU + branch) x %.2f cycles / (256 CUs x kernel cycles), one scalar unit per CU) = **%.2f**" % (R["salu_per_cu"], summary["scalar_issue_utilisation"])]
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
            summary["valu_lane_util"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
            out += ["", "VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = **%.2f**" % summary["valu_lane_util"]]
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            out += ["", "SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.2f" % (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"])]
        out += ["", "raw counters: " + ", ".join("%s=%.4g" % kv for kv in sorted(c.items()))]
    open(os.path.join(root, tag + "_pmc.md"), "w").write("\n".join(out) + "\n")
    ck = [r for r in rows if r[0] == KERNEL]
    import socket, time
    js = {"tag": tag, "command": cmd, "kernel": KSHORT, "source_sha256_16": lib_stamp(),
          "collected_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "collected_on": "gpurun lease, host " + socket.gethostname(),
          "read_bytes_per_launch": ck[0][2] if ck else None, "written_bytes_per_launch": ck[0][3] if ck else None,
          "traffic_bytes_per_launch": (ck[0][2] + ck[0][3]) if ck else None, "corrections": "FETCH_SIZE KiB x2 (gfx950), WRITE_SIZE KiB x1", **summary}
    json.dump(js, open(os.path.join(root, tag + "_hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(js))


if __name__ == "__main__":
    main()
