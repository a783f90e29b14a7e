#!/usr/bin/env python3
"""Condenses `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr of a compile) into one line per kernel.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> res.txt ; python profiles/tools/resource_usage.py res.txt [filter]"""
import re, subprocess, sys

def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = [b.split("\n")[0].split(" [")[0] for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    print("| kernel | VGPR | AGPR | SGPR | scratch B/lane | VGPR spill | SGPR spill | waves/SIMD | LDS B |")
    print("|---|---|---|---|---|---|---|---|---|")
    for b, d in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        d = re.sub(r"^void ommx::", "", d)
        d = re.sub(r"\(ommx::ClassifyParams.*", "", d)
        if flt and flt not in d:
            continue
        print("| `%s` | %d | %d | %d | %d | %d | %d | %d | %d |" % (d[:120], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
              g("VGPRs Spill"), g("SGPRs Spill"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))

if __name__ == "__main__":
    main()
