#!/usr/bin/env python3
"""Compile one HIP source of the library for gfx950 with -Rpass-analysis=kernel-resource-usage and print a
compact table: kernel, VGPRs, AGPRs, SGPRs, scratch bytes/lane, occupancy (waves/SIMD), LDS bytes.
    python scripts/resource_usage.py kernels_fused.hip [extra hipcc flags] > profiles/r02_resource_kernels_fused.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artemis_amd.build import HIPCC, HIP_FLAGS  # noqa: E402


def main():
    src = os.path.join(ROOT, "artemis_amd", "csrc", sys.argv[1])
    cmd = [HIPCC] + HIP_FLAGS + sys.argv[2:] + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: [^ ]* (Function Name|Name): (\S+)", line)
        if m:
            cur = {"name": m.group(2)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    print("# %s" % " ".join(cmd[:-4] + ["-c", os.path.relpath(src, ROOT)]))
    print("%-110s %5s %5s %5s %8s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
    for r, n in zip(rows, names):
        n = re.sub(r"artemis::\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*$", "", n)
        print("%-110s %5d %5d %5d %8d %4d %7d" % (n[:110], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1),
                                                  r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))


if __name__ == "__main__":
    main()
