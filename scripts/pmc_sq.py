#!/usr/bin/env python3
"""SQ counters of the stage kernels: python3 scripts/pmc_sq.py <tag> -- <program args...>
Runs `rocprofv3 --pmc <group> --kernel-trace` once per counter group (never with other trace domains) and prints
per-kernel means for kernels whose name contains 'stage' (or one of the comma-separated PMC_SQ_KERNELS)."""
import collections, csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def library_identity(scope):
    """bench.library_identity(scope), asked of a SHORT-LIVED CHILD: this script starts the profiled program through
    rocprofv3, which replaces itself with it -- a hop the GPU boxes refuse once the starting process tree has had the HIP
    runtime loaded, which dlopen of libartemis_hip.so does."""
    import subprocess
    out = subprocess.check_output([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; print(bench.library_identity(%r))" % (ROOT, scope)],
                                  stderr=subprocess.DEVNULL)
    return out.decode().strip().splitlines()[-1]

GROUPS = [["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU"],
          ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"],
          ["SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"],
          ["SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VMEM"],
          ["SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_MISC"],
          ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_IFETCH"]]
if os.environ.get("PMC_SQ_GROUPS"):  # e.g. PMC_SQ_GROUPS=0,5
    GROUPS = [GROUPS[int(x)] for x in os.environ["PMC_SQ_GROUPS"].split(",")]
tag = sys.argv[1]
prog = sys.argv[sys.argv.index("--") + 1:]
acc = collections.defaultdict(dict)
for gi, g in enumerate(GROUPS):
    out = os.path.join(ROOT, "gpurun_out", "sq_%s_%d" % (tag, gi))
    os.makedirs(out, exist_ok=True)
    cmd = ["rocprofv3", "--pmc"] + g + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--", "python3"] + prog
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print("group", g, "failed:", r.stdout[-400:])
        continue
    tot, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if not any(n in k for n in os.environ.get("PMC_SQ_KERNELS", "stage").split(",")):
                continue
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == g[0]:
                cnt[k] += 1
    for k in tot:
        for c, v in tot[k].items():
            acc[k][c] = v / max(cnt[k], 1)
        acc[k]["launches"] = cnt[k]
res = {k[:150]: v for k, v in acc.items()}
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "sq_%s.json" % tag), "w"), indent=1)
if os.environ.get("PMC_SQ_RECORD"):
    # the record bench.py quotes as roofline.fp64_issue (only while the kernel sources are unchanged): mean VALU
    # wave-instructions per launch over the headline stage kernels (non-curvilinear, non-flux instantiations)
    sys.path.insert(0, ROOT)
    ks = {k: v for k, v in res.items() if "stage_fused_kernel" in k and "SQ_INSTS_VALU" in v}
    if ks:
        rec = {"library_identity": library_identity("fused"), "command": " ".join(prog), "env": {},
               "valu_wave_instructions_per_launch": sum(v["SQ_INSTS_VALU"] for v in ks.values()) / len(ks),
               "valu_active_cycles": sum(v.get("SQ_ACTIVE_INST_VALU", 0.0) for v in ks.values()) / len(ks),
               "wave_cycles": sum(v.get("SQ_WAVE_CYCLES", 0.0) for v in ks.values()) / len(ks),
               "kernels": sorted(ks)}
        json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "%s_pmc_sq.json" % os.environ["PMC_SQ_RECORD"]), "w"), indent=1)
for k, v in res.items():
    print(k[:120])
    print("   ", {c: round(x, 1) for c, x in v.items()})
