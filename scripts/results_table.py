#!/usr/bin/env python3
"""The results table of DESIGN.md section 7 from the bench lines and PMC records under profiles/<round>_*:
    python scripts/results_table.py r05"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"


def line(name):
    p = os.path.join(ROOT, "profiles", "%s_%s.json" % (R, name))
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        return None


def rec(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "%s_%s.json" % (R, name))))
    except Exception:
        return None


rows = [("256^3 Sedov, RK2, HLLC + PLM (BASELINE configs[1]: the headline)", "bench_line", "pmc_traffic"),
        ("1024^2 dusty shearing sheet with drag (config 3 at its own size)", "cfg3_1024_line", "cfg3_1024_pmc_traffic"),
        ("... 4096^2", "cfg3_line", "cfg3_pmc_traffic"),
        ("... 1024^2, two dust species", "cfg3_1024_2dust_line", "cfg3_1024_2dust_pmc_traffic"),
        ("disk_sph.in x 2: 256 x 128^2 spherical, alpha viscosity, gravity, rotating frame", "disk_sph_line", "disk_sph_pmc_traffic"),
        ("... with a refined midplane region (configs[3]'s combination; 464 blocks of 32^3)", "disk_sph_smr_line", "disk_sph_smr_pmc_traffic"),
        ("configs[4] in 3-D: cylindrical disk + planet + dust + drag, four adaptive levels, 7 064 blocks of 16^3", "disk_amr_line", "disk_amr_pmc_traffic"),
        ("... on the deck's own 32^3 blocks", "disk_amr_block32_line", None),
        ("... 16^3 blocks, six remeshes of 300-770 leaves inside the 24 timed cycles", "disk_amr_remesh_in_timed_region_line", None),
        ("configs[3]'s combination through RCCL send / recv-to-self (`--loopback`)", "disk_sph_smr_loopback_line", None),
        ("configs[4]'s combination through RCCL send / recv-to-self (`--loopback`)", "disk_amr_loopback_line", None),
        ("256^3 linear wave, PPM + HLLC (cell-centred stage: PPM is in no tile march)", "linwave3d_ppm_line", None),
        ("256^3 linear wave, PLM + HLLC (tuned tile march)", "linwave3d_plm_line", None)]
print("| workload (`bench.py --workload ...`) | zone-cycles/s | ms per step | kernel(s) of a stage: ms | achieved / 8 TB/s | HBM traffic, measured / algorithmic per stage | CPU oracle, zone-cycles/s (threads: `cpu_baseline.threads` of the line) |")
print("|---|---|---|---|---|---|---|")
for what, ln, tr in rows:
    d = line(ln)
    if not d:
        print("| %s | (no record) | | | | | |" % what)
        continue
    rf = d.get("roofline") or {}
    t = rec(tr) if tr else None
    alg = rf.get("algorithmic_bytes_per_launch")
    traffic = rf.get("traffic") or (t and t.get("hbm_bytes_per_launch"))
    cpu = d.get("cpu_baseline") or {}
    print("| %s | %.3g | %.3f | %.3f | **%.3f** | %s | %s |" % (
        what, d["value"], d["ms_per_step"], rf.get("launch_ms", float("nan")), rf.get("frac", float("nan")),
        ("%.2f / %.2f GB (%.2f x)" % (traffic / 1e9, alg / 1e9, traffic / alg)) if (traffic and alg) else "—",
        ("%.3g" % cpu["value"]) if cpu.get("value") else "—"))
d = line("bench_line")
if d:
    rf = d["roofline"]
    fi = rf.get("fp64_issue") or {}
    print()
    print("Headline details: VALU lane-instructions per zone-stage %s, fp64-issue floor fraction %s; drop-in contracts %s; "
          "overlap emulation %s zone-cycles/s." % (fi.get("lane_instructions_per_zone_stage"), fi.get("frac"),
                                                    json.dumps({k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("dropin") or {}).items()}),
                                                    d["config"].get("overlap_emulation_zcps")))
d = line("disk_amr_line")
if d and d.get("remesh"):
    r = d["remesh"]
    b = r.get("batched") or {}
    print()
    print("Remesh (configs[4] mesh): single-leaf %.1f ms = %.2f cycle-times (build %.1f, hand-over %.1f, tagging %.1f); "
          "batched (the criterion at a lowered threshold, ~300 leaves each): %s ms, %.2f cycle-times on average (max %.2f), "
          "build share %.2f; %.0f B per zone live (%.0f with the allocator's cache)." % (
              r["ms_mean"], r["remesh_over_cycle"], r["ms_mean_split"]["build_state"], r["ms_mean_split"]["hand_over"],
              r["ms_mean_split"]["tagging_incl_cycles_without_remesh"],
              ", ".join("%.0f" % e["ms"] for e in b.get("events", []) if e["created"] > 50), b.get("over_cycle_mean", 0),
              b.get("over_cycle_max", 0), b.get("build_state_share", 0), b.get("bytes_per_zone_live", 0), b.get("bytes_per_zone_now", 0)))
