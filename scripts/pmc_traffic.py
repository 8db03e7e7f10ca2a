#!/usr/bin/env python3
"""Measure the HBM traffic of the dominant kernel with rocprofv3 PMC counters and write the record
bench.py quotes as `roofline.traffic`.

    gpurun -- python3 scripts/pmc_traffic.py [--tag r02] [--workload sedov3d|disk_sph|ssheet_dust] [bench args]

Separate `rocprofv3 --pmc` passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2:
MI355X_MICROARCH.md "rocprofv3 PMC slots"; nothing but --kernel-trace next to --pmc).  Sedov: two passes of
`python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-overlap-emulation`, whose `dropin` legs also run the
per-task kernels (the calibration), and two of the same command with `--no-dropin` as well -- every stage launch of
that run is a full-size launch of one of the two headline instantiations, their number is known (2 x (warmup + steps +
the kernel-timing leg's cycles)) and the record is refused by bench.py if the profile holds any other count.  The calibration: gfx950's FETCH_SIZE under-reports coalesced streams (exactly 1/2
for 16-byte lanes per the guide; this code base loads 8 bytes per lane), so the read correction is
measured in the same profile on kernels whose byte count is known exactly (cons_to_prim: 5 arrays in,
5 out over the interior; prim_to_cons: 5 in, 9 out over the whole block) and applied to the stage kernel.
The record carries the identity the loaded library reports for its own code (artemis_hip_object_sha / artemis_hip_source_sha):
bench.py quotes it only from a library that reports the same one.
This script never touches the GPU itself; the profiled program is started by rocprofv3.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def library_identity(scope):
    """bench.library_identity(scope), asked of a SHORT-LIVED CHILD: this script starts the profiled program through
    rocprofv3, which replaces itself with it -- a hop the GPU boxes refuse once the starting process tree has had the HIP
    runtime loaded, which dlopen of libartemis_hip.so does."""
    import subprocess
    out = subprocess.check_output([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; print(bench.library_identity(%r))" % (ROOT, scope)],
                                  stderr=subprocess.DEVNULL)
    return out.decode().strip().splitlines()[-1]



def run_pass(counter, outdir, bench_args):
    os.makedirs(outdir, exist_ok=True)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "p", "--",
           "python3", os.path.join(ROOT, "bench.py")] + bench_args
    env = dict(os.environ, TMPDIR="/tmp")
    with open(os.path.join(outdir, "run.log"), "w") as log:
        subprocess.check_call(cmd, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT)
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under " + outdir
    acc = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"]
            s = acc.setdefault(k, [0.0, 0])
            s[0] += float(row["Counter_Value"])
            s[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def pick(table, *needles):
    out = {k: v for k, v in table.items() if all(n in k for n in needles)}
    return out


def whole_stage(args, extra):
    """disk_sph / ssheet_dust: several kernels per stage, so the record is HBM bytes per STAGE summed over every
    kernel of the timed run.  FETCH_SIZE is corrected with the ratio calibrated on the Sedov run's known streams
    (profiles/r*_pmc_traffic.json: 0.620 in every run so far; these workloads launch no kernel with an exactly
    known byte count)."""
    n = {"disk_sph": 256, "ssheet_dust": args.n if args.n in (1024, 4096) else 4096}.get(args.workload)
    refined = args.workload in ("disk_amr", "disk_sph_smr")

    def bench_args_for(steps):
        a = ["--workload", args.workload, "--steps", str(steps), "--warmup", "2", "--no-cpu-baseline"] + (["--n", str(n)] if n else []) + extra
        if args.workload == "disk_amr":  # (stages only: the remesh measurements that follow the timed region are left out)
            a.append("--no-remesh-leg")
        return a
    ratio, src = 0.6202, "default"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            rec = json.load(open(path))
            rs = [c["fetch_ratio"] for c in rec["calibration"].values() if isinstance(c, dict)]
            ratio, src = sum(rs) / len(rs), os.path.relpath(path, ROOT)
            break
        except Exception:
            continue
    kib = 1024.0

    def is_stage(k, names):
        # one stage = one launch of the GAS stage kernel over the whole pack (the dust march is the instantiation whose last
        # template argument is true; the refined meshes' fix-up instantiations of stage_cell_kernel end in `true>` as well)
        hit = ("stage_fused_kernel" in k or "stage2d_kernel" in k or ("stage_cell_kernel<0" in k and "true>(" not in k) or
               ("stage_curv_kernel" in k and "false>(" in k))
        if any("stage_curv_kernel" in q for q in names):  # (a curvilinear pack: its stages are the march's launches)
            hit = hit and "stage_curv_kernel" in k
        return hit

    def measure(steps, tagx):
        bench_args = bench_args_for(steps)
        scratch = os.path.join(ROOT, "gpurun_out", "pmc_%s%s" % (args.tag, tagx))
        fetch = run_pass("FETCH_SIZE", scratch + "_fetch", bench_args)
        write = run_pass("WRITE_SIZE", scratch + "_write", bench_args)
        kernels = {}
        for k, (fk, cnt) in fetch.items():
            wk = write.get(k, (0.0, 0))[0]
            kernels[k[:160]] = {"launches": cnt, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "bytes_total": cnt * (fk * kib / ratio + wk * kib)}
        nstage = sum(v["launches"] for k, v in kernels.items() if is_stage(k, kernels))
        assert nstage, "no stage kernel in the profile"
        return bench_args, kernels, nstage
    # every launch of a run (initialisation, warm-up and timed cycles alike) is in its profile.  Uniform meshes: normalise
    # by the launches of the stage kernel, which runs once per stage (the set-up is a fraction of a per cent).  Refined
    # meshes: the initial refinement loop launches as many bytes as several stages, so TWO runs that differ in the number
    # of timed cycles only are profiled and the record is the DIFFERENCE of their totals over the difference of their stages.
    bench_args, kernels, nstage = measure(6, "")
    total = sum(v["bytes_total"] for v in kernels.values())
    method = "bytes of EVERY kernel of the run divided by the number of stage-kernel launches (= stages)"
    if refined:
        _, k2, n2 = measure(12, "_b")
        t2 = sum(v["bytes_total"] for v in k2.values())
        assert n2 > nstage
        per_kernel = {}
        for k, v in k2.items():
            d = v["bytes_total"] - kernels.get(k, {"bytes_total": 0.0})["bytes_total"]
            per_kernel[k] = {"launches_in_the_extra_cycles": v["launches"] - kernels.get(k, {"launches": 0})["launches"],
                             "bytes_per_stage": d / (n2 - nstage)}
        per_stage = (t2 - total) / (n2 - nstage)
        method = ("two runs, --steps 6 and --steps 12: (bytes of every kernel of the second - of the first) / (stages of the "
                  "second - of the first): the set-up and the initial refinement loop cancel")
        kernels = {"first_run": kernels, "difference_per_stage": per_kernel}
        nstage = n2 - nstage
    else:
        per_stage = total / nstage
    rec = {
        "source": "scripts/pmc_traffic.py --workload %s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, "
                  "--kernel-trace only) of `python3 bench.py %s`, MI355X; %s" % (args.workload, " ".join(bench_args), method),
        "workload": args.workload, "sha_scope": "all", "library_identity": library_identity("all"),
        "env": {k: v for k, v in os.environ.items() if k.startswith("ARTEMIS_")},
        "fetch_correction": "true_read = FETCH_SIZE / %.4f (calibration of %s)" % (ratio, src),
        "stages": nstage, "kernels": kernels, "hbm_bytes_per_launch": per_stage,
    }
    out = args.out or os.path.join(ROOT, "gpurun_out", "%s_%s_pmc_traffic.json" % (args.tag, ("cfg3" if n == 4096 else "cfg3_%d" % n) if args.workload == "ssheet_dust" else args.workload))
    rec["n"] = n
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps({"hbm_bytes_per_stage": per_stage, "stages": nstage, "out": out}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--out", default=None)
    ap.add_argument("--workload", default="sedov3d", choices=["sedov3d", "disk_sph", "ssheet_dust", "disk_sph_smr", "disk_amr"])
    args, extra = ap.parse_known_args()
    from bench import ALG_BYTES_PER_CELL_STAGE
    if args.workload != "sedov3d":
        return whole_stage(args, extra)
    steps, warmup = 6, 2
    bench_args = ["--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-overlap-emulation", "--n", str(args.n)] + extra
    scratch = os.path.join(ROOT, "gpurun_out", "pmc_%s" % args.tag)
    fetch = run_pass("FETCH_SIZE", scratch + "_fetch", bench_args)   # with the dropin legs: the calibration kernels
    write = run_pass("WRITE_SIZE", scratch + "_write", bench_args)
    head_args = bench_args + ["--no-dropin"]                         # the headline path alone: full-size launches only
    hfetch = run_pass("FETCH_SIZE", scratch + "_hfetch", head_args)
    hwrite = run_pass("WRITE_SIZE", scratch + "_hwrite", head_args)
    expected = 2 * (warmup + steps + max(5, min(steps, 40)))          # rk2: two launches per cycle (bench.py's legs)
    n, ng = args.n, 2
    interior, entire = float(n) ** 3, float(n + 2 * ng) ** 3
    kib = 1024.0
    calib = {}
    for name, rd, wr in (("cons_to_prim_kernel", 5 * 8 * interior, 5 * 8 * interior),
                         ("prim_to_cons_kernel<false, false>", 5 * 8 * entire, 9 * 8 * entire)):  # (not the ghost-zone form <false, true>)
        f, w = pick(fetch, name), pick(write, name)
        if f and w:
            fk, wk = list(f.values())[0][0], list(w.values())[0][0]
            calib[name] = {"true_read_KiB": rd / kib, "FETCH_SIZE_KiB": fk, "fetch_ratio": fk * kib / rd,
                           "true_write_KiB": wr / kib, "WRITE_SIZE_KiB": wk, "write_ratio": wk * kib / wr}
    assert calib, "calibration kernels not in the profile (did the dropin legs run?)"
    ratio = sum(c["fetch_ratio"] for c in calib.values()) / len(calib)
    wratio = sum(c["write_ratio"] for c in calib.values()) / len(calib)
    stage = {}
    for k, (fk, cnt) in pick(hfetch, "stage_fused_kernel").items():
        wk = hwrite.get(k, (0.0, 0))[0]
        stage[k] = {"launches": cnt, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                    "read_bytes_corrected": fk * kib / ratio, "write_bytes": wk * kib}
    assert stage, "stage_fused_kernel not in the profile"
    # mean over the launches of the two headline instantiations (rk2: one launch of each per cycle) in the run WITHOUT
    # the dropin and emulation legs.  (A run with them also launches the WRITE_CONS instantiations -- 4th template
    # argument true -- and the FLUXES one -- 8th true: the flux TASK through the tile march; off_headline guards
    # against those should the flags change.)
    def off_headline(name):
        targs = [t.strip() for t in name.split("stage_fused_kernel<", 1)[1].split(">", 1)[0].split(",")]
        return targs[3] == "true" or (len(targs) > 7 and targs[7] == "true")
    for k2, v in stage.items():
        v["headline_path"] = not off_headline(k2)
    head = [v for v in stage.values() if v["headline_path"]] or list(stage.values())
    tot = sum(v["launches"] for v in head)
    assert tot == expected, "stage launches in the profile: %d, expected %d (2 x cycles of the run)" % (tot, expected)
    per_launch = sum(v["launches"] * (v["read_bytes_corrected"] + v["write_bytes"]) for v in head) / tot
    rec = {
        "source": "scripts/pmc_traffic.py: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) "
                  "of `python3 bench.py %s`, MI355X" % " ".join(bench_args),
        "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch",
        "workload": "sedov3d", "sha_scope": "fused", "library_identity": library_identity("fused"),
        "env": {k: v for k, v in os.environ.items() if k.startswith("ARTEMIS_")},
        "calibration": dict(calib, fetch_correction="true_read = FETCH_SIZE / %.4f (mean of the calibration kernels); "
                                                    "WRITE_SIZE as reported (calibrates at %.3f)" % (ratio, wratio)),
        "stage_fused_kernel": stage,
        "hbm_bytes_per_launch": per_launch,
        "headline_launches": tot, "headline_launches_expected": expected,
        "headline_command": "python3 bench.py " + " ".join(head_args),
        "algorithmic_bytes_per_launch": ALG_BYTES_PER_CELL_STAGE * interior,
    }
    out = args.out or os.path.join(ROOT, "gpurun_out", "%s_pmc_traffic.json" % args.tag)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps({"hbm_bytes_per_launch": per_launch, "fetch_ratio": ratio, "write_ratio": wratio, "out": out}))


if __name__ == "__main__":
    main()
