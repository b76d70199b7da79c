#!/usr/bin/env python3
"""Collect the round's profile artefacts on the GPU box (this process never touches the GPU; every pass is a child):

    cd /tmp && TMPDIR=/tmp python $REPO/tools/collect_profiles.py r02        -> $REPO/gpurun_out/r02/*

  <tag>_bench_n1.json                    python3 bench.py                                       (the default line)
  <tag>_bench_strong_n1.json             python3 bench.py --scaling strong --single-mode --steps 2
  <tag>_bench_kernel_stats_<mode>.csv    rocprofv3 --kernel-trace --stats -- python3 bench.py --single-mode --precision <mode> ...
  <tag>_pmc_hbm_traffic.json             two more runs per mode with --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), folded
  <tag>_pmc_mfma_busy.json               one run per mode with --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
Copy what should be judged into profiles/ (tracked)."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
COMMON = ["--single-mode", "--no-cpu-baseline", "--no-kernel-timing", "--no-pmc-traffic", "--no-slam-loop"]


def run(cmd, log):
    with open(log, "w") as f:
        return subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=f, stderr=subprocess.STDOUT).returncode


def fold(path, counter):
    """{kernel: [launches, summed counter, summed ns]} over the launches of at least a tenth of the kernel's largest grid: the bench plan's (the process
    also runs the calibration, ~90 four-frame forwards whose launches of the same instantiations are 1 / 32 of the plan's)"""
    by = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                a = by.setdefault((row["Kernel_Name"], int(row.get("Grid_Size") or row.get("Grid_Size_X") or 0)), [0, 0.0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
                a[2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    gmax, agg = {}, {}
    for (name, grid) in by:
        gmax[name] = max(gmax.get(name, 0), grid)
    for (name, grid), a in by.items():
        if grid * 10 >= gmax[name]:               # at least a tenth of the kernel's largest grid: the plan's launches
            t = agg.setdefault(name, [0, 0.0, 0.0])
            for i in range(3):
                t[i] += a[i]
    return agg


def main(tag, only_stats=False):
    out = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out, exist_ok=True)
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not only_stats:
        with open(os.path.join(out, f"{tag}_bench_n1.json"), "w") as f:
            subprocess.run(["python3", BENCH], cwd="/tmp", stdout=f, stderr=open(os.path.join(out, "bench_n1.err"), "w"))
        with open(os.path.join(out, f"{tag}_bench_strong_n1.json"), "w") as f:
            subprocess.run(["python3", BENCH, "--scaling", "strong", "--single-mode", "--steps", "2", "--warmup", "1", "--no-pmc-traffic"],
                           cwd="/tmp", stdout=f, stderr=open(os.path.join(out, "bench_strong.err"), "w"))
    traffic, busy = {"modes": {}}, {"modes": {}}
    for mode in ("accurate", "fast"):
        d = os.path.join(out, f"stats_{mode}")
        run([rp, "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", mode, "--", "python3", BENCH, *COMMON,
             "--precision", mode, "--steps", "3", "--warmup", "1"], os.path.join(out, f"stats_{mode}.log"))
        for p in glob.glob(os.path.join(d, "**", f"{mode}_kernel_stats.csv"), recursive=True):
            shutil.copy(p, os.path.join(out, f"{tag}_bench_kernel_stats_{mode}.csv"))
        # The summary above averages a kernel over ALL its launches of the process -- since round 6 that includes the calibration's ~90 four-frame
        # forwards, whose launches of the same instantiation are short.  The same trace per (kernel, grid size): the rows with the bench plan's grid
        # are the timed steps' launches (+ the warm-up step), which is what bench.py's HIP events average.
        for p in glob.glob(os.path.join(d, "**", f"{mode}_kernel_trace.csv"), recursive=True):
            by = {}
            with open(p, newline="") as f:
                for row in csv.DictReader(f):
                    g = int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0)
                    a = by.setdefault((row["Kernel_Name"], g), [0, 0.0, 1e30, 0.0])
                    dt = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    a[0] += 1; a[1] += dt; a[2] = min(a[2], dt); a[3] = max(a[3], dt)
            tot = sum(a[1] for a in by.values()) or 1.0
            gmax, plan = {}, {}
            for (name, g) in by:
                gmax[name] = max(gmax.get(name, 0), g)
            for (name, g), a in by.items():
                if g * 10 >= gmax[name]:
                    t = plan.setdefault(name, [0, 0.0])
                    t[0] += a[0]; t[1] += a[1]
            with open(os.path.join(out, f"{tag}_bench_kernel_stats_{mode}_plan.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "note: launches of at least a tenth of the kernel's largest grid = the bench plan's (warm-up + timed steps)"])
                for name, t in sorted(plan.items(), key=lambda kv: -kv[1][1])[:40]:
                    w.writerow([name, t[0], int(t[1]), round(t[1] / t[0], 1)])
            with open(os.path.join(out, f"{tag}_bench_kernel_stats_{mode}_by_grid.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Name", "Grid_Size_X", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
                for (name, g), a in sorted(by.items(), key=lambda kv: -kv[1][1])[:60]:
                    w.writerow([name, g, a[0], int(a[1]), round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 2), int(a[2]), int(a[3])])
        if only_stats:
            continue
        per = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, f"pmc_{counter}_{mode}")
            run([rp, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", "python3", BENCH, *COMMON,
                 "--precision", mode, "--steps", "1", "--warmup", "0"], os.path.join(out, f"pmc_{counter}_{mode}.log"))
            for p in glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True):
                for k, (n, v, _) in fold(p, counter).items():
                    per.setdefault(k, {})[counter] = (n, v / n)
        traffic["modes"][mode] = {"kernels": {
            k: dict(launches=max(v.get("FETCH_SIZE", (0, 0))[0], v.get("WRITE_SIZE", (0, 0))[0]), FETCH_SIZE_KB=v.get("FETCH_SIZE", (0, 0.0))[1],
                    WRITE_SIZE_KB=v.get("WRITE_SIZE", (0, 0.0))[1],
                    hbm_bytes_per_launch_corrected=(2.0 * v.get("FETCH_SIZE", (0, 0.0))[1] + v.get("WRITE_SIZE", (0, 0.0))[1]) * 1024.0)
            for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0.0))[1] * kv[1].get("FETCH_SIZE", (0, 0))[0])[:24]}}
        d = os.path.join(out, f"pmc_busy_{mode}")
        run([rp, "--kernel-trace", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", d, "-o", "p", "--",
             "python3", BENCH, *COMMON, "--precision", mode, "--steps", "1", "--warmup", "0"], os.path.join(out, f"pmc_busy_{mode}.log"))
        for p in glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True):
            mf, ga = fold(p, "SQ_VALU_MFMA_BUSY_CYCLES"), fold(p, "GRBM_GUI_ACTIVE")
            rows = {}
            for k in mf:
                if k in ga and ga[k][1] > 0:
                    cyc = ga[k][1] / 8.0                       # GRBM_GUI_ACTIVE is summed over the 8 XCDs
                    rows[k] = dict(launches=mf[k][0], mfma_busy_frac=round(mf[k][1] / (cyc * 1024.0), 4),
                                   eff_clock_ghz=round(cyc / max(ga[k][2], 1.0), 3), total_ms=round(ga[k][2] / 1e6, 3))
            busy["modes"][mode] = dict(sorted(rows.items(), key=lambda kv: -kv[1]["total_ms"])[:16])
    if only_stats:
        for d in glob.glob(os.path.join(out, "stats_*")):
            if os.path.isdir(d):
                shutil.rmtree(d, ignore_errors=True)
        print("collected (kernel stats only):", sorted(os.listdir(out)))
        return
    traffic["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `python3 bench.py --single-mode --precision MODE --steps 1 "
                       "--warmup 0 --no-cpu-baseline --no-kernel-timing --no-pmc-traffic` (default batch, f16); values are KB per launch, mean over a kernel's "
                       "launches; hbm_bytes_per_launch_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 64 of every 128 streamed bytes, "
                       "MI355X_MICROARCH.md 'HBM'; it counts L2 misses, Infinity-Cache hits included)")
    traffic["batch"] = 128
    busy["note"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on the same command. mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                    "(GRBM_GUI_ACTIVE/8 x 1024 SIMDs); eff_clock_ghz = (GRBM_GUI_ACTIVE/8) / summed kernel time")
    json.dump(traffic, open(os.path.join(out, f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)
    json.dump(busy, open(os.path.join(out, f"{tag}_pmc_mfma_busy.json"), "w"), indent=1)
    for d in glob.glob(os.path.join(out, "stats_*")) + glob.glob(os.path.join(out, "pmc_*")):
        if os.path.isdir(d):
            shutil.rmtree(d, ignore_errors=True)
    print("collected:", sorted(os.listdir(out)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02", only_stats="--only-stats" in sys.argv)
