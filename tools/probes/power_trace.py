"""Probe: board power and shader clock while the bench's plan runs (is the step power-limited?).
Samples the amdgpu hwmon files (power1_average / power1_input, freq1_input) every 20 ms from this process while a child process runs
`bench.py --single-mode ...`; prints min / median / max per phase and the power cap.
    python3 tools/probes/power_trace.py [extra bench args]          (on the GPU box; this process never touches the GPU)"""
import glob
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rd(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def hwmons():
    out = []
    for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if rd(os.path.join(h, "name")) == "amdgpu":
            out.append(h)
    return out


def sample(h):
    p = rd(os.path.join(h, "power1_average")) or rd(os.path.join(h, "power1_input"))
    f = rd(os.path.join(h, "freq1_input"))
    t = rd(os.path.join(h, "temp1_input")) or rd(os.path.join(h, "temp2_input"))
    return (float(p) / 1e6 if p and p.lstrip("-").isdigit() else None, float(f) / 1e6 if f and f.isdigit() else None,
            float(t) / 1e3 if t and t.lstrip("-").isdigit() else None)


def main():
    hs = hwmons()
    print(f"hwmon nodes: {hs}")
    if not hs:
        print("no amdgpu hwmon node readable: falling back to rocm-smi once")
        print(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout[-3000:])
        return
    idle = {h: sample(h) for h in hs}
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--single-mode", "--no-slam-loop", "--no-pmc-traffic", "--no-outlier-leg", "--no-pcie-leg",
            "--steps", "12", "--warmup", "2"] + sys.argv[1:]
    t0 = time.time()
    child = subprocess.Popen(args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
    allrows = {h: [] for h in hs}
    while child.poll() is None:
        t = time.time() - t0
        for h in hs:
            allrows[h].append((t,) + sample(h))
        time.sleep(0.02)
    # the box's sysfs shows every tenant's board: ours is the one whose power rose the most over its idle reading while the child ran
    def rise(h):
        pw = sorted(r[1] for r in allrows[h] if r[1] is not None)
        return (pw[int(0.9 * len(pw))] - (idle[h][0] or 0.0)) if pw else -1.0
    h = max(hs, key=rise)
    rows = allrows[h]
    print(f"board: {h} (90th-percentile power {rise(h):.0f} W above its idle reading; the others: {sorted(round(rise(o)) for o in hs if o != h)})")
    for name in ("power1_cap", "power1_cap_max", "power1_cap_default"):
        v = rd(os.path.join(h, name))
        print(f"{name}: {float(v) / 1e6 if v and v.isdigit() else v} W")
    print("idle:", idle[h])
    line = child.stdout.read().strip().splitlines()[-1] if child.stdout else ""
    import json
    try:
        d = json.loads(line)
        print(f"bench: {d['value']} frames/s, {d['ms_per_step']} ms per step, dominant-kernel frac {d['roofline']['frac']}")
        print(f"bench line's own reading over its timed steps (the board of the device's PCI function): {d.get('power')}")
    except Exception:
        print("bench line unreadable:", line[-300:])
    busy = [r for r in rows if r[1] is not None and r[1] > 0.5 * max(x[1] for x in rows if x[1] is not None)]
    print(f"{len(rows)} samples over {rows[-1][0]:.1f} s; {len(busy)} with power above half the maximum seen (the plan running)")
    for label, sel in (("all", rows), ("plan running", busy)):
        for k, unit, idx in (("power", "W", 1), ("shader clock", "MHz", 2), ("temperature", "C", 3)):
            v = [r[idx] for r in sel if r[idx] is not None]
            if v:
                print(f"  {label:13s} {k:13s}: min {min(v):8.1f}  median {statistics.median(v):8.1f}  max {max(v):8.1f} {unit}")
    # the last 3 s in 100 ms bins (the timed steps)
    tail = [r for r in rows if r[0] > rows[-1][0] - 4.0]
    print("last 4 s, 200 ms bins: (t, W, MHz)")
    b0 = tail[0][0] if tail else 0
    for i in range(20):
        seg = [r for r in tail if b0 + 0.2 * i <= r[0] < b0 + 0.2 * (i + 1)]
        pw = [r[1] for r in seg if r[1] is not None]
        fq = [r[2] for r in seg if r[2] is not None]
        if pw:
            print(f"   {seg[0][0]:6.1f}  {statistics.mean(pw):7.1f}  {statistics.mean(fq) if fq else float('nan'):7.1f}")


if __name__ == "__main__":
    main()
