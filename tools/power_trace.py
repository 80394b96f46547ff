#!/usr/bin/env python3
"""Diagnostic: socket power, shader clock and temperature of GPU 0 sampled from sysfs while a load runs.

    python tools/power_trace.py OUT_PREFIX -- <command ...>

Samples /sys/class/drm/card*/device/{hwmon/*/power1_input, pp_dpm_sclk, gpu_busy_percent, hwmon/*/temp*_input}
every 20 ms from one second before the command starts until it ends, writes OUT_PREFIX.csv (one row per
sample) and OUT_PREFIX.txt (a summary of the samples taken while the device was busy), and prints the
summary.  Reading sysfs needs no privileges and touches no GPU setting."""
import glob
import os
import re
import subprocess
import sys
import threading
import time


def find_card():
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if glob.glob(os.path.join(dev, "hwmon/*/power1_input")) or os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
            return dev
    return None


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def main():
    if "--" not in sys.argv or len(sys.argv) < 4:
        print(__doc__)
        return 2
    prefix = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    dev = find_card()
    if not dev:
        print("no amdgpu card with telemetry under /sys/class/drm")
        return 1
    hw = (glob.glob(os.path.join(dev, "hwmon/*")) or [""])[0]
    temps = sorted(glob.glob(os.path.join(hw, "temp*_input")))
    cap_w = float(read(os.path.join(hw, "power1_cap")) or 0) / 1e6
    rows = []
    stop = threading.Event()

    def sample():
        t0 = time.perf_counter()
        while not stop.is_set():
            p = read(os.path.join(hw, "power1_input")) or read(os.path.join(hw, "power1_average"))
            sclk = ""
            for line in read(os.path.join(dev, "pp_dpm_sclk")).splitlines():
                if line.rstrip().endswith("*"):
                    m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
                    sclk = m.group(1) if m else ""
            busy = read(os.path.join(dev, "gpu_busy_percent"))
            rows.append((time.perf_counter() - t0, float(p or 0) / 1e6, float(sclk or 0), float(busy or 0),
                         [float(read(t) or 0) / 1000 for t in temps]))
            time.sleep(0.02)

    th = threading.Thread(target=sample, daemon=True)
    th.start()
    time.sleep(1.0)
    t_start = time.perf_counter()
    rc = subprocess.call(cmd)
    t_cmd = time.perf_counter() - t_start
    time.sleep(0.3)
    stop.set()
    th.join()
    with open(prefix + ".csv", "w") as f:
        f.write("t_s,power_w,sclk_mhz,gpu_busy_percent," + ",".join(os.path.basename(t) for t in temps) + "\n")
        for t, p, c, b, tt in rows:
            f.write("%.3f,%.1f,%.0f,%.0f,%s\n" % (t, p, c, b, ",".join("%.1f" % x for x in tt)))
    idle = [r for r in rows if r[0] < 0.9]
    peak = max((r[1] for r in rows), default=0)
    hot = [r for r in rows if r[1] >= 0.8 * peak] if peak > 0 else rows     # samples under load
    def stats(v):
        v = sorted(v)
        return (v[0], v[len(v) // 2], v[-1]) if v else (0, 0, 0)
    lines = ["command: " + " ".join(cmd), "exit code %d, %.1f s; %d samples, %d under load (power >= 0.8 of the largest sample)"
             % (rc, t_cmd, len(rows), len(hot)),
             "power cap (power1_cap): %.0f W" % cap_w,
             "idle before the command: power %.0f W, sclk %.0f MHz" % (stats([r[1] for r in idle])[1], stats([r[2] for r in idle])[1]),
             "under load: socket power min / median / max = %.0f / %.0f / %.0f W" % stats([r[1] for r in hot]),
             "under load: sclk (pp_dpm_sclk, current level) min / median / max = %.0f / %.0f / %.0f MHz" % stats([r[2] for r in hot]),
             "under load: hottest sensor max = %.0f C" % max((max(r[4]) for r in hot if r[4]), default=0)]
    with open(prefix + ".txt", "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    return rc


if __name__ == "__main__":
    sys.exit(main())
