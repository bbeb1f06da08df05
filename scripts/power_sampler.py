#!/usr/bin/env python3
"""Samples socket power and shader clock of every amdgpu card at ~20 Hz into a CSV (no HIP: reads sysfs hwmon / pp_dpm_sclk; falls back
to `amd-smi metric` / `rocm-smi` when sysfs is not readable).  Started as a SEPARATE process before the workloads, stopped with SIGTERM.
    python3 scripts/power_sampler.py out.csv [hz]"""
import glob, json, os, signal, subprocess, sys, time

out = sys.argv[1]
hz = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
stop = False
signal.signal(signal.SIGTERM, lambda *a: globals().__setitem__("stop", True))
signal.signal(signal.SIGINT, lambda *a: globals().__setitem__("stop", True))


def rd(path):
    try:
        return open(path).read().strip()
    except Exception:
        return None


cards = []
for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
    hw = glob.glob(dev + "/hwmon/hwmon*")
    if not hw:
        continue
    h = hw[0]
    p = next((x for x in (h + "/power1_average", h + "/power1_input") if rd(x) is not None), None)
    f = h + "/freq1_input" if rd(h + "/freq1_input") is not None else None
    if p or f or rd(dev + "/pp_dpm_sclk"):
        cards.append(dict(name=dev.split("/")[-2], power=p, freq=f, dpm=dev + "/pp_dpm_sclk", busy=dev + "/gpu_busy_percent"))

with open(out, "w") as fo:
    if cards:
        fo.write("# source: sysfs " + json.dumps([{k: v for k, v in c.items()} for c in cards]) + "\n")
        fo.write("t," + ",".join(f"{c['name']}_W,{c['name']}_sclk_MHz,{c['name']}_busy" for c in cards) + "\n")
        while not stop:
            t = time.time()
            row = [f"{t:.3f}"]
            for c in cards:
                pw = rd(c["power"]) if c["power"] else None
                fq = rd(c["freq"]) if c["freq"] else None
                if fq is None:
                    d = rd(c["dpm"]) or ""
                    cur = [l for l in d.splitlines() if l.strip().endswith("*")]
                    fq_mhz = cur[0].split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", "") if cur else ""
                else:
                    fq_mhz = f"{int(fq) / 1e6:.0f}"
                row += [f"{int(pw) / 1e6:.1f}" if pw else "", fq_mhz, rd(c["busy"]) or ""]
            fo.write(",".join(row) + "\n")
            fo.flush()
            time.sleep(max(0.0, 1.0 / hz - (time.time() - t)))
    else:
        tool = "/opt/rocm/bin/amd-smi"
        fo.write("# source: amd-smi metric --power --clock --json (sysfs hwmon not readable)\n")
        fo.write("t,json\n")
        while not stop:
            t = time.time()
            try:
                r = subprocess.run([tool, "metric", "--power", "--clock", "--json"], capture_output=True, text=True, timeout=5).stdout
                fo.write(f"{t:.3f}," + json.dumps(json.loads(r), separators=(",", ":")).replace("\n", " ") + "\n")
            except Exception as e:  # noqa: BLE001
                fo.write(f"{t:.3f},\"error {e}\"\n")
            fo.flush()
            time.sleep(max(0.0, 1.0 / hz - (time.time() - t)))
