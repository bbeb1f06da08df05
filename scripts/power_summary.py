"""Per-phase statistics of a power trace (scripts/power_trace.sh): for the busiest card, mean / max socket power and mean / min shader
clock between each phase's start and end marks (the first and last second of a phase are dropped: process start-up, model creation)."""
import csv, json, sys, os
d = sys.argv[1]
lines = [l for l in open(os.path.join(d, "samples.csv")) if not l.startswith("#")]
src = [l for l in open(os.path.join(d, "samples.csv")) if l.startswith("#")]
print(src[0].strip()[:300] if src else "")
rows = list(csv.DictReader(lines))
marks = {}
for l in open(os.path.join(d, "phases.txt")):
    k, t = l.split()
    marks[k] = float(t)
cards = sorted({c[:-2] for c in rows[0] if c.endswith("_W")}) if rows and "json" not in rows[0] else []
if not cards:
    print("no sysfs samples; raw file kept"); sys.exit(0)
def num(x):
    try: return float(x)
    except Exception: return None
# the card under test: highest MEAN power inside the marked phases (a host has 8 cards; the others belong to other jobs)
def in_phase(t):
    return any(marks.get(p + "_start", 1e30) <= t <= marks.get(p + "_end", -1) for p in ("soak_noise", "soak_zero", "soak_trained", "micro_random", "micro_zero"))
sel_rows = [r for r in rows if in_phase(float(r["t"]))]
busiest = max(cards, key=lambda c: sum(num(r[c + "_W"]) or 0 for r in sel_rows))
print(f"{len(rows)} samples over {float(rows[-1]['t']) - float(rows[0]['t']):.1f} s = {len(rows) / (float(rows[-1]['t']) - float(rows[0]['t'])):.1f} Hz; cards {cards}; busiest {busiest}")
print(f"{'phase':14s} {'seconds':>8s} {'W mean':>8s} {'W max':>8s} {'sclk mean':>10s} {'sclk min':>9s} {'sclk max':>9s}  result")
def stats(t0, t1, name, extra=""):
    sel = [r for r in rows if t0 <= float(r["t"]) <= t1]
    w = [num(r[busiest + "_W"]) for r in sel]; w = [x for x in w if x is not None]
    f = [num(r[busiest + "_sclk_MHz"]) for r in sel]; f = [x for x in f if x is not None]
    if not w and not f:
        print(f"{name:14s} no samples"); return
    mean = lambda v: sum(v) / len(v) if v else float("nan")
    print(f"{name:14s} {t1 - t0:8.1f} {mean(w):8.1f} {max(w) if w else float('nan'):8.1f} {mean(f):10.0f} {min(f) if f else float('nan'):9.0f} {max(f) if f else float('nan'):9.0f}  {extra}")
stats(float(rows[0]["t"]), marks["idle_end"], "idle")
for ph in ("soak_noise", "soak_zero", "soak_trained", "micro_random", "micro_zero"):
    if ph + "_start" not in marks or ph + "_end" not in marks: continue
    extra = ""
    p = os.path.join(d, ph + ".json")
    if os.path.exists(p):
        try:
            j = json.loads(open(p).read().strip().splitlines()[-1])
            extra = f"{j['ms_per_step']:.3f} ms/step, mlp {j['kernel_ms']['mlp']:.3f} ms = {j['roofline']['achieved']:.0f} TFLOP/s, encode {j['kernel_ms']['encode']:.3f}"
            # the bench spends its first seconds building the model: the timed region is the LAST steps*ms_per_step seconds before the end mark
            t1 = marks[ph + "_end"] - 1.0
            t0 = max(marks[ph + "_start"], t1 - j["steps"] * j["ms_per_step"] / 1e3 + 1.0)
            stats(t0, t1, ph, extra); continue
        except Exception as e:
            extra = f"(no JSON: {e})"
    p = os.path.join(d, ph + ".txt")
    if os.path.exists(p):
        ls = [l for l in open(p) if l.startswith("sustain")]
        extra = ls[-1].strip().split(": ", 1)[1] if ls else ""
    stats(marks[ph + "_start"] + 2.0, marks[ph + "_end"] - 0.5, ph, extra)
