"""How much of the time the HIP lanes of the product pipeline run concurrently, from a rocprofv3 kernel trace (CSV): per stream
(= lane) the busy time, and the time during which kernels of two or more streams are in flight together.
usage: lane_overlap.py <kernel_trace.csv> [excerpt.csv]   (excerpt: every launch of a 6 ms window inside the product pipeline's timed region, for the timeline)"""
import csv, re, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
if not rows:
    sys.exit("empty trace")
key = "Stream_Id" if "Stream_Id" in rows[0] else ("Queue_Id" if "Queue_Id" in rows[0] else None)
ev = []
busy = defaultdict(float)
names = defaultdict(lambda: defaultdict(int))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    lane = r.get(key, "?") if key else "?"
    busy[lane] += (e - s) * 1e-6
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    names[lane][nm] += 1
    ev.append((s, 1, lane))
    ev.append((e, -1, lane))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
active = defaultdict(int)
last = ev[0][0]
any_busy = both = 0.0
for t, d, lane in ev:
    n = sum(1 for v in active.values() if v > 0)
    if n >= 1:
        any_busy += (t - last) * 1e-6
    if n >= 2:
        both += (t - last) * 1e-6
    last = t
    active[lane] += d
print(f"trace: {len(rows)} kernel launches over {(t1 - t0) * 1e-6:.1f} ms; lanes identified by {key}")
for lane in sorted(busy, key=lambda k: -busy[k]):
    top = sorted(names[lane].items(), key=lambda kv: -kv[1])[:4]
    print(f"  lane {lane}: busy {busy[lane]:9.2f} ms  ({', '.join(f'{n} x {c}' for n, c in top)})")
print(f"some lane busy: {any_busy:.2f} ms; two or more lanes busy at once: {both:.2f} ms ({100 * both / max(any_busy, 1e-9):.1f} % of the busy time)")

if len(sys.argv) > 2:
    # a 6 ms window: stream, kernel, start and end in microseconds relative to the window
    solves = sorted(int(r["Start_Timestamp"]) for r in rows if "ba_solve_kernel" in r["Kernel_Name"])
    w0 = solves[len(solves) // 3] if solves else t0   # (a third into the solver launches: the timed region of the product pipeline)
    w1 = w0 + 6_000_000
    with open(sys.argv[2], "w") as out:
        out.write("stream,kernel,start_us,end_us\n")
        for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
            a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if b < w0 or a > w1:
                continue
            nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            out.write(f"{r.get(key, '?')},{nm},{(a - w0) / 1e3:.2f},{(b - w0) / 1e3:.2f}\n")
