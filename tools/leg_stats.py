"""Per-leg kernel statistics of a bench.py run from a rocprofv3 --kernel-trace csv (VERDICT r03 next #3b): rocprofv3's own
*_kernel_stats.csv has ONE row per kernel for the whole process, which mixes bench.py's legs (timed two-stream frames, one-stream leg,
brick-off legs, budget table, PSNR sweep).  This tool cuts the trace into the legs of the dominant kernel exactly as tools/union_digest.py
does (a pause of more than 3 ms between two of its dispatches) and writes, for every leg, a stats csv of ALL kernels dispatched inside the
leg's time window, plus the leg's union of the dominant kernel's intervals:

    python tools/leg_stats.py <kernel_trace.csv> <out_prefix> [dominant kernel substring]
    -> <out_prefix>_leg<k>_kernel_stats.csv   (Name, Calls, TotalDurationNs, AverageNs, MinNs, MaxNs, Percentage)
    -> <out_prefix>_legs.txt                  (one line per leg: dispatches, span, sum, union of the dominant kernel)

so `roofline.avg_launch_ms` (sum / dispatches of the timed leg) and `roofline.union` (union / frames) can be recomputed from profiles/ alone."""
import csv
import sys
from collections import defaultdict

path, prefix = sys.argv[1], sys.argv[2]
needle = sys.argv[3] if len(sys.argv) > 3 else "fused_infer_kernel<2, 32, 64, 0, false>"
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
dom = [x for x in rows if needle in x[2]]
if not dom:
    sys.exit("no dispatch of " + needle)
legs, cur = [], [dom[0]]
for a in dom[1:]:
    if a[0] - max(e for _, e, _, _ in cur[-8:]) > 3_000_000:
        legs.append(cur); cur = []
    cur.append(a)
legs.append(cur)


def union_ns(xs):
    total, lo, hi = 0, xs[0][0], xs[0][1]
    for s, e, _, _ in xs[1:]:
        if s > hi:
            total += hi - lo; lo, hi = s, e
        else:
            hi = max(hi, e)
    return total + hi - lo


t_ref = dom[0][0]
with open(prefix + "_legs.txt", "w") as f:
    f.write(f"dominant kernel: {needle}; {len(dom)} dispatches in {len(legs)} legs (a leg ends at a pause > 3 ms between two of its dispatches)\n")
    f.write(f"{'leg':>3} {'dispatches':>10} {'queues':>6} {'start ms':>10} {'span ms':>9} {'sum ms':>9} {'avg ms':>8} {'union ms':>9} {'sum/union':>9}\n")
    for k, xs in enumerate(legs):
        lo, hi = xs[0][0], max(e for _, e, _, _ in xs)
        sm, un = sum(e - s for s, e, _, _ in xs), union_ns(xs)
        f.write(f"{k:>3} {len(xs):>10} {len({q for *_, q in xs}):>6} {(lo - t_ref) / 1e6:>10.2f} {(hi - lo) / 1e6:>9.3f} {sm / 1e6:>9.3f} {sm / len(xs) / 1e6:>8.4f} {un / 1e6:>9.3f} {sm / un:>9.3f}\n")
        per = defaultdict(list)
        for s, e, name, _ in rows:
            if s >= lo and e <= hi:
                per[name].append(e - s)
        total = sum(sum(v) for v in per.values()) or 1
        with open(f"{prefix}_leg{k}_kernel_stats.csv", "w") as g:
            w = csv.writer(g)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
            for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v), round(100.0 * sum(v) / total, 2)])
print(open(prefix + "_legs.txt").read())
