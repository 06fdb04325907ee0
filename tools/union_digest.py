"""Interval digest of one kernel from a rocprofv3 --kernel-trace csv (the digest needed to reproduce `roofline.union` belongs
under profiles/): python tools/union_digest.py <kernel_trace.csv> [kernel substring] [out_intervals.csv]

Splits the kernel's dispatches into legs (a pause of more than 3 ms between two dispatches of it: bench.py's warm-up + timed frames, the
one-stream leg, the brick-off leg, the PSNR sweep), and prints per leg: dispatches, span, the SUM of the launch durations, and the UNION of
their intervals (overlapping launches of the two streams count once).  algorithmic fraction of a leg = bytes per sample x its samples / union
/ 8 TB/s; bench.py prints the same quantity from HIP events for the timed frames.  The intervals themselves (start and duration in us,
relative to the kernel's first dispatch, with the queue id) go to out_intervals.csv so the union can be recomputed without the full trace."""
import csv
import sys
path = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "fused_infer_kernel<2, 32, 64, 0, false>"
out = sys.argv[3] if len(sys.argv) > 3 else None
iv = []
for r in csv.DictReader(open(path)):
    if needle in r["Kernel_Name"]:
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
iv.sort()
if not iv:
    sys.exit("no dispatch of " + needle)
t_ref = iv[0][0]
legs, cur = [], [iv[0]]
for a in iv[1:]:
    if a[0] - max(e for _, e, _ in cur[-8:]) > 3_000_000:
        legs.append(cur); cur = []
    cur.append(a)
legs.append(cur)


def union_ns(xs):
    total, lo, hi = 0, xs[0][0], xs[0][1]
    for s, e, _ in xs[1:]:
        if s > hi:
            total += hi - lo; lo, hi = s, e
        else:
            hi = max(hi, e)
    return total + hi - lo


print(f"kernel: {needle}; {len(iv)} dispatches in {len(legs)} legs (a leg ends at a pause > 3 ms)")
print(f"{'leg':>3} {'dispatches':>10} {'queues':>6} {'start ms':>10} {'span ms':>9} {'sum ms':>9} {'union ms':>9} {'sum/union':>9} {'union/span':>10}")
for k, xs in enumerate(legs):
    span = max(e for _, e, _ in xs) - xs[0][0]
    sm = sum(e - s for s, e, _ in xs)
    un = union_ns(xs)
    print(f"{k:>3} {len(xs):>10} {len({q for _, _, q in xs}):>6} {(xs[0][0] - t_ref) / 1e6:>10.2f} {span / 1e6:>9.3f} {sm / 1e6:>9.3f} {un / 1e6:>9.3f} {sm / un:>9.3f} {un / span:>10.3f}")
if out:
    with open(out, "w") as f:
        f.write("start_us,duration_us,queue\n")
        for s, e, q in iv:
            f.write(f"{(s - t_ref) / 1e3:.2f},{(e - s) / 1e3:.2f},{q}\n")
