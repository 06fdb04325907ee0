"""Timeline of ONE steady-state frame from a rocprofv3 kernel trace of tools/share_probe.py: per kernel its queue, start
offset, duration and the gap to the previous kernel of the same queue.  usage: share_timeline.py <kernel_trace.csv> [frame]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
k.sort()
# frames start with a march_kernel<true, ...> on the main queue; take frame boundaries from the first-half's first march
firsts = [i for i, r in enumerate(k) if "march_kernel<true" in r[2] or "march_kernel<(bool)1" in r[2] or "walk_kernel<true" in r[2] or "walk_kernel<(bool)1" in r[2] or "walk8_kernel<true" in r[2] or "walk8_kernel<(bool)1" in r[2]]
queues = sorted({r[3] for r in k})
# a frame = from one first-march of the lowest queue to the next
q0 = k[firsts[0]][3] if firsts else None
starts = [i for i in firsts if k[i][3] == q0]
want = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 5
a, b = starts[want], starts[want + 1]
t0 = k[a][0]
last_end = {}
print("frame %d: %d kernels, %.1f us from first march to the next frame's first march" % (want, b - a, (k[b][0] - t0) / 1e3))
busy = {}
for s, e, n, q in k[a:b]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    busy[q] = busy.get(q, 0) + (e - s)
    short = n.split("(")[0].replace("vnr::", "")[:44]
    print("q%-3s +%8.1f us  dur %7.1f us  gap %6.1f us  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, short))
for q in busy:
    print("queue %s busy %.1f us" % (q, busy[q] / 1e3))
# frame periods over the whole run
per = [(k[starts[i + 1]][0] - k[starts[i]][0]) / 1e3 for i in range(len(starts) - 1)]
per = per[len(per) // 2:]
print("median frame period (second half of the run): %.1f us over %d frames" % (sorted(per)[len(per) // 2], len(per)))
