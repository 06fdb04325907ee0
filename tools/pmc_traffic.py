#!/usr/bin/env python3
"""Fabric traffic of the evaluation kernel from rocprofv3 --pmc passes of bench.py itself (tools/r06_profiles.sh):

  pmc_traffic.py <dir with pmc_bench_{FETCH,WRITE}_SIZE[_h1]/bench_counter_collection.csv> <bench.json of the same box> > profiles/r02_pmc_traffic.json

A pass renders `warmup + steps` frames; the first frames run before the brick image exists (the network builds it behind the second
frame's launches), so only the LAST `frames` frames are used: the dispatches of fused_infer_kernel<2,32,64,0,false>
split evenly over the frames of the pass (argument 4, default 4 = --warmup 1 --steps 3)."""
import csv
import json
import os
import sys

KERNEL = "fused_infer_kernel<2, 32, 64, 0, false>"


def frames_of(path, n_frames):
    """-> list of frames, each a list of counter values (KiB) of the evaluation kernel's dispatches.  Dispatch ids are in enqueue
    order and every steady-state frame enqueues the same number of launches (ITERS x halves)"""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows if KERNEL in r["Kernel_Name"]]
    # the first frame of a pass launches more (no iteration count to predict from, and it keeps the evaluation launch behind its last
    # march): the surplus dispatches are the pass's first ones, the frames that are used are the last
    per = len(vals) // n_frames
    vals = vals[len(vals) - per * n_frames:]
    return [vals[i * per:(i + 1) * per] for i in range(n_frames)]


def leg(d, suffix, samples_per_frame, n_last, n_frames):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(d, f"pmc_bench_{c}{suffix}", "bench_counter_collection.csv")
        if not os.path.exists(p):
            return None
        fr = frames_of(p, n_frames)
        use = fr[-n_last:]
        out[c] = {"frames_in_pass": len(fr), "frames_used": len(use), "launches_per_frame": len(use[0]),
                  "kib_per_frame": [round(sum(f), 1) for f in fr], "kib_per_frame_used": sum(sum(f) for f in use) / len(use)}
    read = out["FETCH_SIZE"]["kib_per_frame_used"] * 1024.0 * 2.0
    write = out["WRITE_SIZE"]["kib_per_frame_used"] * 1024.0
    return {"raw": out, "read_corrected_per_frame": read, "write_per_frame": write, "read_per_sample": read / samples_per_frame,
            "write_per_sample": write / samples_per_frame, "bytes_per_sample": (read + write) / samples_per_frame,
            "traffic_over_algorithmic": (read + write) / samples_per_frame / 528.0}


def main():
    d, bench = sys.argv[1], json.load(open(sys.argv[2]))
    n_last = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    n_frames = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    spf = bench["samples_per_frame"]
    doc = {"what": "fabric (L2-miss) traffic of fused_infer_kernel<2,32,64,0,false> on the default bench frame, measured on bench.py itself",
           "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-psnr "
                      "--no-alone --no-brick-off --train-steps 300   (tools/r06_profiles.sh; one counter per pass, program directly after --; "
                      "VNR_AMD_BRICK=1 so that the brick image exists from the first launch; VNR_AMD_RENDER_HALVES=1 for the one-stream leg)",
           "unit_note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB here (derived counters, 1024-byte units)",
           "correction": "FETCH_SIZE counts 64 B per TCC_EA0_RDREQ on gfx950 but every request moves a 128-B line: x2 (MI355X_MICROARCH.md HBM "
                         "section; profiles/r01_pmc_traffic.json calibration).  WRITE_SIZE as reported.",
           "frames_used": "the last %d frames of each pass (the earlier ones run before the brick image exists)" % n_last,
           "samples_per_frame": spf, "algorithmic_bytes_per_sample": 528}
    # what the kernel was compiled from when the passes ran: bench.py puts these numbers into its line only while the hash still holds
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    doc["source_files"] = list(bench.EVAL_KERNEL_SOURCES)
    doc["source_sha16"] = bench.sources_sha16(bench.EVAL_KERNEL_SOURCES)
    two, one = leg(d, "", spf, n_last, n_frames), leg(d, "_h1", spf, n_last, n_frames)
    if two:
        doc["two_streams"] = two
    doc["one_stream"] = one or two
    if not one:
        doc["one_stream_note"] = "no one-stream pass in this set: the two-stream figure stands in (round 1 measured 206 vs 202 B per sample)"
    json.dump(doc, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
