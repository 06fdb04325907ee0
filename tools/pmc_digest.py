#!/usr/bin/env python3
"""Digest of the round-5 counter passes (tools/r06_infer_bound.sh: one rocprofv3 --pmc pass per hardware block on the whole bench frame):

  pmc_digest.py <dir with <pass>.summary.txt> <out dir>
    -> <out>/r05_infer_bound.txt            what bounds fused_infer_kernel: per-dispatch means of every counter, derived ratios, a reading
    -> <out>/r05_mfma_pmc.json              roofline.mfma.util_by_counters of bench.py (stamped with the sources' hash)
    -> <out>/r05_train_atomic_pmc.json      train_roofline.requests_by_counters of bench.py

Ratios use GRBM_GUI_ACTIVE / 8 (one count per XCD) as the kernel's duration in cycles; 256 CUs (one TA / TCP each), 1024 SIMDs."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

EVAL = "fused_infer_kernel<2, 32, 64, 0, false>"
ROUND = os.environ.get("VNR_PROFILE_ROUND", "r06")   # prefix of the files written (profiles/<round>_*)
TRAIN_SOURCES = ("network_train.hip", "grid_device.h")


def read(path):
    """-> {kernel: {counter: (dispatches, mean)}}"""
    out = {}
    for line in open(path):
        m = re.match(r"(.*?)\s+(\S+)\s+dispatches=\s*(\d+) sum=(\S+) mean=(\S+)\s*$", line)
        if m:
            out.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(5)))
    return out


def kernel(doc, needle):
    for k, v in doc.items():
        if needle in k:
            return v
    return {}


def main():
    d, out = sys.argv[1], sys.argv[2]
    passes = {n[:-len(".summary.txt")]: read(os.path.join(d, n)) for n in sorted(os.listdir(d)) if n.endswith(".summary.txt")}
    lines = ["fused_infer_kernel<2, 32, 64, 0, false> on the C4 bench frame (1024^2, L16 F2 T2^22 + 3x64): hardware counters, one rocprofv3 --pmc pass per",
             "block (tools/r06_infer_bound.sh; the program itself after `--`; counters only).  Per-dispatch MEANS; a dispatch evaluates ~3.4 M samples",
             "(~53 k wave tiles of 64).  Under --pmc the dispatches are serialised, so these describe the kernel with the GPU to itself.", ""]
    tiles_on = [0.0]
    for leg, label in (("on", "brick image in use (the bench default)"), ("off", "brick image off: the hashed parameter blob")):
        c = {}
        for name, doc in passes.items():
            if name.endswith("_" + leg) or (leg == "on" and name in ("tcc_atomic",)):
                for k, v in kernel(doc, EVAL).items():
                    if k != "GRBM_GUI_ACTIVE" or "GRBM_GUI_ACTIVE" not in c:
                        c[k] = v
        if not c:
            continue
        lines.append(f"== {label}")
        for k in sorted(c):
            lines.append(f"   {k:40s} dispatches {c[k][0]:5d}   mean {c[k][1]:.6g}")
        T = c.get("GRBM_GUI_ACTIVE", (0, 0))[1] / 8.0
        g = lambda k: c.get(k, (0, 0.0))[1]
        if T > 0:
            lines.append(f"   -> duration {T:.4g} cycles per dispatch (GRBM_GUI_ACTIVE / 8 XCDs)")
            if g("SQ_WAVE_CYCLES"):
                lines.append(f"   -> waves resident on average {g('SQ_WAVE_CYCLES') * 4 / T:.0f} of 4096 slots (SQ_WAVE_CYCLES is in quad-cycles)")
                lines.append(f"   -> wave time waiting for anything (s_waitcnt ...) {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}; waiting for an issue slot {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}")
            if g("SQ_INSTS_VALU"):
                if leg == "on":
                    tiles_on[0] = g("SQ_INSTS_VMEM_RD") / 65.0   # with the image every level is 4 gathers: 64 + the queue record per tile of 64 samples
                tiles = tiles_on[0] or g("SQ_INSTS_VMEM_RD") / 65.0
                lines.append(f"   -> per wave tile of 64 samples: VALU instructions {g('SQ_INSTS_VALU') / tiles:.0f}, SALU {g('SQ_INSTS_SALU') / tiles:.0f}, VMEM reads {g('SQ_INSTS_VMEM_RD') / tiles:.0f}"
                             + (" (64 gathers + the queue record)" if leg == "on" else " (hashed levels: a second gather per row for the lanes whose x is 3 mod 4; every odd x until round 5)"))
                lines.append(f"   -> VALU issue: {g('SQ_INSTS_VALU') * 4 / (1024 * T):.3f} of the SIMD cycles (4 cycles per wave64 instruction, 1024 SIMDs)")
            if g("TA_TA_BUSY_sum"):
                lines.append(f"   -> texture addresser busy {g('TA_TA_BUSY_sum') / 256 / T:.3f} of the time (TA_TA_BUSY summed over 256 TAs)")
            if g("TCP_PENDING_STALL_CYCLES_sum"):
                lines.append(f"   -> L1 (TCP): stalled with requests pending {g('TCP_PENDING_STALL_CYCLES_sum') / 256 / T:.3f} of the time; tag-conflict stalls {g('TCP_READ_TAGCONFLICT_STALL_CYCLES_sum') / 256 / T:.3f}; "
                             f"hit rate {1 - g('TCP_TCC_READ_REQ_sum') / max(g('TCP_TOTAL_CACHE_ACCESSES_sum'), 1):.3f}; mean latency of an L2 read {g('TCP_TCC_READ_REQ_LATENCY_sum') / max(g('TCP_TCC_READ_REQ_sum'), 1):.0f} cycles")
            if g("TCC_REQ_sum"):
                lines.append(f"   -> L2 (TCC): hit rate {g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1):.3f}; fabric reads {g('TCC_EA0_RDREQ_sum'):.4g} x 128 B = {g('TCC_EA0_RDREQ_sum') * 128 / 1e6:.0f} MB per dispatch")
            if g("SQ_VALU_MFMA_BUSY_CYCLES"):
                lines.append(f"   -> matrix cores busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * T):.4f} of the SIMD cycles")
        lines.append("")
    # (the reading of these numbers is written by hand under the file in profiles/: a paragraph generated here would keep last round's figures)
    os.makedirs(out, exist_ok=True)
    open(os.path.join(out, ROUND + "_infer_bound.txt"), "w").write("\n".join(lines))
    print("\n".join(lines))

    # ---- JSON for bench.py ----------------------------------------------------------------------------------------------------
    mf = kernel(passes.get("mfma_on", {}), EVAL)
    if mf.get("SQ_VALU_MFMA_BUSY_CYCLES") and mf.get("GRBM_GUI_ACTIVE"):
        util = mf["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (mf["GRBM_GUI_ACTIVE"][1] / 8.0 * 1024.0)
        json.dump({"what": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) of fused_infer_kernel<2,32,64,0,false> on the bench frame, brick image in use",
                   "util_by_counters": round(util, 4), "dispatches": mf["GRBM_GUI_ACTIVE"][0], "counters": {k: v[1] for k, v in mf.items()},
                   "source_files": list(bench.EVAL_KERNEL_SOURCES), "source_sha16": bench.sources_sha16(bench.EVAL_KERNEL_SOURCES)},
                  open(os.path.join(out, ROUND + "_mfma_pmc.json"), "w"), indent=1)
    at = passes.get("tcc_atomic", {})
    # (the step's default form runs the hashed levels through grid_backward_kernel; VNR_AMD_TRAIN_OVERLAP=1: grid_backward_persistent_kernel)
    gb = kernel(at, "grid_backward_persistent_kernel") or kernel(at, "grid_backward_kernelI") or kernel(at, "grid_backward_kernel<")
    gl = kernel(at, "grid_backward_lds_kernel")
    if gb.get("TCC_EA0_ATOMIC_sum"):
        json.dump({"what": "TCC_EA0_ATOMIC_sum per dispatch (one dispatch of each kernel per training step) of the C4 model's step, 65 536 samples",
                   "requests_per_step": {"grid_backward_persistent_kernel (or grid_backward_kernel)": round(gb["TCC_EA0_ATOMIC_sum"][1]), "grid_backward_lds_kernel": round(gl.get("TCC_EA0_ATOMIC_sum", (0, 0))[1]),
                                         "total": round(gb["TCC_EA0_ATOMIC_sum"][1] + gl.get("TCC_EA0_ATOMIC_sum", (0, 0))[1])},
                   "dispatches": gb["TCC_EA0_ATOMIC_sum"][0],
                   "source_files": list(TRAIN_SOURCES), "source_sha16": bench.sources_sha16(TRAIN_SOURCES)},
                  open(os.path.join(out, ROUND + "_train_atomic_pmc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
