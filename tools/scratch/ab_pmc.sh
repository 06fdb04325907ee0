#!/bin/bash
# SQ instruction / cycle counters of the evaluation kernel for library variants: ab_pmc.sh <tag> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc
for t in "$@"; do
  export VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$t.so
  bash $R/tools/pmc_share.sh "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" 1 90 sq_$t > $R/gpurun_out/pmc/ab_$t.txt 2>&1
  python3 - $R/gpurun_out/pmc/ab_$t.txt $t <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
m = re.search(r"([\d.]+) ms per frame, ([\d.]+) M samples", txt)
vals = {k: float(v) for k, v in re.findall(r"fused_infer_kernel<2, 32, 0>\s+(\S+)\s+dispatches=\s*\d+ sum=([\d.e+]+)", txt)}
frames = 46
tiles = float(m.group(2)) * 1e6 * frames / 64 if m else 1
print(f"[{sys.argv[2]}] frame {m.group(1) if m else '?'} ms (profiled);", " ".join(f"{k[3:]}={v / tiles:.0f}/tile" for k, v in sorted(vals.items())))
PY
done
