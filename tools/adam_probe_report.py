"""reads <dir>/probe_kernel_trace.csv written by `rocprofv3 --kernel-trace --output-format csv -o probe -- python3 tools/adam_probe.py`"""
import csv
import statistics as st
import sys

rows = list(csv.DictReader(open(sys.argv[1] + "/probe_kernel_trace.csv")))
def dur(sub):
    t = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows if sub in r["Kernel_Name"])
    return [d for _, d in t]
adam = dur("adam_kernel")[20:]          # 20 warm-up steps
a, b = adam[0::2], adam[1::2]
print("adam dispatches after warm-up:", len(adam))
print("normal step (65 536 samples): mean %.4f ms  min %.4f  max %.4f" % (st.mean(a), min(a), max(a)))
print("near-empty batch (sweep only): mean %.4f ms  min %.4f  max %.4f" % (st.mean(b), min(b), max(b)))
gb = dur("grid_backward")[20:]
print("grid_backward normal / tiny : mean %.4f / %.4f ms" % (st.mean(gb[0::2]), st.mean(gb[1::2])))
for name in ("weight_grad_kernel<64>", "weight_grad_kernel<32>", "weight_grad_kernel<16>"):
    d = dur(name)[20:]
    if d:
        print("%-24s normal / tiny : mean %.4f / %.4f ms" % (name, st.mean(d[0::2]), st.mean(d[1::2])))
