"""kernel timeline of a few training steps from a rocprofv3 --kernel-trace csv: python tools/train_timeline.py trace.csv [first_step] [n_steps]"""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 150
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
# a step starts at the sampler kernel that precedes fused_infer_kernel<2, 32, 2> (the training forward)
starts = [i for i, r in enumerate(rows) if "fused_infer_kernel<2, 32, 2>" in r["Kernel_Name"]]
lo, hi = starts[first], starts[first + n]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    name = r["Kernel_Name"].split("(")[0].replace("void vnr::", "").replace("vnr::", "")[:48]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  q{r['Queue_Id']:>2}  grid {(r.get('Grid_Size') or r.get('Grid_Size_X') or '?'):>10}  {name}")
