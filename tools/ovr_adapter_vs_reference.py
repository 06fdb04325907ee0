"""Development-time check (this container only: it reads /root/reference/device/ as text): the OVR-facing half of ovr_plugin/device_nnvolume_amd.cpp
is not compiled here (OVR's headers are not in the image); every identifier it uses that is not this library's own must at least occur in the
reference's own plugin sources, so that a member name cannot be a typo nobody's compiler has seen.  usage: python tools/ovr_adapter_vs_reference.py"""
import glob
import os
import re
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.isdir("/root/reference/device"):
    print("the reference is not here"); sys.exit(0)
mine = open(os.path.join(ROOT, "ovr_plugin", "device_nnvolume_amd.cpp")).read()
code = mine[mine.index("#if defined(VNR_HAVE_OVR)"):]
code = re.sub(r"//[^\n]*", " ", re.sub(r"/\*.*?\*/", " ", code, flags=re.S))
code = re.sub(r'"[^"\n]*"', " ", code)
ref = "".join(open(f, errors="ignore").read() for f in glob.glob("/root/reference/device/*") if os.path.isfile(f))
KEYWORDS = set("""if else for while return const auto int float double bool void size_t uint32_t uint8_t char static inline struct class public private
override namespace using typedef new delete this nullptr true false std string vector shared_ptr make_shared runtime_error throw try catch unsigned long sizeof
defined endif include define ifdef ifndef""".split())
OWN = {"bytes", "dev_", "device_nnvolume_amd"}   # this file's locals
ids = sorted({i for i in re.findall(r"[A-Za-z_]\w*(?:::[A-Za-z_]\w*)*", code)
              if i not in KEYWORDS and len(i) > 2 and not i.startswith(("vnrAmd", "vnr_amd", "VNR_", "std::"))})
unknown = [i for i in ids if i not in OWN and i.split("::")[-1] not in ref]
print(f"{len(ids)} identifiers in the OVR-facing half; {len(ids) - len(unknown)} occur in the reference's device/ sources or are this file's own" + (f"; unknown: {unknown}" if unknown else ""))
sys.exit(1 if unknown else 0)
