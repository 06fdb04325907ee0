"""Development-time check (this container only: it reads /root/reference/serializer.cpp as text): every JSON key and enumeration string the
reference's scene reader mentions is a string of this library's reader (csrc/scene.cpp) too, or is listed below with the reason it is not.
usage: python tools/scene_keys_vs_reference.py"""
import os
import re
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/serializer.cpp"
if not os.path.exists(REF):
    print("the reference is not here"); sys.exit(0)


def literals(path):
    t = open(path).read()
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    t = re.sub(r"//[^\n]*", " ", t)
    return set(re.findall(r'"([A-Za-z_][\w .\-]*)"', t))


NOT_HERE = {
    "serializer.h": "an include",
    "fileUpperLeft": "read into a local the reference never uses (serializer.cpp:274, 301)",
    "LITTLE_ENDIAN": "the default; this reader tests for BIG_ENDIAN only, any other string is little endian there as here (enum mapping, :36-40)",
    "transferFunction": "decoded by OVR's tfn module, absent from the tree: the shim hands the node to the application's decoder (vnr_api_shim.hpp)",
}
ref, mine = literals(REF), literals(os.path.join(ROOT, "instantvnr_amd", "csrc", "scene.cpp"))
unexplained = sorted(k for k in ref - mine if k not in NOT_HERE)
print(f"{len(ref)} strings in the reference's reader, {len(ref & mine)} of them here, {len(ref - mine) - len(unexplained)} explained:")
for k in sorted(ref - mine):
    print(f"   {k}: {NOT_HERE.get(k, 'UNEXPLAINED')}")
sys.exit(1 if unexplained else 0)
