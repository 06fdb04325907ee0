"""Development-time check (this container only: it reads /root/reference/api.h as text): every function api.h declares, by name, number of
parameters and the parameters' types (namespace prefixes, parameter names, defaults and white space dropped), is defined by
include/vnr_api_shim.hpp with the same list.  usage: python tools/api_surface_vs_reference.py"""
import os
import re
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/api.h"
if not os.path.exists(REF):
    print("the reference is not here"); sys.exit(0)


def strip(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "<(":
            depth += 1
        if ch in ">)":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def ptype(p):
    p = p.split("=")[0].strip()
    p = re.sub(r"\b(vnr|std)::", "", p)
    # drop a trailing parameter name (an identifier after the type), keep pointers / references
    m = re.match(r"^(.*?[\w>\*&])\s+(\w+)$", p)
    if m and m.group(2) not in ("int", "float", "double", "bool", "char", "size_t", "uint8_t"):
        p = m.group(1)
    return re.sub(r"\s+", "", p)


RETURNS = {}


def signatures(text, definitions):
    sigs = set()
    pat = r"([\w:<>\*&]+(?:\s*[\*&])?)\s+(vnr[A-Z]\w*)\s*\(([^()]*(?:\([^()]*\)[^()]*)*)\)\s*" + (r"\{" if definitions else r";")
    for m in re.finditer(pat, strip(text)):
        sig = (m.group(2), tuple(ptype(p) for p in split_params(m.group(3))))
        sigs.add(sig)
        RETURNS[(definitions, sig)] = re.sub(r"\b(vnr|std)::", "", m.group(1)).replace(" ", "")
    return sigs


ref = signatures(open(REF).read(), False)
shim = signatures(open(os.path.join(ROOT, "include", "vnr_api_shim.hpp")).read(), True)
missing = sorted(s for s in ref if s not in shim)
by_arity = {(n, len(p)) for n, p in shim}
print(f"api.h declares {len(ref)} functions ({len({n for n, _ in ref})} names); the shim defines {len(shim)}")
for n, p in missing:
    print(("   same name and number of parameters, types written differently: " if (n, len(p)) in by_arity else "   MISSING: ") + n + "(" + ", ".join(p) + ")")
hard = [s for s in missing if (s[0], len(s[1])) not in by_arity]
ret_diff = [(s, RETURNS[(False, s)], RETURNS[(True, s)]) for s in sorted(ref) if s in shim and RETURNS[(False, s)] != RETURNS[(True, s)]]
for s_, a, b in ret_diff:
    print(f"   return type of {s_[0]}: {a} there, {b} here")
print(f"{len(ref) - len(missing) - len(ret_diff)} of {len(ref)} declarations have the same return type too")
hard += ret_diff
print(f"{len(ref) - len(missing)} of {len(ref)} declarations have a definition with the same parameter types; {len(hard)} have none with that name and arity")


# the enumerations of api.h (rendering modes, value types): the same enumerators in the same order
def enums(text):
    out = {}
    for m in re.finditer(r"\benum\s+(?:class\s+)?(\w+)\s*(?::\s*\w+\s*)?\{([^}]*)\}", strip(text)):
        out[m.group(1)] = [e.split("=")[0].strip() for e in m.group(2).split(",") if e.strip()]
    return out


ref_e, shim_e = enums(open(REF).read()), enums(open(os.path.join(ROOT, "include", "vnr_api_shim.hpp")).read())
for name, items in ref_e.items():
    ok = shim_e.get(name) == items
    hard += [] if ok else [name]
    print(f"{'ok ' if ok else 'BAD'} enum {name}: {len(items)} enumerators" + ("" if ok else f"; here {shim_e.get(name)}"))
sys.exit(1 if hard else 0)
