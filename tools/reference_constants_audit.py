"""Development-time audit (this container only: it reads /root/reference as text): the constants and defaults this library took from the
reference, each checked on both sides by a pattern at the place the source comment cites.  A constant that drifted, or a citation that points
at nothing, fails here instead of in a frame nobody can compare.  usage: python tools/reference_constants_audit.py"""
import json
import os
import re
import sys
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.isdir(REF):
    print("the reference is not here"); sys.exit(0)
CS = "instantvnr_amd/csrc/"
# (what, reference file, pattern there, file here, pattern here)
ROWS = [
    ("samples per ray and iteration, the reference's default (here the library default is 24, measured faster; VNR_RM_N_ITERS pins either)",
     "core/renderer/method_raymarching.cu", r"int n_iters = 16;", CS + "render.hip", r'getenv\("VNR_RM_N_ITERS"\)'),
    ("a ray ends at alpha >= 0.9999", "core/instantvnr_types.h", r"#define nearly_one 0\.9999f", CS + "march_device.h", r"#define VNR_NEARLY_ONE 0\.9999f"),
    ("the far end of an unbounded interval", "core/instantvnr_types.h", r"#define float_large 1e20f", CS + "march_device.h", r"#define VNR_FLOAT_LARGE 1e20f"),
    ("an empty macrocell: |max opacity| <= epsilon", "core/renderer/method_raymarching.cu", r"fabsf\(r\) <= float_epsilon", CS + "march_device.h", r"fabsf\(r\) <= FLT_EPSILON"),
    ("adaptive step: 15 x the base step", "core/renderer/raytracing.h", r"scale = 15 \* base_sampling_step", CS + "march_device.h", r"scale = 15\.0f \* base_step"),
    ("adaptive step: opacity clamped to [0.1, 1]", "core/renderer/raytracing.h", r"clamp\(max_opacity, 0\.1f, 1\.f\)", CS + "march_device.h", r"clampf\(max_opacity, 0\.1f, 1\.0f\)"),
    ("macrocells of 16^3 voxels", "CMakeLists.txt", r"set\(MACROCELL_SIZE_MIP 4\)", CS + "volume.h", r"kMacrocellSizeMip = 4"),
    ("training batch of 65 536 samples", "core/network.cu", r"m_batch_size = 1 << 16", CS + "volume.h", r"batch_size_ = 1u << 16"),
    ("sampler stream: pcg32 seeded with 1337", "core/samplers/neural_sampler.cu", r"rng\{ 1337 \}", CS + "volume.h", r"rng_seed_ = 1337"),
    ("rendering mode of a new renderer", "api.cpp", r"set_rendering_mode\(5\)", CS + "renderer.h", r"int mode_ = 5;"),
    ("out-of-core sampler: 1024 slabs replaced per step", "core/samplers/neural_sampler.cpp", r"int VNR_NUM_CONCURRENT_BLOCKS = 1024;", CS + "capi.cpp", r"ncb = 1024;"),
    ("out-of-core sampler: 64 x as many resident", "core/samplers/neural_sampler.cpp", r"VNR_NUM_BLOCKS = VNR_NUM_CONCURRENT_BLOCKS\*64;", CS + "capi.cpp", r"nb = ncb \* 64;"),
    ("gradient shading flips a step that would leave [0, 1]", "core/renderer/raytracing.h", r"ext\.x > 1\.f-float_epsilon", CS + "render.hip", r"c\.x \+ stp\.x > 1\.0f - FLT_EPSILON"),
    ("isosurface: a corner is inside when value <= isovalue", "core/marching_cube.cu", r"values\[i\] <= volume\.isovalue", CS + "marching_cubes.hip", r"<= "),
]
bad = 0
for what, rf, rp, mf, mp in ROWS:
    r = re.search(rp, open(os.path.join(REF, rf)).read()) is not None
    m = re.search(mp, open(os.path.join(ROOT, mf)).read()) is not None
    bad += not (r and m)
    print(f"{'ok ' if r and m else 'BAD'} {what}: {rf} {'has' if r else 'LACKS'} /{rp}/, {mf} {'has' if m else 'LACKS'} /{mp}/")
# the optimizer defaults of the reference's example model against the library's defaults (network.h)
model = json.loads("\n".join(l for l in open(os.path.join(REF, "example-model.json")).read().splitlines() if not l.lstrip().startswith("//")))   # (the file carries // comments)
text = open(os.path.join(ROOT, CS, "network.h")).read()
opt = model["optimizer"]
nested = opt.get("nested", opt)
want = {"learning_rate": nested["learning_rate"], "beta1": nested["beta1"], "beta2": nested["beta2"], "epsilon": nested["epsilon"], "l2_reg": nested["l2_reg"],
        "decay_start": opt["decay_start"], "decay_interval": opt["decay_interval"], "decay_base": opt["decay_base"]}
for k, v in want.items():
    m = re.search(r"\b%s = ([0-9.eE+-]+)f?\b" % k, text)
    ok = m is not None and abs(float(m.group(1)) - float(v)) <= 1e-12 * max(1.0, abs(float(v)))
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} optimizer default {k}: example-model.json {v}, network.h {m.group(1) if m else 'not found'}")
# params.json: the keys network.cu:827-939 writes and reads, against every string of this library's sources
import glob  # noqa: E402


def literals(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return set(re.findall(r'"([A-Za-z_][\w .\-]*)"', re.sub(r"//[^\n]*", " ", text)))


ref_keys = literals("\n".join(open(os.path.join(REF, "core/network.cu")).read().split("\n")[820:945]))
ref_keys -= {"saving parameters to ", "total time", "mismatch macrocell dimension or spacing"}   # two log lines and a printf (the library compares the macrocell too: tests/test_cabi.py)
mine = set()
for f in glob.glob(os.path.join(ROOT, CS, "*.hip")) + glob.glob(os.path.join(ROOT, CS, "*.cpp")) + glob.glob(os.path.join(ROOT, CS, "*.h")):
    mine |= literals(open(f).read())
lack = sorted(ref_keys - mine)
bad += len(lack)
print(f"{'ok ' if not lack else 'BAD'} params.json: {len(ref_keys)} keys and messages of network.cu:827-939, {len(ref_keys) - len(lack)} of them in this library's sources" + (f"; missing {lack}" if lack else ""))
print(f"{len(ROWS) + len(want) - min(bad, len(ROWS) + len(want))} of {len(ROWS) + len(want)} constants agree")
sys.exit(1 if bad else 0)
