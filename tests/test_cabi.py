"""CPU-side checks of the drop-in boundary: libvnr_amd.so loads, exports every symbol include/vnr_amd.h
declares, and the host-only entry points (JSON text <-> BSON, handles, error convention) behave like the
reference's api.cpp.  No compute call is made here."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from instantvnr_amd import _lib, api


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    return _lib.lib()


def test_library_exports_every_declared_symbol(L):
    names = _lib.declared_symbols()
    assert len(names) > 80
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert b"gfx950" in L.vnrAmdVersion()


def test_header_cites_reference_for_api_functions():
    text = open(_lib.HEADER).read()
    for ref in ["api.h:28", "api.cpp:174-188", "api.cpp:206-220", "network.cu:1043-1052", "renderer.h:84-94"]:
        assert ref in text


def test_json_text_with_comments_to_bson_and_back(L):
    text = """{ // model, like example-model.json
      "loss": {"otype": "L1"}, /* block */ "n": 3, "big": 5000000000, "f": 0.5, "neg": -7,
      "arr": [1, 2.5, "x", null, true], "s": "h\\u00e9llo\\n" }"""
    b = api.json_to_bson(text)
    back = json.loads(api.bson_to_json_text(b))
    assert back == {"loss": {"otype": "L1"}, "n": 3, "big": 5000000000, "f": 0.5, "neg": -7,
                    "arr": [1, 2.5, "x", None, True], "s": "héllo\n"}
    bson = pytest.importorskip("bson")
    dec = bson.decode(b)
    assert dec["big"] == 5000000000 and dec["arr"] == [1, 2.5, "x", None, True]
    # byte-identical to pymongo's encoder when keys are sorted (nlohmann's std::map order)
    want = bson.encode({k: back[k] for k in sorted(back)})
    assert b == want


def test_bson_binary_roundtrip_matches_pymongo(L):
    bson = pytest.importorskip("bson")
    blob = np.arange(300, dtype=np.uint16).tobytes()
    doc = {"parameters": {"n_params": 300, "params_binary": bson.Binary(blob, 0), "params_type": "__half"},
           "volume": {"dims": {"x": 4, "y": 5, "z": 6}}}
    enc = bson.encode(doc)
    out = C.c_void_p()
    n = C.c_size_t()
    _lib.check(L.vnrAmdJsonConvert(enc, len(enc), api.JSON_BSON, api.JSON_BSON, C.byref(out), C.byref(n)))
    again = C.string_at(out, n.value)
    L.vnrAmdFreeHost(out)
    assert again == enc


def test_bson_reader_bounds_children_by_their_parent_and_limits_depth(L):
    """a corrupt or crafted params file: a child document that claims to reach beyond its parent, a string or binary that does, and
    nesting deeper than 64 levels are parse errors, not overruns or a blown stack"""
    import struct
    out, n = C.c_void_p(), C.c_size_t()

    def convert(b):
        return L.vnrAmdJsonConvert(b, len(b), api.JSON_BSON, api.JSON_TEXT, C.byref(out), C.byref(n))

    def doc(elements):   # elements: raw bytes of the element list
        return struct.pack("<i", 4 + len(elements) + 1) + elements + b"\0"

    inner = doc(b"\x10a\0" + struct.pack("<i", 7))
    good = doc(b"\x03d\0" + inner) + b"PADDINGPADDING"        # trailing bytes behind the top-level document are not its business
    assert convert(good[:len(good) - 14]) == 0
    L.vnrAmdFreeHost(out)
    # the child claims 8 more bytes than it has: they exist in the buffer (the padding) but lie outside the parent
    lying = bytearray(good)
    off = 4 + 3                                                  # parent length, element header "\x03d\0"
    struct.pack_into("<i", lying, off, len(inner) + 8)
    assert convert(bytes(lying)) != 0 and b"bson parse error" in L.vnrAmdGetLastError()
    # a string whose length runs past its document
    s = doc(b"\x02s\0" + struct.pack("<i", 50) + b"abc\0") + b"x" * 64
    assert convert(s) != 0 and b"bson parse error" in L.vnrAmdGetLastError()
    # 65 nested documents
    deep = doc(b"")
    for _ in range(65):
        deep = doc(b"\x03d\0" + deep)
    assert convert(deep) != 0 and b"nested too deeply" in L.vnrAmdGetLastError()
    ok = doc(b"")
    for _ in range(40):
        ok = doc(b"\x03d\0" + ok)
    assert convert(ok) == 0
    L.vnrAmdFreeHost(out)


def test_text_reader_refuses_what_nlohmann_refuses_and_limits_its_depth(L):
    """found by tests/test_json_fuzz.py: 200 000 nested '[' ran the recursive text parser off the stack; "1-2" parsed as 1; a lone
    \\ud800 became four bytes that are not UTF-8; a key with an embedded U+0000 was written into a BSON document it then corrupted
    (nlohmann::json: parse_error.101, parse_error.101, out_of_range.409)"""
    out, n = C.c_void_p(), C.c_size_t()

    def convert(b, fin=api.JSON_TEXT, fout=api.JSON_BSON):
        rc = L.vnrAmdJsonConvert(b, len(b), fin, fout, C.byref(out), C.byref(n))
        if rc == 0:
            L.vnrAmdFreeHost(out)
        return rc

    assert convert(b'{"a":' * 200000 + b"1" + b"}" * 200000) != 0 and b"nested too deeply" in L.vnrAmdGetLastError()
    assert convert(b"[" * 200000) != 0 and b"nested too deeply" in L.vnrAmdGetLastError()
    assert convert(b'{"a":' * 60 + b"1" + b"}" * 60) == 0
    for bad in (b'{"a": 1-2}', b'{"a": --3}', b'{"a": 1e}', b'{"a": .}', b'{"a": 1.2.3}', b'{"a": -}'):
        assert convert(bad) != 0 and b"malformed number" in L.vnrAmdGetLastError(), bad
    for good in (b'{"a": -0}', b'{"a": 1e-2}', b'{"a": 2.5E+3}', b'{"a": 99999999999999999999}'):
        assert convert(good) == 0, good
    for bad in (b'{"a": "\\ud800"}', b'{"a": "\\ud800\\u0041"}', b'{"a": "\\udc00"}'):
        assert convert(bad) != 0 and b"surrogate" in L.vnrAmdGetLastError(), bad
    assert json.loads(api.bson_to_json_text(api.json_to_bson('{"a": "\\ud83d\\ude00"}'))) == {"a": "\U0001F600"}
    assert convert(b'{"a\\u0000b": 1}') != 0 and b"U+0000" in L.vnrAmdGetLastError()
    assert convert(b'{"a\\u0000b": 1}', api.JSON_TEXT, api.JSON_TEXT) == 0     # (text has no such restriction)


def test_json_save_and_load_files(L, tmp_path):
    p = str(tmp_path / "m.json")
    api.vnrSaveJsonText({"a": 1, "b": [1, 2]}, p)
    assert api.vnrCreateJsonText(p) == {"a": 1, "b": [1, 2]}
    q = str(tmp_path / "m.bson")
    api.vnrSaveJsonBinary({"a": 1, "b": [1, 2]}, q)
    assert json.loads(api.bson_to_json_text(api.vnrCreateJsonBinary(q))) == {"a": 1, "b": [1, 2]}


def test_error_convention(L):
    out = C.c_void_p()
    n = C.c_size_t()
    bad = b"{ not json"
    assert L.vnrAmdJsonConvert(bad, len(bad), 0, 1, C.byref(out), C.byref(n)) != 0
    assert b"json parse error" in L.vnrAmdGetLastError()
    with pytest.raises(api.VnrAmdError):
        api.vnrCreateJsonText("/nonexistent/file.json")
    # params without volume.dims: api.cpp:214-216
    b = api.json_to_bson({"model": {}})
    assert not L.vnrAmdCreateNeuralVolumeFromParams(b, len(b), api.JSON_BSON)
    assert b"volume dims" in L.vnrAmdGetLastError()


def test_camera_and_tfn_handles(L):
    cam = api.vnrCreateCamera()
    api.vnrCameraSet(cam, (1, 2, 3), (0, 0, 0), (0, 1, 0))
    assert np.allclose(api.vnrCameraGetPosition(cam), (1, 2, 3))
    assert np.allclose(api.vnrCameraGetFocus(cam), 0) and np.allclose(api.vnrCameraGetUpVec(cam), (0, 1, 0))
    t = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(t, [[0, 0, 0], [1, 1, 1]])
    api.vnrTransferFunctionSetAlpha(t, [0.0, 0.5, 1.0])
    api.vnrTransferFunctionSetValueRange(t, (0, 1))
    nc, na = C.c_int(), C.c_int()
    _lib.check(L.vnrAmdTransferFunctionGetSizes(t.h, C.byref(nc), C.byref(na)))
    assert (nc.value, na.value) == (2, 3)


def test_no_cpu_fallback_without_device(L):
    if L.vnrAmdHasDevice():
        pytest.skip("a GPU is present")
    with pytest.raises(api.VnrAmdError):
        api.vnrCreateNeuralVolume({"encoding": {"otype": "HashGrid"}, "network": {"otype": "FullyFusedMLP", "n_neurons": 64}},
                                  (8, 8, 8))


def test_constants_taken_from_the_reference_are_still_what_the_reference_says(L):
    """tools/reference_constants_audit.py, where the reference's sources are present (this container, not the GPU box): 22 constants and
    defaults (ray termination, empty-cell test, adaptive step, macrocell size, batch size, sampler seed, default mode, out-of-core slab
    counts, the example model's optimizer) are found by pattern on both sides; and the reference's own example-model.json, // comments and
    all, goes through this library's JSON reader"""
    import subprocess
    import sys
    if not os.path.exists("/root/reference/example-model.json"):
        pytest.skip("the reference's sources are not on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "reference_constants_audit.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "22 of 22 constants agree" in out.stdout, out.stdout[-1500:]
    model = api.vnrCreateJsonText("/root/reference/example-model.json")
    assert model["encoding"]["otype"] == "HashGrid" and model["encoding"]["n_levels"] == 8 and model["encoding"]["log2_hashmap_size"] == 19
    assert model["network"] == {"otype": "FullyFusedMLP", "activation": "ReLU", "n_neurons": 64, "n_hidden_layers": 4, "output_activation": "None"}
    assert model["loss"]["otype"] == "L1" and model["optimizer"]["nested"]["otype"] == "Adam"


def test_the_shim_defines_every_declaration_of_the_reference_s_api_h():
    """tools/api_surface_vs_reference.py, where the reference's api.h is present: each of its declarations (name, parameter types, return type;
    namespace prefixes, parameter names and defaults dropped) has a definition in include/vnr_api_shim.hpp"""
    import subprocess
    import sys
    if not os.path.exists("/root/reference/api.h"):
        pytest.skip("the reference's sources are not on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "api_surface_vs_reference.py")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "59 of 59 declarations have a definition with the same parameter types" in out.stdout, out.stdout[-1500:]
    assert "59 of 59 declarations have the same return type too" in out.stdout and "ok  enum vnrRenderMode: 17 enumerators" in out.stdout


def test_bench_and_smoke_fail_loudly_without_a_device(L):
    """bench.py and __graft_entry__.smoke() are the product path: without a GPU they end with the library's error, not with numbers from a
    CPU stand-in (the oracle is reachable from bench.py only as the `cpu_baseline` leg, after the timed region)"""
    if L.vnrAmdHasDevice():
        pytest.skip("a GPU is present")
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "no HIP capable devices" in out.stderr and not any(line.startswith("{") for line in out.stdout.splitlines())
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=root, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "smoke ok" not in out.stdout


def _build_and_run_shim_smoke(tmp_path, bson=False):
    """bson: the shim's production branch (json::to_bson / from_bson) against tests/compat/json/json.hpp, a test-only stand-in for the
    nlohmann >= 3.8 of a reference application (the image has 3.1.1, which has no BSON)"""
    import shutil
    import subprocess
    inc = "/opt/conda/include"
    if not os.path.exists(os.path.join(inc, "json.hpp")) or not shutil.which("g++"):
        pytest.skip("no nlohmann::json / g++ in this image")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / ("shim_smoke_bson" if bson else "shim_smoke"))
    extra = ["-DVNR_SHIM_SMOKE_BSON", "-I", os.path.join(root, "tests", "compat")] if bson else []
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(root, "include")] + extra + ["-I", inc,
                           os.path.join(root, "tests", "shim_smoke.cpp"), "-o", exe,
                           "-L", os.path.join(root, "instantvnr_amd"), "-lvnr_amd",
                           "-Wl,-rpath," + os.path.join(root, "instantvnr_amd")])
    fixture = str(tmp_path / "bson_params_like.bson")
    shutil.copy(os.path.join(root, "tests", "golden", "bson_params_like.bson"), fixture)
    return subprocess.call([exe], env=dict(os.environ, VNR_SHIM_SMOKE_FIXTURE=fixture))


def test_cpp_api_shim_compiles_and_throws_like_the_reference(L, tmp_path):
    """include/vnr_api_shim.hpp restores the api.h C++ signatures over the C-ABI.  Compiled here against the only
    nlohmann::json in the image (3.1.1, no BSON => VNR_SHIM_JSON_TEXT_TRANSPORT); errors surface as std::runtime_error."""
    rc = _build_and_run_shim_smoke(tmp_path)
    assert rc == (0 if L.vnrAmdHasDevice() else 42)   # 42 = std::runtime_error ("no HIP capable devices")


def test_cpp_api_shim_production_bson_branch_compiles_and_round_trips(L, tmp_path):
    """the branch a reference application takes (no VNR_SHIM_JSON_TEXT_TRANSPORT: json::to_bson / from_bson, api.cpp:23-47)
    type-checks and runs: the params-like fixture keeps its bytes through from_bson -> to_bson and through vnrSave/LoadJsonBinary, and the
    library's BSON codec reads what the shim hands it"""
    rc = _build_and_run_shim_smoke(tmp_path, bson=True)
    assert rc == (0 if L.vnrAmdHasDevice() else 42)


@pytest.mark.gpu
@pytest.mark.parametrize("bson", [False, True])
def test_cpp_api_shim_runs_its_gpu_half(tmp_path, bson):
    """the same program where a GPU exists (the driver's `-m gpu` run): a C++ application written against api.h creates a neural volume
    and a volume from memory (vnrType), renders, decodes and renders the decoded volume through the shim; exit code 0.  bson: the production
    branch, which also serialises the volume's params into a json and creates a second volume from that json (api.cpp:206-220)"""
    assert _build_and_run_shim_smoke(tmp_path, bson=bson) == 0


@pytest.mark.parametrize("order,ok", [("lib-first", False), ("torch-first", True)])
def test_torch_must_be_imported_before_the_library_is_loaded(L, order, ok):
    """PyTorch's wheel bundles its own ROCm runtime; with the library loaded first torch finds no GPU (measured on
    MI355X: tools/repro_torch_after_lib.py).  The wrong order must fail with an explanation, not later inside torch."""
    import subprocess
    import sys
    first, second = ("_lib.lib()", "import torch") if order == "lib-first" else ("import torch", "_lib.lib()")
    code = (f"import sys; sys.path.insert(0, {_lib.ROOT!r})\n"
            "from instantvnr_amd import _lib\n"
            f"{first}\n{second}\n"
            "try:\n    _lib.require_torch_loaded_first(); print('GUARD-PASSED')\n"
            "except _lib.VnrAmdError as e:\n    print('GUARD-RAISED', 'Import torch first' in str(e))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout
    assert ("GUARD-PASSED" in out) if ok else ("GUARD-RAISED True" in out), out


def test_view_model_tool_reads_and_corrects_params_files(tmp_path):
    """tools/view_model.py (apps/view_model.cpp): what a params.json holds; --correct --dims adds the missing volume dims to a
    legacy file and writes params-corrected.json"""
    import subprocess
    import sys
    import bson
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "view_model.py")
    fixture = os.path.join(root, "tests", "golden", "bson_params_like.bson")
    out = subprocess.run([sys.executable, tool, fixture], capture_output=True, text=True, cwd=tmp_path)
    assert out.returncode == 0, out.stderr
    assert "[info] volume dims: (32, 32, 32)" in out.stdout and "[info] params = 1.52 KB" in out.stdout and '"otype"' in out.stdout
    doc = bson.decode(open(fixture, "rb").read())
    del doc["volume"]
    legacy = tmp_path / "legacy.bson"
    legacy.write_bytes(bson.encode(doc))
    out = subprocess.run([sys.executable, tool, str(legacy), "--correct", "--dims", "10,20,30"], capture_output=True, text=True, cwd=tmp_path)
    assert out.returncode == 0 and "does not contain dimension data" in out.stdout
    fixed = bson.decode((tmp_path / "params-corrected.json").read_bytes())
    assert fixed["volume"]["dims"] == {"x": 10, "y": 20, "z": 30} and fixed["parameters"] == doc["parameters"]
