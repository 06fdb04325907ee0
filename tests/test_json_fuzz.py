"""the library's JSON / BSON reader and writer (csrc/json.cpp; the reference reads its scene, model and params.json files with
nlohmann::json) under random documents and random damage, in a child process: no crash, no hang, an error message with every refusal,
and whatever is accepted survives a round trip.  CPU only."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("seed", [1, 2])
def test_parsers_survive_random_documents_and_random_damage(seed):
    env = dict(os.environ, PYTHONPATH=os.path.dirname(HERE) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    n = int(os.environ.get("VNR_JSON_FUZZ", "4000"))
    r = subprocess.run([sys.executable, os.path.join(HERE, "json_fuzz_worker.py"), str(seed), str(n)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    counts = json.loads(r.stdout.strip().splitlines()[-1])
    assert counts["round trips"] == (n + 3) // 4
    # the damage is neither always fatal nor never: both branches of both readers ran
    for k in ("text ok", "text error", "bson ok", "bson error"):
        assert counts[k] > n // 50, counts


def test_parsers_are_clean_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """the same stream of documents and damage through csrc/json.cpp compiled with -fsanitize=address,undefined (CPU build: the GPU boxes
    have no sanitizer runtime): an out-of-bounds read that happens to land in mapped memory does not crash the .so, but it stops here"""
    env = dict(os.environ, PYTHONPATH=os.path.dirname(HERE) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    corpus = str(tmp_path / "corpus.bin")
    r = subprocess.run([sys.executable, os.path.join(HERE, "json_fuzz_worker.py"), "3", "2500", corpus], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    exe = str(tmp_path / "json_asan")
    csrc = os.path.join(os.path.dirname(HERE), "instantvnr_amd", "csrc")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + csrc,
                        os.path.join(HERE, "json_asan_harness.cpp"), os.path.join(csrc, "json.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    h = subprocess.run([exe, corpus], capture_output=True, text=True, timeout=600)
    assert h.returncode == 0, (h.returncode, h.stderr[-3000:])
    parsed, refused = (int(x) for x in h.stdout.split() if x.isdigit())
    assert parsed > 500 and refused > 500, h.stdout


def test_scene_readers_are_clean_under_sanitizers_on_damaged_scenes(tmp_path):
    """scene documents in both of the reference's formats with keys deleted or replaced by values of the wrong type, sign or size, through
    parse_scene_volume / parse_scene_camera / parse_scene_tfn_range (csrc/scene.cpp, host code) built with -fsanitize=address,undefined"""
    import copy
    import struct

    import numpy as np
    vidi = {"dataSource": [{"format": "REGULAR_GRID_RAW_BINARY", "fileName": "/nonexistent/a.raw", "dimensions": {"x": 20, "y": 12, "z": 9},
                            "type": "UNSIGNED_BYTE", "offset": 16, "endian": "LITTLE_ENDIAN"}],
            "view": {"camera": {"eye": {"x": 10.0, "y": 20.0, "z": -300.0}, "center": {"x": 16.0, "y": 8.0, "z": 4.0}, "up": {"x": 0.0, "y": 1.0, "z": 0.0}, "fovy": 42.0},
                     "volume": {"transferFunction": {}, "scalarMappingRange": {"minimum": 0.25, "maximum": 0.5},
                                "scalarMappingRangeUnnormalized": {"minimum": 3.0, "maximum": 200.0}}}}
    diva = {"version": "DIVA", "volume": {"dims": {"x": 20, "y": 12, "z": 9}, "type": "FLOAT", "offset": 16, "range": {"x": 3.0, "y": 200.0}, "filename": "/nonexistent/a.raw"}}
    pool = [0, -1, 1, 17, 2 ** 31, 2 ** 40, 0.5, -2.5, 1e30, "x", "", "FLOAT", "DOUBLE", "BIG_ENDIAN", "VIDI3D", "DIVA", None, True, False, [], {}, [1, 2], {"x": 1},
            {"x": -1, "y": 2, "z": 3}, ["a.raw", "b.raw"], [[]], {"minimum": "a"}]

    def paths(doc, prefix=()):
        out = []
        if isinstance(doc, dict):
            for k, v in doc.items():
                out.append(prefix + (k,)); out += paths(v, prefix + (k,))
        elif isinstance(doc, list):
            for k, v in enumerate(doc):
                out.append(prefix + (k,)); out += paths(v, prefix + (k,))
        return out

    rng = np.random.default_rng(9)
    corpus = str(tmp_path / "scenes.bin")
    with open(corpus, "wb") as f:
        for i in range(4000):
            sc = copy.deepcopy(vidi if rng.uniform() < 0.6 else diva)
            for _ in range(int(rng.integers(0, 4))):
                ps = paths(sc)
                if not ps:
                    break
                p = ps[int(rng.integers(0, len(ps)))]
                node = sc
                for k in p[:-1]:
                    node = node[k]
                if rng.uniform() < 0.3:
                    del node[p[-1]]
                else:
                    node[p[-1]] = copy.deepcopy(pool[int(rng.integers(0, len(pool)))])
            text = json.dumps(sc).encode()
            f.write(struct.pack("<BI", 0, len(text)) + text)
    exe = str(tmp_path / "scene_asan")
    csrc = os.path.join(os.path.dirname(HERE), "instantvnr_amd", "csrc")
    # the host compiler on the host side of the two files (the HIP headers are plain C++ to it): a CPU build, nothing of it runs on a GPU
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-I" + csrc, os.path.join(HERE, "scene_asan_harness.cpp"), os.path.join(csrc, "scene.cpp"),
                        os.path.join(csrc, "json.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    h = subprocess.run([exe, corpus], capture_output=True, text=True, timeout=600)
    assert h.returncode == 0, (h.returncode, h.stderr[-3000:])
    ok, refused = (int(x) for x in h.stdout.split() if x.isdigit())
    assert ok > 300 and refused > 300, h.stdout
