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
