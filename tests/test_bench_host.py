"""Host logic of bench.py that runs without a GPU: the stamp that decides whether a committed counter result still describes the kernel it is
printed beside.  (The level table the training bound is priced with comes from the library, vnrAmdNeuralVolumeLevelTable: tests/test_gpu_network.py.)"""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_the_source_stamp_follows_the_code_not_its_comments(bench, tmp_path, monkeypatch):
    """counter results enter the bench line only while `source_sha16` of their JSON equals the stamp of the files the kernel is compiled from
    NOW: a changed statement changes the stamp, a changed comment or blank line does not"""
    d = tmp_path / "instantvnr_amd" / "csrc"
    d.mkdir(parents=True)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (d / "k.h").write_text("// header\nint f(int x)\n{\n  return x + 1;   // plus one\n}\n")
    a = bench.sources_sha16(["k.h"])
    (d / "k.h").write_text("// another header, longer\n\nint f(int x)\n{\n  return x + 1;\n}\n\n")
    assert bench.sources_sha16(["k.h"]) == a
    (d / "k.h").write_text("int f(int x)\n{\n  return x + 2;\n}\n")
    assert bench.sources_sha16(["k.h"]) != a
    assert len(a) == 16
