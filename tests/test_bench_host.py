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


_AGREE = """
import importlib.util, os, subprocess, sys
spec = importlib.util.spec_from_file_location("bench_under_test", sys.argv[1])
m = importlib.util.module_from_spec(spec)
sys.argv = ["bench.py"]
spec.loader.exec_module(m)
class R: pass
def fake_run(*a, **k):                      # the probe child of this rank: says ok or names a failed collective
    r = R(); r.returncode = 0 if os.environ["FAKE_CHILD_OK"] == "1" else 3
    r.stdout = "probe: 5 collectives ok" if r.returncode == 0 else "FAILED: all-gather (in place)"; r.stderr = ""
    return r
subprocess.run = fake_run
transport, probe = m.choose_transport()
print(transport, "|", probe["rccl"])
"""


@pytest.mark.parametrize("verdicts", [(True, True, True), (True, False, True), (False, False, False)])
def test_the_parents_of_the_probe_children_agree_on_one_transport(verdicts, tmp_path):
    """bench.py --gpus N: every rank's probe child reports for itself; the parents must all take RCCL or all take the shared-memory
    fallback (ranks that meet with different transports hang in the rendezvous: ADVICE r04).  Three ranks as three processes, the child's
    verdict faked per rank; a verdict file this rank left behind in an earlier, killed run must not speak for it."""
    import subprocess
    run_id = f"test{os.getpid()}_{'' .join('1' if v else '0' for v in verdicts)}"
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"vnr_bench_probe_29731_{run_id}")
    with open(f"{base}_1", "w") as f:       # stale: rank 1 said ok in a run that was killed
        f.write("ok")
    procs = []
    for rank in (1, 0, 2):                  # (rank 1 first: it removes its stale file before anybody polls)
        env = dict(os.environ)
        env.update({"RANK": str(rank), "WORLD_SIZE": "3", "MASTER_PORT": "29731", "VNR_BENCH_RUN_ID": run_id, "FAKE_CHILD_OK": "1" if verdicts[rank] else "0"})
        env.pop("VNR_AMD_DIST_TRANSPORT", None)
        procs.append((rank, subprocess.Popen([sys.executable, "-c", _AGREE, os.path.join(ROOT, "bench.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        if rank == 1:
            import time
            time.sleep(1.0)
    out = {}
    for rank, p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se[-2000:]
        out[rank] = so.strip().splitlines()[-1]
    want = "rccl" if all(verdicts) else "shm"
    assert all(o.split(" | ")[0] == want for o in out.values()), out
    if not all(verdicts):
        for rank, v in enumerate(verdicts):
            if v:
                assert "FAILED on another rank" in out[rank], out
    for rank in range(3):                   # every rank removes its file when it exits
        assert not os.path.exists(f"{base}_{rank}")
