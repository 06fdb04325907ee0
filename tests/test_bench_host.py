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


def _run_bench(args, env_extra, timeout=180):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VNR_AMD_DIST_TRANSPORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_a_launcher_world_that_is_not_what_gpus_asks_for_is_refused():
    """`--gpus 8` inside a one-rank environment (or `--gpus 1` inside an eight-rank one) must not print a line at all: rc 2, before the
    library is opened (a bare `--gpus 8` used to render on one GPU and print n_gpus 1)"""
    for gpus, world in ((8, "1"), (1, "8"), (4, "2")):
        out = _run_bench(["--gpus", str(gpus), "--steps", "2"], {"WORLD_SIZE": world, "RANK": "0"})
        assert out.returncode == 2 and "refusing" in out.stderr and not out.stdout.strip(), (gpus, world, out.stderr[-500:])


_LAUNCH = """
import importlib.util, json, os, sys
if os.environ.get("RANK") is not None:       # a rank started by the launcher under test: say who I am and how I was started
    rank = int(os.environ["RANK"])
    rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VNR_BENCH_RUN_ID")}
    rec["argv"] = sys.argv[1:]
    print(json.dumps(rec), flush=True)
    sys.exit(int(os.environ.get("FAKE_FAIL_RANK", "-1")) == rank and 7 or 0)
"""


def test_bare_gpus_n_starts_n_ranks_with_a_launcher_environment(bench, tmp_path, monkeypatch, capfd):
    """the launching half of `python bench.py --gpus N` without a GPU: N children with RANK 0..N-1, one WORLD_SIZE / MASTER_PORT / run id for
    all, the arguments passed through; rank 0's stdout is the program's stdout, the other ranks' goes to stderr; the exit code is 0 only if
    every rank's is"""
    import json
    fake = tmp_path / "fake_rank.py"
    fake.write_text(_LAUNCH)
    monkeypatch.setattr(bench, "__file__", str(fake))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "4"])
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 8)

    class A:
        gpus = 3
    import signal
    handlers = {s: signal.getsignal(s) for s in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    request_restore = lambda: [signal.signal(s, h) for s, h in handlers.items()]   # (the launcher ends its ranks when it is signalled)
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(A)
    request_restore()
    assert e.value.code == 0
    out, err = capfd.readouterr()
    r0 = json.loads(out.strip())
    assert r0["RANK"] == "0" and r0["WORLD_SIZE"] == "3" and r0["LOCAL_WORLD_SIZE"] == "3" and r0["MASTER_ADDR"] == "127.0.0.1" and r0["argv"] == ["--gpus", "3", "--steps", "4"]
    others = [json.loads(l.split("] ", 1)[1]) for l in err.splitlines() if l.startswith("[rank ")]
    assert sorted(o["RANK"] for o in others) == ["1", "2"] and all(o["LOCAL_RANK"] == o["RANK"] for o in others)
    assert {o["MASTER_PORT"] for o in others} == {r0["MASTER_PORT"]} and {o["VNR_BENCH_RUN_ID"] for o in others} == {r0["VNR_BENCH_RUN_ID"]}
    # one rank fails -> the launcher's exit code is that rank's
    monkeypatch.setenv("FAKE_FAIL_RANK", "1")
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(A)
    request_restore()
    assert e.value.code == 7
    # fewer devices than ranks: the ranks are told to share devices over the host-staged transport, and the launcher says so
    monkeypatch.delenv("FAKE_FAIL_RANK")
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 1)
    capfd.readouterr()
    with pytest.raises(SystemExit):
        bench.launch_ranks(A)
    request_restore()
    out, err = capfd.readouterr()
    assert "NOT an N-GPU measurement" in err


def test_bare_gpus_2_without_a_device_fails_loudly_and_prints_no_line():
    """the whole program, bare, on a machine without a GPU: both ranks end with the library's error, the launcher with a non-zero code and
    no JSON line (never a one-GPU line for a two-GPU question)"""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    out = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"VNR_AMD_DIST_TIMEOUT": "20"})
    assert out.returncode != 0 and not any(line.startswith("{") for line in out.stdout.splitlines()), out.stdout[-500:]
    assert "exited with code" in out.stderr
