"""More than one rank on real kernels, on the ONE GPU of a test box.

RCCL refuses two ranks on one device, so these tests start 2 and 4 processes (tests/dist_gpu_worker.py) that share the GPU and
exchange through the library's host-staged "shm" transport: the rendezvous, the renderer's distributed mode (interleaved tile
rows rendered into the rank's slot of the gathered buffer, in-place all-gather, de-interleave, pipelined frames), data-parallel
training (replica broadcast, per-rank sample streams, fp16 gradient exchange range by range, range-wise Adam), the macrocell merge
and the out-of-core sampler's per-rank slab sets are the code that runs on 8 GPUs; only the bytes travel differently.
(A GPU box admits at most 6 GPU processes of one user: 4 ranks + this process is the largest world these tests use.)
The "rccl" transport itself runs here as a one-rank communicator: librccl is opened with dlopen, initialised, and every
collective is issued on the library's device buffers and streams.
Also here: asynchronous frames (vnrAmdRendererSetAsync), the host half of the pipeline."""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from instantvnr_amd import api
from instantvnr_amd import dist as vdist
from instantvnr_amd import synthetic as syn
from instantvnr_amd._lib import check, lib

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_gpu_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(scenario, world, tmp_path, transport="shm", extra_env=None, timeout=420):
    port = _free_port()
    procs, outs = [], []
    for rank in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "VNR_AMD_DIST_TRANSPORT": transport, "VNR_AMD_DIST_TIMEOUT": "120",
                    "VNR_AMD_DIST_FORCE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.update(extra_env or {})
        out = str(tmp_path / f"{scenario}_{world}_{rank}.npz")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, WORKER, scenario, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=timeout)[0].decode("utf-8", "replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, log) in enumerate(zip(procs, logs)):
        assert p.returncode == 0, f"rank {rank} of {world} failed:\n{log[-3000:]}"
    return [dict(np.load(o, allow_pickle=False)) for o in outs]


# ------------------------------------------------------------------------------------------------ tiles
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_frames_equal_the_unsharded_frame(world, tmp_path):
    res = run_ranks("frames", world, tmp_path)
    for r in res:
        assert int(r["world"]) == world and str(r["transport"]) == "shm"
        for case in range(5):
            assert bool(r[f"case{case}_sync"]), (int(r["rank"]), case, "synchronous MapFrame")
            assert bool(r[f"case{case}_pipe"]), (int(r["rank"]), case, "pipelined frames")
            assert float(r[f"case{case}_coverage"]) > 0.1
            assert int(r[f"case{case}_samples"]) > 0 or case == 3   # this rank did march its share (case 3: the monolithic marcher keeps no statistics)


# ------------------------------------------------------------------------------------------------ data-parallel training
def _reference_concatenated(world, steps, vol):
    """one process, one step per DP step on the CONCATENATED batch of all ranks (their pcg32 streams), rank 0's initial parameters"""
    os.environ["VNR_AMD_INIT_SEED"] = "100"
    sv = api.vnrCreateSimpleVolume(vol)
    nv = api.vnrCreateNeuralVolume(syn.model_config(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2),
                                   sv, online_macrocell_construction=False)
    samplers = []
    for r in range(world):
        s = api.vnrCreateSimpleVolume(vol)
        # the stream train_data_parallel gives rank r (volume.h set_sampler_rank): default stream + r; a SimpleVolume has no
        # seed call of its own in the C-ABI, so each stream gets a carrier neural volume
        carrier = api.vnrCreateNeuralVolume(syn.model_config(n_levels=2, n_features=2, log2_hashmap_size=8, base_resolution=4, n_hidden_layers=1), s)
        check(lib().vnrAmdNeuralVolumeSetSamplerSeed(carrier.h, 1337, 0xda3e39cb94b95bdb + r))
        samplers.append((s, carrier))
    for _ in range(steps):
        cs, vs = zip(*[api.simple_volume_take_samples(s, 65536) for s, _ in samplers])
        api.neural_forward_backward(nv, np.concatenate(cs), np.concatenate(vs))
        api.neural_train_end(nv, 1.0, True)
    return api.neural_get_params_fp16(nv).astype(np.float32), api.vnrNeuralVolumeGetPSNR(nv)


@pytest.mark.parametrize("world", [2, 4])
def test_data_parallel_training_on_several_ranks(world, tmp_path):
    steps = 20
    res = run_ranks("train", world, tmp_path, extra_env={"TEST_STEPS": str(steps)})
    # replicas started from different seeds ...
    assert len({int(r["checksum_before"]) for r in res}) == world
    # ... and are identical after training, in both forms of the step
    assert len({int(r["checksum"]) for r in res}) == 1
    assert len({int(r["checksum_by_hand"]) for r in res}) == 1
    assert all(int(r["step"]) == steps for r in res)
    p = res[0]["params"].view(np.float16).astype(np.float32)
    hand = res[0]["params_by_hand"].view(np.float16).astype(np.float32)
    want, psnr_ref = _reference_concatenated(world, steps, syn.analytic_volume(32))
    moved = np.abs(want - _initial_params()).mean()
    # Gradients are accumulated in half precision with packed atomics, as tcnn does (grid.h: grad_t = __half for F > 1): the one
    # process adds world x 65 536 samples into these small tables in fp16, the ranks add 65 536 each and the exchange sums them in
    # fp32, so the two differ by the rounding of a few thousand fp16 adds per entry (and by the order of the atomics).  Relative to
    # how far 20 steps move the parameters: measured 1.6 % at world 2, 3.4 % at world 4 (< 2 % with the fp32 blob of round 1).
    assert np.abs(p - want).mean() < 0.06 * moved + 1e-5, (np.abs(p - want).mean(), moved)
    assert np.abs(hand - want).mean() < 0.06 * moved + 1e-5
    assert abs(float(res[0]["psnr"]) - psnr_ref) < 1.0


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_optimizer_step_equals_the_replicated_step(world, tmp_path):
    """reduce-scatter + Adam on 1/world + all-gather against all-reduce + Adam on everything, on identical gradients (world 3: ranges
    that do not divide, so their last parameters are all-reduced and updated by everyone)"""
    res = run_ranks("sharded_optimizer", world, tmp_path)
    assert len({int(r["checksum"]) for r in res}) == 1          # one model on all ranks
    for r in res:
        assert bool(np.all(r["equal"])), (int(r["rank"]), r["equal"])
        assert r["moved"][0] > 0.2 and r["moved"][-1] >= r["moved"][0]   # the steps did update (40 % of the entries carry a gradient per rank)
        assert bool(r["refused"]) and bool(r["equal_after_gather"]) and int(r["step"]) == 6


ODD_MODELS = [
    # parameter counts and level offsets that are odd, not multiples of 8, or smaller than the world: the 8-aligned slices of the
    # reduce-scatter, the all-reduced remainder and the compacting optimizer's ranges [lo, hi) that start and end inside a group of eight
    {"model": dict(n_levels=7, n_features=1, log2_hashmap_size=10, base_resolution=5, n_hidden_layers=2), "encoding": {"type": "Tiled"}, "network": {"n_neurons": 16}},
    {"model": dict(n_levels=3, n_features=8, log2_hashmap_size=9, base_resolution=3, n_hidden_layers=1), "encoding": {}, "network": {"n_neurons": 128}},
    {"model": dict(n_levels=5, n_features=2, log2_hashmap_size=12, base_resolution=3, n_hidden_layers=3, per_level_scale=1.5), "encoding": {"type": "Dense"},
     "network": {"n_neurons": 32, "activation": "Sigmoid"}},
]


@pytest.mark.parametrize("world,model", [(3, 0), (4, 0), (3, 1), (4, 2)])
def test_sharded_optimizer_step_on_models_of_odd_sizes(world, model, tmp_path):
    import json
    res = run_ranks("sharded_optimizer", world, tmp_path, extra_env={"TEST_MODEL_JSON": json.dumps(ODD_MODELS[model])})
    assert len({int(r["checksum"]) for r in res}) == 1
    for r in res:
        assert bool(np.all(r["equal"])), (int(r["rank"]), int(r["n_params"]), r["equal"])
        assert r["moved"][0] > 0.2
        assert bool(r["refused"]) and bool(r["equal_after_gather"]) and int(r["step"]) == 6
    assert int(res[0]["n_params"]) % 8 != 0 or model != 0      # (the first model's count is odd: 7 levels of 125 entries)


def test_replicas_are_synchronised_again_after_one_rank_changes_its_parameters(tmp_path):
    res = run_ranks("resync", 2, tmp_path)
    a, b = res
    assert int(a["checksum_a"]) == int(b["checksum_a"])
    assert int(a["checksum_changed"]) != int(b["checksum_changed"])     # rank 1 did load something else ...
    assert int(a["checksum_b"]) == int(b["checksum_b"])                 # ... and the next call made the replicas one again
    assert int(a["checksum_c"]) == int(b["checksum_c"]) and int(a["checksum_c"]) != int(a["checksum_b"])
    assert int(a["step"]) == int(b["step"]) == 2                        # rank 0's re-created model restarts the schedule for everyone


def test_shares_of_an_eighth_frame_size_take_the_unpinned_branch_and_equal_the_frame_at_32(tmp_path):
    """VNR_RM_N_ITERS unset (tests/conftest.py pins 16 for the rest of the suite): 2 ranks x 196 608 pixels, i.e. what a rank of the
    8-GPU bench renders: N_ITERS 32, 4 ray parts, packing fused into the evaluation kernel"""
    saved = os.environ.pop("VNR_RM_N_ITERS", None)
    try:
        res = run_ranks("frames_unpinned", 2, tmp_path, timeout=600)
    finally:
        if saved is not None:
            os.environ["VNR_RM_N_ITERS"] = saved
    for r in res:
        assert float(r["coverage"]) > 0.1 and int(r["samples"]) > 500_000   # (a 1024 x 384 frame of a cube: 16 % of the pixels hit it)
        assert bool(np.all(r["equal_32"])) and bool(r["equal_32_sync"]), (int(r["rank"]), r["equal_32"])
        # the branch was taken: as many iterations as the frame pinned to 32 needs, fewer than at 24; against the frame at 24 only last bits move
        assert int(r["iterations"]) <= int(r["iterations_32"]) < int(r["iterations_24"])
        assert not bool(np.all(r["equal_24"])) and float(r["max_diff_24"]) < 1e-3
        print(f"rank {int(r['rank'])}: shares vs the unsharded frame at 24: equal {r['equal_24']}, max |diff| {float(r['max_diff_24']):.2e}")


def test_data_parallel_training_with_one_feature_per_level(tmp_path):
    """n_features_per_level = 1 sums its grid gradients in fp32 (a float image folded into the fp16 blob, network_train.hip): in the
    data-parallel step the fold of a level range must come before that range's exchange.  Two ranks from different seeds end with one model,
    in both forms of the step, and the loss falls."""
    res = run_ranks("train", 2, tmp_path, extra_env={"TEST_STEPS": "30", "TEST_FEATURES": "1"})
    assert len({int(r["checksum_before"]) for r in res}) == 2
    assert len({int(r["checksum"]) for r in res}) == 1 and len({int(r["checksum_by_hand"]) for r in res}) == 1
    assert all(int(r["step"]) == 30 for r in res)
    p = res[0]["params"].view(np.float16).astype(np.float32)
    hand = res[0]["params_by_hand"].view(np.float16).astype(np.float32)
    assert np.isfinite(p).all() and np.abs(p - hand).mean() < 0.05 * np.abs(p).mean() + 1e-5
    assert float(res[0]["psnr"]) > 15.0, float(res[0]["psnr"])


def _initial_params():
    os.environ["VNR_AMD_INIT_SEED"] = "100"
    sv = api.vnrCreateSimpleVolume(syn.analytic_volume(32))
    nv = api.vnrCreateNeuralVolume(syn.model_config(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2), sv)
    return api.neural_get_params_fp16(nv).astype(np.float32)


def test_online_macrocell_is_merged_over_the_ranks(tmp_path):
    world = 2
    res = run_ranks("macrocell", world, tmp_path)
    assert np.array_equal(res[0]["value_range"], res[1]["value_range"])
    assert np.array_equal(res[0]["max_opacity"], res[1]["max_opacity"])
    # the union of both ranks' 3 batches, through the same macrocell kernel in one process
    vol = syn.analytic_volume(32)
    os.environ["VNR_AMD_INIT_SEED"] = "9"
    sv = api.vnrCreateSimpleVolume(vol)
    nv = api.vnrCreateNeuralVolume(syn.model_config(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2),
                                   sv, online_macrocell_construction=True)
    for r in range(world):
        s = api.vnrCreateSimpleVolume(vol)
        carrier = api.vnrCreateNeuralVolume(syn.model_config(n_levels=2, n_features=2, log2_hashmap_size=8, base_resolution=4, n_hidden_layers=1), s)
        check(lib().vnrAmdNeuralVolumeSetSamplerSeed(carrier.h, 1337, 0xda3e39cb94b95bdb + r))
        for _ in range(3):
            c, v = api.simple_volume_take_samples(s, 65536)
            dc, dv = api.DeviceArray.from_numpy(c), api.DeviceArray.from_numpy(v)
            check(lib().vnrAmdNeuralVolumeUpdateMacrocell(nv.h, c.shape[0], dc.ptr, dv.ptr, None))
    check(lib().vnrAmdSynchronize())
    want = api.volume_macrocell(nv)["value_range"]
    assert np.array_equal(res[0]["value_range"], want)
    assert (want[..., 1] > 0).mean() > 0.9   # 6 x 65 536 samples reach nearly every cell of a 32^3 volume's 2^3 macrocells


def test_out_of_core_training_sharded_over_ranks(tmp_path):
    """BASELINE C5 in small: a uint8 volume stays in its file, each rank keeps its OWN random slab set resident, batches are
    sharded over the ranks and the gradients exchanged every step"""
    n = 64
    vol = (syn.analytic_volume(n) * 255.0).round().astype(np.uint8)
    path = tmp_path / "c5_small.raw"
    vol.tofile(path)
    res = run_ranks("ooc", 2, tmp_path, extra_env={"TEST_OOC_FILE": str(path), "TEST_OOC_SIZE": str(n), "TEST_STEPS": "300"})
    assert not np.array_equal(res[0]["slabs"], res[1]["slabs"])          # different slab sets
    assert int(res[0]["checksum"]) == int(res[1]["checksum"])            # one model
    assert np.array_equal(res[0]["value_range"], res[1]["value_range"])  # one macrocell
    assert float(res[0]["psnr"]) > 25.0, float(res[0]["psnr"])
    assert abs(float(res[0]["psnr"]) - float(res[1]["psnr"])) < 1e-3


# ------------------------------------------------------------------------------------------------ RCCL, one rank
def test_rccl_transport_on_a_one_rank_communicator(tmp_path):
    """librccl through dlopen, ncclCommInitRank, and every collective of both sharded paths on the library's own device buffers
    and streams (in-place all-gather of the share, broadcast of parameters and optimizer state, fp16 all-reduce range by range on
    the communication stream, min / max of the macrocell): with one rank they are the identity, so results equal the plain path"""
    res = run_ranks("frames", 1, tmp_path, transport="rccl")[0]
    assert str(res["transport"]) == "rccl" and int(res["rccl_ranks_seen"]) == 1      # ncclCommCount: what the N > 1 bench line reports
    for case in range(5):
        assert bool(res[f"case{case}_sync"]) and bool(res[f"case{case}_pipe"]), case
    steps = 20
    tr = run_ranks("train", 1, tmp_path, transport="rccl", extra_env={"TEST_STEPS": str(steps)})[0]
    want, psnr_ref = _reference_concatenated(1, steps, syn.analytic_volume(32))
    p = tr["params"].view(np.float16).astype(np.float32)
    moved = np.abs(want - _initial_params()).mean()
    assert np.abs(p - want).mean() < 0.02 * moved + 1e-5
    mc = run_ranks("macrocell", 1, tmp_path, transport="rccl")[0]
    assert (mc["value_range"][..., 1] > 0).mean() > 0.5


@pytest.mark.parametrize("world,transport", [(2, "shm"), (3, "shm"), (1, "rccl")])
def test_collective_self_test(world, transport, tmp_path):
    """the first-contact self-test bench.py runs before its timed region with N > 1: in-place all-gather of a frame share,
    reduce-scatter(Avg) of fp16 on a slice length that divides nothing + all-gather of the slices, broadcast, all-reduce(Sum), each
    compared with what the rank computes by itself (world 3: a mean that fp16 cannot hold exactly), each with a completion deadline"""
    for r in run_ranks("selftest", world, tmp_path, transport=transport):
        assert bool(r["ok"]), str(r["report"])
        text = str(r["report"])
        for what in ("all-gather (in place", "reduce-scatter (Avg", "all-reduce (Avg", "broadcast", "all-reduce (Sum"):
            assert what in text and "ok" in text, text


def _bench_two_ranks(transport):
    """bench.py as the driver starts it for N = 2, both ranks on this box's one GPU; transport None = let bench.py choose"""
    import json
    port = _free_port()
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--size", "64", "--fb", "256", "--levels", "6",
            "--log2-hashmap-size", "14", "--hidden-layers", "2", "--train-steps", "120", "--no-psnr", "--no-cpu-baseline"]
    procs = []
    for rank in range(2):
        env = dict(os.environ)
        env.pop("VNR_AMD_DIST_TRANSPORT", None)
        if transport:
            env["VNR_AMD_DIST_TRANSPORT"] = transport
        else:
            env["VNR_BENCH_PROBE_LIMIT"] = "90"
        env.update({"RANK": str(rank), "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "VNR_AMD_DIST_TIMEOUT": "120", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.pop("VNR_RM_N_ITERS", None)
        procs.append(subprocess.Popen(args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=400) for p in procs]
    for rank, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank}:\n{o[-1500:]}\n{e[-2500:]}"
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not any(l.startswith("{") for l in outs[1][0].splitlines())
    return json.loads(lines[0])


def test_bench_line_of_two_ranks_carries_per_rank_times(tmp_path):
    """bench.py as the driver starts it for N = 2 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; here both ranks on this box's one GPU over the
    host-staged transport, a small workload): the self-test runs, ONE JSON line comes from rank 0, n_gpus = 2, and it carries the per-rank
    share / gather / training-exchange times"""
    d = _bench_two_ranks("shm")
    # (round 5) two keys beside n_gpus answer "did RCCL see N ranks": here it did not, and the line does not call that curve "strong"
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"].startswith("strong-over-shm") and "NOT an RCCL" in d["scaling"]
    assert d["transport"].startswith("shm (chosen") and d["rccl_ranks_seen"] == 0
    pr = d["per_rank"]
    for k in ("share_ms", "gather_ms", "train_step_ms", "train_exchange_ms"):
        assert len(pr[k]) == 2 and all(v > 0 for v in pr[k]), (k, pr[k])
    assert "all-gather (in place" in d["collective_self_test"] and "reduce-scatter" in d["collective_self_test"]
    assert "transport_probe" not in d          # the environment chose


def test_bench_probes_rccl_in_a_child_and_goes_on_over_shared_memory_when_it_fails():
    """with no transport chosen bench.py first meets the other ranks over RCCL in child processes.  Two ranks on ONE device is something
    RCCL refuses, which makes this box a rehearsal of a new installation where RCCL does not come up: the children fail (or are killed
    at the limit), every rank goes on over the host-staged transport, and the ONE line says so instead of silently reporting shm numbers
    as RCCL's"""
    d = _bench_two_ranks(None)
    assert d["n_gpus"] == 2 and d["value"] > 0
    tp = d["transport_probe"]
    assert tp["rccl"].startswith("FAILED") and "shm" in tp["fallback"], tp
    assert "shm" in d["config"]["parallelism"], d["config"]["parallelism"]
    assert d["transport"].startswith("shm-fallback") and d["rccl_ranks_seen"] == 0 and d["scaling"] != "strong"
    assert "all-gather (in place" in d["collective_self_test"]          # the run's own self-test, over the transport it uses


def test_bare_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher environment must measure two ranks, not print n_gpus 1 for a
    one-GPU run.  bench.py becomes the launcher (a parent that never touches the GPU starts two fresh copies of itself with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*); on this one-GPU box it sees fewer devices than ranks, tells the ranks to share the device over the host-staged
    transport and says so: ONE line on stdout, n_gpus 2, not called an RCCL curve."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VNR_AMD_DIST_TRANSPORT", "VNR_RM_N_ITERS")}
    env["VNR_AMD_DIST_TIMEOUT"] = "120"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--size", "64", "--fb", "256", "--levels", "6",
                          "--log2-hashmap-size", "14", "--hidden-layers", "2", "--train-steps", "120", "--no-psnr", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["launcher"].startswith("bench.py")
    # either the launcher saw that the box has fewer devices than ranks (and said so), or the ranks' RCCL probe failed on the shared device
    assert d["transport"].startswith("shm") and "NOT an RCCL" in d["scaling"] and d["rccl_ranks_seen"] == 0
    assert "2 ranks on 1 visible GPU" in d["transport"] and "NOT an N-GPU measurement" in out.stderr or d["transport"].startswith("shm-fallback"), d["transport"]
    print("\nbare --gpus 2:", d["transport"], "| launcher:", d["launcher"][:40])
    assert len(d["per_rank"]["share_ms"]) == 2
    # and a launcher environment that contradicts --gpus is refused, on a GPU box too
    env.update({"WORLD_SIZE": "1", "RANK": "0"})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and not out.stdout.strip()


def test_a_rank_s_share_is_as_fast_with_a_fifth_stream_alive_in_the_process():
    """the stream budget (profiles/r05_stream_budget.txt): the HIP runtime gives the first four streams of a process a hardware
    queue each and every later one the least used queue; a small share's ray parts run one behind the other when two of them share a queue.
    With FOUR parts (rounds 2 - 4) one more stream in the process -- which every multi-GPU rank has: its communication stream -- made the 1/8
    share of the bench frame 45 % slower (0.55 -> 0.85 ms), i.e. the first 8-GPU run would have scaled 4.3 x where the one-GPU probe promised
    6.5 x.  Three parts on streams created together with the library's own (Runtime::init) must not care how many streams come later -- one (a
    rank's communication stream) or two (an out-of-core sampler's beside it): tools/share_probe.py as a rank runs it"""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ms = {}
    for extra in (0, 1, 2, 0):
        env = dict(os.environ)
        env.update({"SHARE_PARTS": "8", "SHARE_FRAMES": "40", "SHARE_PIPELINED": "1", "SHARE_EXTRA_STREAMS": str(extra)})
        for k in ("VNR_AMD_SMALL_SHARE_PARTS", "VNR_RM_N_ITERS", "VNR_AMD_RENDER_HALVES", "GPU_MAX_HW_QUEUES"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "share_probe.py")], env=env, capture_output=True, text=True, timeout=300, cwd=root)
        m = re.search(r"share 1/8: ([0-9.]+) ms per frame", out.stdout)
        assert out.returncode == 0 and m, out.stdout[-1500:] + out.stderr[-1500:]
        ms.setdefault(extra, []).append(float(m.group(1)))
    alone, beside = min(ms[0]), max(ms[1] + ms[2])
    print(f"\n1/8 share of the bench frame: {alone:.3f} ms alone, {ms[1][0]:.3f} / {ms[2][0]:.3f} ms with one / two more streams alive in the process")
    # wall-clock bars belong to a perf run, not to the correctness suite (a busy or different GPU fails them without a defect: ADVICE r05):
    # the figures are always printed, the bars hold with VNR_TEST_TIMING=1 (the builder's own GPU runs set it)
    if os.environ.get("VNR_TEST_TIMING") == "1":
        assert beside < 1.15 * alone, (ms, "two ray parts of the share ended up on one hardware queue")
        assert alone < 0.75, ms          # (0.54 - 0.57 measured; 0.85 was the broken state)


# ------------------------------------------------------------------------------------------------ asynchronous frames
def _download(ptr, n_pixels):
    out = np.empty((n_pixels, 4), np.float32)
    check(lib().vnrAmdMemcpyD2H(out.ctypes.data_as(C.c_void_p), ptr, out.nbytes))
    return out


def test_asynchronous_frames_equal_synchronous_ones(monkeypatch):
    """vnrAmdRendererSetAsync: vnrRender returns after enqueueing the iterations the previous frame needed; MapFrame completes the
    frame.  Frames must equal the synchronous renderer's, also when a frame needs MORE iterations than the previous one (the
    camera moves from far to near and the sampling rate rises: the prediction is too short and MapFrame has to launch the rest) and when
    it needs fewer."""
    size = (96, 64)
    n_pixels = size[0] * size[1]
    vol = syn.analytic_volume(48)
    sv = api.vnrCreateSimpleVolume(vol)
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors)
    api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    monkeypatch.setenv("VNR_RM_N_ITERS", "4")   # read when a renderer is created: many iterations per frame on a small volume

    def renderer(asynchronous):
        r = api.vnrCreateRenderer(sv)
        api.vnrRendererSetTransferFunction(r, tfn)
        api.vnrRendererSetFramebufferSize(r, size)
        api.vnrRendererSetMode(r, 5)
        api.vnrRendererSetOutputAsDeviceFramebuffer(r, True)
        check(lib().vnrAmdRendererSetAsync(r.h, 1 if asynchronous else 0))
        return r
    r_sync, r_async = renderer(False), renderer(True)
    iterations = []
    for distance, rate in ((3.0, 1.0), (3.0, 1.0), (0.9, 4.0), (0.9, 4.0), (0.9, 1.0), (3.0, 0.5), (1.5, 2.0)):
        cam = syn.oblique_camera((48, 48, 48), distance_scale=distance)
        frames = []
        for r in (r_sync, r_async):
            camera = api.vnrCreateCamera()
            api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
            api.vnrRendererSetCamera(r, camera)
            api.vnrRendererSetVolumeSamplingRate(r, rate)
            api.vnrRender(r)
            frames.append(_download(api.vnrRendererMapFrame(r), n_pixels))
        assert np.array_equal(frames[0], frames[1])
        a, b = api.vnrRendererGetFrameStats(r_sync), api.vnrRendererGetFrameStats(r_async)
        assert a["n_samples"] == b["n_samples"] and a["n_iterations"] == b["n_iterations"] and a["n_rays_hit"] == b["n_rays_hit"]
        iterations.append(a["n_iterations"])
    assert max(iterations) > min(iterations) + 1   # the scene really changed its iteration count
    # statistics before MapFrame complete the frame too
    api.vnrRender(r_async)
    st = api.vnrRendererGetFrameStats(r_async)
    assert st["n_iterations"] == iterations[-1]


def test_pipelined_frames_of_one_renderer_equal_synchronous_ones(monkeypatch):
    """vnrAmdRendererRenderPipelined without a process group: the head of frame k + 1 (ray generation, first batch of samples) is
    enqueued before the host has seen frame k complete.  Frames accumulate over a still camera (the case that pipelines), then
    the camera moves (a frame that restarts the accumulation is not pipelined), then accumulate again; every frame handed out
    must equal the synchronous renderer's, statistics included."""
    size = (96, 80)
    n_pixels = size[0] * size[1]
    vol = syn.analytic_volume(48)
    sv = api.vnrCreateSimpleVolume(vol)
    nv = api.vnrCreateNeuralVolume(syn.model_config(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2), sv,
                                   online_macrocell_construction=False)
    api.vnrNeuralVolumeTrain(nv, 30, True)
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors)
    api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    for volume in (sv, nv):
        def renderer():
            r = api.vnrCreateRenderer(volume)
            api.vnrRendererSetTransferFunction(r, tfn)
            api.vnrRendererSetFramebufferSize(r, size)
            api.vnrRendererSetMode(r, 5)
            api.vnrRendererSetOutputAsDeviceFramebuffer(r, True)
            return r
        r_sync, r_pipe = renderer(), renderer()
        want, got, want_samples, got_samples = [], [], [], []
        out = C.c_void_p()
        st = vdist.ShardedRenderer(vdist.Context(), r_pipe, size[0], size[1])
        for distance, frames in ((1.2, 4), (2.5, 1), (0.9, 3)):
            cam = syn.oblique_camera((48, 48, 48), distance_scale=distance)
            for r in (r_sync, r_pipe):
                camera = api.vnrCreateCamera()
                api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
                api.vnrRendererSetCamera(r, camera)    # (completes a frame of the pipelined renderer that is still in flight)
            for _ in range(frames):
                api.vnrRender(r_sync)
                want.append(_download(api.vnrRendererMapFrame(r_sync), n_pixels))
                want_samples.append(api.vnrRendererGetFrameStats(r_sync)["n_samples"])
                check(lib().vnrAmdRendererRenderPipelined(r_pipe.h, C.byref(out)))
                if out.value:
                    got.append(_download(out, n_pixels))
                    got_samples.append(st.completed_stats()["n_samples"])
        check(lib().vnrAmdRendererFlushPipeline(r_pipe.h, C.byref(out)))
        got.append(_download(out, n_pixels))
        got_samples.append(st.completed_stats()["n_samples"])
        assert len(got) == len(want) == 8
        for k, (g, w) in enumerate(zip(got, want)):
            assert np.array_equal(g, w), k
        assert got_samples == want_samples


def test_pipelined_renderer_survives_what_an_application_does_between_frames():
    """tools/pipeline_soak.py: 1500 vnrAmdRendererRenderPipelined calls with camera moves, mode switches (5 / 6 / 8), resizes, transfer
    function updates, accumulation resets and training steps in between; after every event the pipelined frame equals the frame of a
    fresh sequential renderer in the same state, bit for bit"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "pipeline_soak.py"), "1500"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "[soak] ok: 1500 pipelined frames" in out.stdout


def _run_tool_ranks(argv, world, cwd, timeout=420):
    """one process per rank of a command-line tool, torchrun-style environment, host-staged transport (all ranks on this box's GPU)"""
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "VNR_AMD_DIST_TRANSPORT": "shm", "VNR_AMD_DIST_TIMEOUT": "120", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "VNR_AMD_INIT_SEED": str(500 + rank)})
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, cwd=str(cwd), stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            outs.append((o.decode("utf-8", "replace"), e.decode("utf-8", "replace")))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} of {world} failed:\n{o[-1500:]}\n{e[-2500:]}"
    return outs


def test_command_line_tools_on_two_ranks(tmp_path):
    """tools/vnr_cmd_train.py and tools/vnr_cmd_render.py with one process per GPU (here: two ranks on one GPU): the trainer's steps are
    data-parallel steps and rank 0 writes params.json; the renderer's screenshot is the one-process screenshot, bit for bit"""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    vol = syn.analytic_volume(48)
    vol.astype(np.float32).tofile(tmp_path / "v.raw")
    scene = {"dataSource": [{"format": "REGULAR_GRID_RAW_BINARY", "fileName": str(tmp_path / "v.raw"), "dimensions": {"x": 48, "y": 48, "z": 48}, "type": "FLOAT"}],
             "view": {"camera": {"eye": {"x": 100.0, "y": 90.0, "z": -60.0}, "center": {"x": 24.0, "y": 24.0, "z": 24.0}, "up": {"x": 0.0, "y": 1.0, "z": 0.0}, "fovy": 40.0},
                      "volume": {"transferFunction": {}}}}
    (tmp_path / "scene.json").write_text(json.dumps(scene))
    (tmp_path / "model.json").write_text(json.dumps(syn.model_config(n_levels=6, n_features=4, log2_hashmap_size=14, base_resolution=4, n_hidden_layers=2)))
    outs = _run_tool_ranks([os.path.join(root, "tools", "vnr_cmd_train.py"), "--volume", str(tmp_path / "scene.json"), "--network", str(tmp_path / "model.json"),
                            "--max-num-steps", "200", "--quiet"], 2, tmp_path)
    summary = dict(line.strip().split("=") for line in outs[0][0].splitlines() if "=" in line)
    assert summary["STEP"] == "200" and float(summary["PSNR"]) > 28.0
    assert "Summary" not in outs[1][0]                                  # rank 0 reports
    nv = api.vnrCreateNeuralVolume(str(tmp_path / "params.json"))
    assert api.vnrVolumeGetDims(nv) == (48, 48, 48)
    x = np.linspace(0, 1, 64, dtype=np.float32)
    np.save(tmp_path / "table.npy", np.stack([x, 1 - x, 0.5 + 0.5 * np.sin(6 * x), np.clip(1.5 * x - 0.2, 0, 1)], axis=1).astype(np.float32))
    args = [os.path.join(root, "tools", "vnr_cmd_render.py"), "--tfn", str(tmp_path / "scene.json"), "--tfn-table", str(tmp_path / "table.npy"), "--num-frames", "4",
            "--neural-volume", str(tmp_path / "params.json"), "--rendering-mode", "5"]
    outs = _run_tool_ranks(args + ["--exp", "two"], 2, tmp_path)
    assert "Summary: two" in outs[0][0] and "gpus: 2" in outs[0][0] and "Summary" not in outs[1][0]
    one = subprocess.run([sys.executable] + args + ["--exp", "one"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    assert (tmp_path / "two-screenshot.png").read_bytes() == (tmp_path / "one-screenshot.png").read_bytes()
