#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native instantvnr hot path.

Metric (BASELINE.json): fps at 1024^2 on a 1024^3 volume + MLP Msamples/s; PSNR vs ground truth.
One "step" = one frame: sample-streaming ray march (rendering mode 5) of the trained neural volume
(HashGrid L=16 F=2 T=2^22 + 3x64 FullyFusedMLP) at 1024x1024; with N GPUs the image is sharded by interleaved
8-scanline tile rows and every frame is gathered with one RCCL all-gather (strong scaling: total work is fixed).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

torch.distributed.run is only the launcher: this program reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and does everything,
collectives included, through libvnr_amd.so (vnrAmdDist*, RCCL opened with dlopen); it never imports torch.

Untimed setup: generate the synthetic volume on the GPU, train the model (data parallel with an RCCL gradient exchange when
N > 1), build the renderer.  Rank 0 prints ONE JSON line.
Other BASELINE configurations through the same program: C2 + C3 = `--size 128 --fb 512 --levels 8 --features 8
--log2-hashmap-size 19 --hidden-layers 2 --per-level-scale 2 --train-steps 10000` (synthetic vortex field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from instantvnr_amd import api, dist, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--size", type=int, default=1024, help="volume edge (C4: 1024)")
    p.add_argument("--fb", type=int, default=1024, help="framebuffer edge (C4: 1024)")
    p.add_argument("--volume", choices=("auto", "perlin", "vortex"), default="auto",
                   help="synthetic field: perlin fBm generated on the GPU (C4), vortex tubes (C2 / C3 stand-in for vorts1); auto = vortex up to 256^3")
    p.add_argument("--levels", type=int, default=16)
    p.add_argument("--features", type=int, default=2)
    p.add_argument("--log2-hashmap-size", type=int, default=22)
    p.add_argument("--hidden-layers", type=int, default=3)
    p.add_argument("--per-level-scale", type=float, default=0.0, help="0 = finest level resolution equals the volume edge")
    p.add_argument("--train-steps", type=int, default=1500)
    p.add_argument("--opacity-scale", type=float, default=0.06)
    p.add_argument("--camera-distance", type=float, default=1.1, help="camera distance in volume edges (oblique view)")
    p.add_argument("--probe-collectives", action="store_true",
                   help="internal (N > 1): meet the other ranks over RCCL, run the collective self-test, exit 0 / 3; bench.py starts itself in this "
                        "mode in a child process first, so that a transport that fails or hangs on a new installation costs a child, not the run")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-psnr", action="store_true")
    p.add_argument("--no-alone", action="store_true", help="skip the un-timed one-stream leg (roofline.alone)")
    p.add_argument("--no-brick-off", action="store_true", help="skip the un-timed leg without the brick image (train-while-render configuration)")
    p.add_argument("--no-brick-table", action="store_true", help="skip the un-timed legs at smaller budgets of the brick image (inference_cache.budget_table)")
    p.add_argument("--no-interactive", action="store_true", help="skip the un-timed train-while-render leg (`interactive`)")
    p.add_argument("--no-kernel-events", action="store_true", help="diagnostics: no HIP events around the evaluation kernel (roofline.achieved reads 0)")
    p.add_argument("--c5", action="store_true",
                   help="extra un-timed leg: BASELINE C5's training step from an out-of-core file (2 GiB uint8 written to --c5-dir, 16 384 resident "
                        "slabs, 1 024 replaced per refresh), synchronous and asynchronous refresh -> the `c5` object of the line")
    p.add_argument("--c5-dir", default="/tmp")
    p.add_argument("--mode", type=int, default=5, choices=(5, 6, 8, 9, 11, 12, 14, 15),
                   help="rendering mode: 5 = sample streaming (BASELINE metric, default), 8 = the same with gradient shading (4 evaluations per sample)")
    return p.parse_args()


def cpu_baseline(sv, nv, info, dims, tfn_np, cam, fb, mc, pls, hidden_layers, log2_T):
    """the oracle's monolithic ground-truth ray marcher (mode-4 semantics: manual trilinear, macrocell DDA, adaptive
    step, TFN, compositing) on the host cores: same camera / TFN / volume as the timed GPU workload.  A 1/16 probe of
    the frame sizes the run: the whole frame is timed when that fits ~30 s of wall time, else the probe is reported.
    Also reports the oracle's (scalar, one core) network inference rate on the trained parameters."""
    from oracle import oracle
    import ctypes as C
    n = dims[0] * dims[1] * dims[2]
    host = np.empty(n, dtype=np.float32)
    t0 = time.perf_counter()
    check(lib().vnrAmdMemcpyD2H(host.ctypes.data_as(C.c_void_p), lib().vnrAmdSimpleVolumeDeviceData(sv.h), n * 4))
    vol = host.reshape(dims[2], dims[1], dims[0])
    colors, alphas = tfn_np
    sc = oracle.SceneHolder(fb, fb, dims, oracle.TfnHolder(colors, alphas), mc["max_opacity"], cam["from"], cam["at"], cam["up"], cam["fovy"])
    cores = os.cpu_count() or 1
    n_blocks = 8  # probe: 8 evenly spaced blocks of 8 scanlines (1/16 of the frame at fb = 1024)
    rows = [(int((b + 0.5) * fb / n_blocks) - 4, int((b + 0.5) * fb / n_blocks) + 4) for b in range(n_blocks)]
    t1 = time.perf_counter()
    oracle.render_monolithic(sc, vol, n_threads=cores, rows=rows)
    dt = time.perf_counter() - t1
    frac = sum(b - a for a, b in rows) / float(fb)
    sample = f"{n_blocks} blocks x 8 scanlines ({frac:.4f} of the {fb}x{fb} frame), {dt:.2f} s"
    value = frac / dt
    est_full = dt / frac
    if est_full <= 30.0:
        reps = int(max(1, min(16, 10.0 // max(est_full, 1e-3))))
        t2 = time.perf_counter()
        for _ in range(reps):
            oracle.render_monolithic(sc, vol, n_threads=cores)
        dt = time.perf_counter() - t2
        value = reps / dt
        sample = f"{reps} whole {fb}x{fb} frame(s), {dt:.2f} s"
    out = {"value": round(value, 4), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": sample + f"; same camera/TFN on the ground-truth {dims[0]}^3 volume, CPU oracle monolithic marcher "
                              f"(oracle/vnr_oracle.c), {cores} threads over scanlines",
           "d2h_volume_s": round(t1 - t0, 2)}
    # network inference of the oracle (fp32-accumulate MLP, scalar C, 1 core) on the trained parameters
    try:
        ocfg = oracle.grid_config(info["n_levels"], info["n_features_per_level"], log2_T, 16, per_level_scale=pls)
        params = api.neural_get_params_fp16(nv).view(np.uint16)
        coords = np.random.default_rng(5).random((200000, 3), dtype=np.float32)
        t3 = time.perf_counter()
        oracle.network_inference(ocfg, 64, hidden_layers, params, coords)
        out["network_msamples_per_s_1core"] = round(coords.shape[0] / (time.perf_counter() - t3) / 1e6, 4)
    except Exception as e:  # the baseline is informational; never lose the bench line over it
        out["network_msamples_per_s_1core"] = None
        out["network_note"] = str(e)[:200]
    return out


EVAL_KERNEL_SOURCES = ("infer_kernel.h", "infer_tile.h", "grid_device.h", "network_infer.hip", "network_infer_w64.hip")


def sources_sha16(files):
    """names what a kernel is compiled from: counter results of separate rocprofv3 --pmc passes (profiles/r06_*.json carry `source_files` and
    `source_sha16`) enter the line only while this still gives the hash they were stamped with"""
    import hashlib
    import re
    h = hashlib.sha256()
    for n in files:
        text = open(os.path.join(ROOT, "instantvnr_amd", "csrc", n), encoding="utf-8", errors="replace").read()
        # the code, not its commentary: `//` comments, trailing blanks and empty lines do not change what is compiled
        lines = (re.sub(r"\s*//.*$", "", line).rstrip() for line in text.splitlines())
        h.update("\n".join(line for line in lines if line).encode())
    return h.hexdigest()[:16]


def workload_name(a):
    if (a.size, a.fb, a.levels, a.features, a.hidden_layers) == (1024, 1024, 16, 2, 3):
        return "C4"
    if (a.size, a.fb, a.levels, a.features, a.hidden_layers) == (128, 512, 8, 8, 2):
        return "C2 (model trained as C3)"
    return "custom"


def untimed_frames(ren, n, warm=3):
    """n frames of an already configured renderer, timed on the host; -> (fps, samples, kernel ms, launches, union ms)"""
    for _ in range(warm):
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
    check(lib().vnrAmdSynchronize())
    t = time.perf_counter()
    samples = launches = 0
    ms = union = 0.0
    for _ in range(n):
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
        s = api.vnrRendererGetFrameStats(ren)
        samples += s["n_samples"]; ms += s["infer_kernel_ms"]; launches += s["infer_kernel_launches"]; union += s["infer_union_ms"]
    check(lib().vnrAmdSynchronize())
    return n / (time.perf_counter() - t), samples, ms, launches, union


def interactive_leg(a, nv, make_renderer, brick_off):
    """render a frame, train k steps with fast_mode = false, render the next frame ...: frames/s of the loop for k = 1 and k = 10, beside the sum
    of its parts measured alone (the frame without the inference cache, which every optimizer step drops; the training call alone)"""
    L = lib()
    ren = make_renderer(nv, profiling=False)
    out = {"what": "the reference application's loop (apps/int_dual_volume.cpp:631-672): vnrRender + vnrRendererMapFrame, then vnrNeuralVolumeTrain(nv, k, "
                   "fast_mode = false) (optimizer steps + the macrocell's update from every step's samples + its max-opacity refresh per call: "
                   "core/network.cu:231-259, 770-779), every frame; un-timed leg of this run, not `value`", "legs": []}

    def train_alone(k, calls):
        check(L.vnrAmdSynchronize())
        t = time.perf_counter()
        for _ in range(calls):
            api.vnrNeuralVolumeTrain(nv, k, False)
        check(L.vnrAmdSynchronize())
        return (time.perf_counter() - t) * 1e3 / calls

    api.vnrNeuralVolumeTrain(nv, 5, False)     # first launches of the slow-mode kernels
    image_builds0 = api.neural_brick_image(nv).get("builds", 0)
    def loop(k, frames):
        for _ in range(3):
            api.vnrRender(ren); api.vnrRendererMapFrame(ren); api.vnrNeuralVolumeTrain(nv, k, False)
        check(L.vnrAmdSynchronize())
        t = time.perf_counter()
        tiers = [0, 0, 0]
        for _ in range(frames):
            api.vnrRender(ren); api.vnrRendererMapFrame(ren)
            tiers[api.neural_brick_image(nv)["tier"]] += 1
            api.vnrNeuralVolumeTrain(nv, k, False)
        check(L.vnrAmdSynchronize())
        return (time.perf_counter() - t) * 1e3 / frames, tiers

    small0 = api.neural_brick_image(nv)["small_builds"]
    for k, frames in ((1, 40), (10, 20)):
        train_ms = train_alone(k, max(4, 40 // k))
        ms, tiers = loop(k, frames)                                 # the library's own policy: the small tier of the cache, rebuilt per frame
        check(L.vnrAmdNeuralVolumeSetBrickImageMode(nv.h, 0))
        ms_off, _ = loop(k, frames)                                 # no cache at all (rounds 1-5)
        check(L.vnrAmdNeuralVolumeSetBrickImageMode(nv.h, -1))
        st = api.neural_brick_image(nv)
        leg = {"train_steps_per_frame": k, "frames": frames, "fps": round(1e3 / ms, 2), "ms_per_frame_and_training": round(ms, 4),
               "fps_without_any_cache": round(1e3 / ms_off, 2), "ms_without_any_cache": round(ms_off, 4),
               "training_call_alone_ms": round(train_ms, 4), "ms_per_step_fast_mode_false": round(train_ms / k, 4),
               "frames_by_cache_tier": {"none": tiers[0], "small": tiers[1], "full": tiers[2]}}
        if brick_off:
            frame_ms = 1e3 / brick_off["fps"]
            leg["frame_alone_without_cache_ms"] = round(frame_ms, 4)
            leg["sum_of_parts_without_cache_ms"] = round(frame_ms + train_ms, 4)
        out["legs"].append(leg)
    st = api.neural_brick_image(nv)
    out["inference_cache"] = {"policy": "while the parameters change after every frame the first large launch of a frame builds a SMALL image (finest levels within "
                                        "VNR_AMD_BRICK_SMALL_GB, default 0.75 GiB); the full image waits until they have been left alone (and backs off: network.h)",
                              "small_builds_during_the_legs": st["small_builds"] - small0, "full_builds_during_the_legs": st["builds"] - image_builds0}
    # would a SMALL cache rebuilt after every training call pay?  The finest-first levels that fit the budget, built at the first launch after the
    # parameters changed (mode 1), i.e. once per frame of this loop.
    small = []
    for gib in (0.25, 0.75, 1.7):
        api.neural_set_brick_budget(nv, int(gib * 2**30))
        check(L.vnrAmdNeuralVolumeSetBrickImageMode(nv.h, 1))
        for k, frames in ((1, 30), (10, 12)):
            for _ in range(3):
                api.vnrRender(ren); api.vnrRendererMapFrame(ren); api.vnrNeuralVolumeTrain(nv, k, False)
            check(L.vnrAmdSynchronize())
            b0 = api.neural_brick_image(nv)["builds"]
            t = time.perf_counter()
            for _ in range(frames):
                api.vnrRender(ren); api.vnrRendererMapFrame(ren); api.vnrNeuralVolumeTrain(nv, k, False)
            check(L.vnrAmdSynchronize())
            ms = (time.perf_counter() - t) * 1e3 / frames
            st = api.neural_brick_image(nv)
            small.append({"budget_gib": gib, "train_steps_per_frame": k, "fps": round(1e3 / ms, 2), "ms_per_frame_and_training": round(ms, 4),
                          "image_bytes": int(st["bytes"]), "build_ms": round(st["build_ms"], 4), "builds_per_frame": round((st["builds"] - b0) / frames, 2)})
    api.neural_set_brick_budget(nv, 0)
    check(L.vnrAmdNeuralVolumeSetBrickImageMode(nv.h, -1))
    out["budget_sweep_of_a_cache_rebuilt_every_frame"] = small
    out["training_step"] = api.vnrNeuralVolumeGetTrainingStep(nv)
    out["loss"] = round(api.vnrNeuralVolumeGetTrainingLoss(nv), 5)
    del ren
    return out


def c5_leg(a, ctx):
    """BASELINE C5 in the form one box can hold: a 2 GiB uint8 file (1024 x 1024 x 2048, written here once), the C4-shaped model trained
    from it with 16 384 resident slabs (1.6 GiB of HBM) of which 1 024 are replaced per refresh (neural_sampler.cpp:1043-1127), every rank
    with its own slab set and the gradients exchanged every step.  Two legs: the default (synchronous: each step waits for its refresh, the
    reference's semantics) and asynchronous refresh (a step never waits for the storage).  -> dict for the bench line"""
    import ctypes as C
    L = lib()
    nx, ny, nz = 1024, 1024, 2048
    path = os.path.join(a.c5_dir, f"vnr_c5_{nx}x{ny}x{nz}.raw")
    t0 = time.perf_counter()
    if ctx.rank == 0 and (not os.path.exists(path) or os.path.getsize(path) != nx * ny * nz):
        x = np.linspace(0, 1, nx, dtype=np.float32)[None, None, :]
        y = np.linspace(0, 1, ny, dtype=np.float32)[None, :, None]
        with open(path + ".tmp", "wb") as f:
            for z0 in range(0, nz, 16):
                z = (np.arange(z0, min(z0 + 16, nz), dtype=np.float32) / nz)[:, None, None]
                v = 0.5 + 0.5 * np.sin(40 * x + 9 * z) * np.cos(31 * y) * np.sin(23 * z + 5 * x * y)
                f.write((v * 255.0 + 0.5).astype(np.uint8).tobytes())
        os.replace(path + ".tmp", path)
    write_s = time.perf_counter() - t0
    dist.barrier(ctx)
    sv = api.vnrCreateSimpleVolumeOutOfCore(path, (nx, ny, nz), np.uint8, (0.0, 255.0), n_concurrent_blocks=1024, n_blocks=16384)
    info = api.out_of_core_info(sv)
    pls = float(np.exp(np.log(max(nx, ny, nz) / 16.0) / 15))
    cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, base_resolution=16, n_hidden_layers=3, per_level_scale=pls)
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=True)
    dist.train_data_parallel(ctx, nv, 20, fast_mode=False)
    first = api.vnrNeuralVolumeGetTrainingLoss(nv)

    def leg(steps):
        dist.barrier(ctx)
        b0 = api.out_of_core_info(sv)["bytes_read"]
        r0, y0 = C.c_uint64(), C.c_uint64()
        check(L.vnrAmdSimpleVolumeOutOfCoreRefreshStats(sv.h, C.byref(r0), C.byref(y0)))
        t = time.perf_counter()
        dist.train_data_parallel(ctx, nv, steps, fast_mode=False)
        dist.barrier(ctx)
        dt = time.perf_counter() - t
        if ctx.distributed:
            dt = dist.all_reduce_host([dt], dist.MAX)[0]
        r1, y1 = C.c_uint64(), C.c_uint64()
        check(L.vnrAmdSimpleVolumeOutOfCoreRefreshStats(sv.h, C.byref(r1), C.byref(y1)))
        return {"ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "msamples_per_s": round(65536 * ctx.world * steps / dt / 1e6, 1),
                "turnover_gib_per_s_rank0": round((api.out_of_core_info(sv)["bytes_read"] - b0) / dt / 2**30, 2),
                "refreshes_rank0": int(r1.value - r0.value), "steps_beside_a_refresh_rank0": int(y1.value - y0.value),
                "loss": round(api.vnrNeuralVolumeGetTrainingLoss(nv), 5)}

    sync = leg(200)
    check(L.vnrAmdSimpleVolumeOutOfCoreSetAsyncRefresh(sv.h, 1))
    dist.train_data_parallel(ctx, nv, 10, fast_mode=False)
    asyn = leg(400)
    del nv, sv
    dist.barrier(ctx)
    if ctx.rank == 0 and not os.environ.get("VNR_BENCH_KEEP_C5_FILE"):   # 2 GiB in --c5-dir: not left behind
        try:
            os.remove(path)
        except OSError:
            pass
    return {"workload": f"C5 stand-in: {nx}x{ny}x{nz} uint8 file ({nx * ny * nz / 2**30:.0f} GiB, written in {write_s:.1f} s on rank 0), slab "
                        f"{tuple(info['block_dims'])} voxels = {info['block_size_aligned']} B, {info['n_blocks']} resident slabs per rank "
                        f"({info['n_blocks'] * info['block_size_aligned'] / 2**30:.2f} GiB of HBM), {info['n_concurrent_blocks']} replaced per refresh; C4-shaped model, "
                        f"65 536 samples per rank and step, online macrocell; the 4096^3 / 64 GiB file of BASELINE C5 differs in the file only (same slabs, same step)",
            "loss_after_20_steps": round(first, 5), "synchronous_refresh (default, the reference's semantics)": sync,
            "asynchronous_refresh (vnrAmdSimpleVolumeOutOfCoreSetAsyncRefresh)": asyn, "n_gpus": ctx.world}


def probe_collectives_child():
    """the child of choose_transport(): RCCL first contact + every collective of the sharded paths on patterned buffers, with deadlines.
    The ranks' children agree over their control plane, so all of them exit with the same code."""
    try:
        ctx = dist.init_from_env(transport="rccl")
        ok, text = dist.self_test(deadline_s=float(os.environ.get("VNR_BENCH_SELFTEST_DEADLINE", "60")))
        all_ok = dist.all_reduce_host([1.0 if ok else 0.0], dist.MIN)[0] >= 1.0
        if ctx.rank == 0 or not ok:
            print(("ok: " if ok else "FAILED: ") + " ".join(text.split()), flush=True)
    except Exception as e:  # an init that fails outright (no librccl, a device used twice ...)
        print("FAILED: " + " ".join(str(e).split()), flush=True)
        all_ok = False
    sys.stdout.flush()
    os._exit(0 if all_ok else 3)   # no finaliser against a communicator that may hold a collective that never completes


def choose_transport():
    """N > 1 on an installation nobody has run before: which transport carries the collectives?  RCCL is what the design is for; the
    host-staged shared-memory transport (csrc/dist.cpp, one node only) is slower and always works.  Unless the environment has chosen
    (VNR_AMD_DIST_TRANSPORT), every rank first starts itself as a child in --probe-collectives mode (before this process touches the
    GPU: the child is an ordinary fork + exec) on a rendezvous port of its own.  Children that report a failed collective, die, or
    are still running after the limit are killed and the run goes on over shared memory, SAYING SO in its JSON line
    (`transport_probe`); a run whose own self-test then fails still exits without a line."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or os.environ.get("VNR_AMD_DIST_TRANSPORT"):
        return None, None
    import subprocess
    # (the verdict files of the parents' agreement below: a file this rank left behind in a run that was killed must not speak for it now)
    key = f"vnr_bench_probe_{os.environ.get('MASTER_PORT', '29500')}_{os.environ.get('VNR_BENCH_RUN_ID') or os.getppid()}"
    rank = int(os.environ.get("RANK", "0"))
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", key)
    if os.path.exists(f"{base}_{rank}"):
        os.remove(f"{base}_{rank}")
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 17)
    # the probe is bounded tightly (a driver's own clock runs around this program): 60 s for the children to meet, 30 s per collective -> a
    # transport that hangs costs at most 240 s before the run goes on over shared memory (a working RCCL start-up of 8 ranks takes seconds)
    env["VNR_AMD_DIST_TIMEOUT"] = env.get("VNR_BENCH_PROBE_RENDEZVOUS", "60")
    env["VNR_BENCH_SELFTEST_DEADLINE"] = env.get("VNR_BENCH_PROBE_DEADLINE", "30")
    # the child's worst case is its rendezvous timeout + five collectives at their deadline each: the limit must not cut a child that
    # would still have answered
    deadline = float(env["VNR_BENCH_SELFTEST_DEADLINE"])
    limit = max(float(os.environ.get("VNR_BENCH_PROBE_LIMIT", "0")), float(env["VNR_AMD_DIST_TIMEOUT"]) + 5 * deadline + 30)
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe-collectives"], env=env, capture_output=True, text=True, timeout=limit)
        said = (r.stdout.strip().splitlines() or [""])[-1]
        ok = r.returncode == 0
        if not ok and not said:
            said = f"FAILED: child exited with code {r.returncode}: " + " ".join(r.stderr.strip().split())[-300:]
    except subprocess.TimeoutExpired:
        ok, said = False, f"FAILED: no answer within {limit:.0f} s (child killed)"
    probe = {"rccl": said[:600], "seconds": round(time.perf_counter() - t0, 1)}
    # The children agree among themselves before they exit, but a child that is killed at the limit or dies after that agreement leaves ITS
    # parent with another verdict than the others, and ranks that then meet with different transports hang in the rendezvous.
    # So the parents agree too, without any transport: one file per rank in /dev/shm (one node), keyed by the launcher's pid and port;
    # RCCL only if every rank's child said ok.
    # (the ranks of one run share their launcher's pid; a launcher that gives every rank a parent of its own sets VNR_BENCH_RUN_ID)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local_world < world:
        # more than one node: the verdict files are visible to this node's ranks only.  The children have agreed among themselves
        # over the control plane, which spans the nodes; a child killed after that agreement is the case left open here.
        if ok:
            return "rccl", probe
        probe["fallback"] = "none: the host-staged transport is for one node; this run has ranks on several"
        print(f"[bench] rank {rank}: RCCL probe failed on a multi-node run ({said[:300]})", file=sys.stderr, flush=True)
        sys.exit(3)
    with open(f"{base}_{rank}.tmp", "w") as f:
        f.write("ok" if ok else "failed")
    os.replace(f"{base}_{rank}.tmp", f"{base}_{rank}")
    t_wait = time.perf_counter()
    verdicts = {}
    while len(verdicts) < world and time.perf_counter() - t_wait < limit + 60:
        for r in range(world):
            if r not in verdicts and os.path.exists(f"{base}_{r}"):
                verdicts[r] = open(f"{base}_{r}").read().strip()
        if len(verdicts) < world:
            time.sleep(0.05)
    agreed = len(verdicts) == world and all(v == "ok" for v in verdicts.values())
    if ok and not agreed:
        probe["rccl"] = ("FAILED on another rank (" + ", ".join(f"rank {r}: {verdicts.get(r, 'no verdict')}" for r in range(world)) + "); this rank's child: " + said)[:600]
    ok = agreed
    import atexit
    t_decided = time.perf_counter()

    def remove_verdict():
        # the others poll every 50 ms: a process that ends right after deciding (a failed self-test, a test) must leave its verdict readable
        # for them first; a bench run ends a minute later and does not wait here
        time.sleep(max(0.0, 2.0 - (time.perf_counter() - t_decided)))
        if os.path.exists(f"{base}_{rank}"):
            os.remove(f"{base}_{rank}")
    atexit.register(remove_verdict)
    if ok:
        return "rccl", probe
    probe["fallback"] = "shm: host-staged shared-memory transport (one node); the numbers of this line are NOT RCCL's"
    print(f"[bench] rank {os.environ.get('RANK', '0')}: RCCL probe failed ({said[:300]}); continuing over shared memory", file=sys.stderr, flush=True)
    return "shm", probe


def visible_gpu_count():
    """how many GPUs this process could open, WITHOUT a HIP call (the launching parent must never touch the GPU): the KFD topology's nodes
    that have SIMDs, whose properties this process may read (a container sees the other GPUs of its host as nodes it may not read) and whose
    render node it may open, cut by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None when the topology cannot be read at all."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(base)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(base, node, "properties")) if len(line.split()) >= 2)
        except (OSError, ValueError):
            continue                      # not ours
        if int(props.get("simd_count", "0")) <= 0:
            continue                      # a CPU node
        render = os.path.join("/dev/dri", "renderD" + props.get("drm_render_minor", "-1"))
        if os.path.isdir("/dev/dri") and not os.access(render, os.R_OK | os.W_OK):
            continue
        n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            n = min(n, len([x for x in os.environ[var].split(",") if x.strip()]))
    return n if n > 0 else None           # (zero: the rules above do not fit this machine; let the ranks' RCCL probe find out)


def launch_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: THIS process becomes the launcher.  It starts N fresh copies of
    itself (fork + exec of a process that has not touched the GPU and never will), one per rank, with the environment torch.distributed.run
    would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), forwards rank 0's ONE JSON line, and exits non-zero as soon
    as any rank does (the others are ended).  A bare `--gpus 8` can therefore not be mistaken for a one-GPU run."""
    import signal
    import socket
    import subprocess
    import threading
    n = a.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                "VNR_BENCH_RUN_ID": f"self{os.getpid()}", "VNR_BENCH_LAUNCHED_BY": "bench.py", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    devices = visible_gpu_count()
    if devices is not None and devices < n and not env.get("VNR_AMD_DIST_TRANSPORT"):
        # RCCL refuses two ranks on one device: no point in probing it.  The line then says `strong-over-shm (NOT an RCCL measurement)`.
        print(f"[bench] --gpus {n} on a box with {devices} visible GPU(s): ranks share devices (rank % {max(devices, 1)}) and exchange over the "
              "host-staged shared-memory transport; this is a rehearsal of the code path, NOT an N-GPU measurement", file=sys.stderr, flush=True)
        env["VNR_AMD_DIST_TRANSPORT"] = "shm"
        env["VNR_BENCH_TRANSPORT_REASON"] = f"{n} ranks on {devices} visible GPU(s)"
    argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for rank in range(n):
        e = dict(env)
        e.update({"RANK": str(rank), "LOCAL_RANK": str(rank)})
        procs.append(subprocess.Popen(argv, env=e, stdout=subprocess.PIPE, text=True))

    def end_all(*_):
        for p in procs:
            if p.poll() is None:
                p.terminate()
    if threading.current_thread() is threading.main_thread():
        for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            signal.signal(sig, lambda *_: (end_all(), os._exit(130)))

    def forward(rank, p):   # rank 0's stdout is this program's stdout; what another rank prints goes to stderr, labelled
        for line in p.stdout:
            if rank == 0:
                sys.stdout.write(line); sys.stdout.flush()
            else:
                sys.stderr.write(f"[rank {rank}] {line}"); sys.stderr.flush()
    threads = [threading.Thread(target=forward, args=(r, p), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                print(f"[bench] rank {r} of {n} exited with code {code}; ending the other ranks", file=sys.stderr, flush=True)
                time.sleep(2.0)   # the ranks agree on a failed self-test among themselves: let them say so before they are ended
                end_all()
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    sys.exit(rc)


def main():
    a = parse()
    if a.probe_collectives:
        probe_collectives_child()
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:
            launch_ranks(a)        # does not return
    elif int(os.environ["WORLD_SIZE"]) != a.gpus:
        # never a line whose n_gpus is not what was asked for
        print(f"[bench] --gpus {a.gpus} but the launcher's WORLD_SIZE is {os.environ['WORLD_SIZE']}: refusing to measure something else than was asked for "
              "(start one process per GPU, or start `python bench.py --gpus N` bare and it starts its ranks itself)", file=sys.stderr, flush=True)
        sys.exit(2)
    if os.environ.get("VNR_BENCH_DUMP_AFTER"):  # diagnostics: where does a run that hangs under the profiler stand?
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VNR_BENCH_DUMP_AFTER"]), exit=False, file=sys.stderr)
    # a rank that hangs without closing its socket must not stall a bench run for ever (the library itself waits without bound)
    os.environ.setdefault("VNR_AMD_DIST_STEADY_TIMEOUT", "1800")
    transport, transport_probe = choose_transport()
    ctx = dist.init_from_env(transport=transport)
    if a.gpus != ctx.world:   # (cannot happen after the checks above; a line with another n_gpus than was asked for must not exist)
        print(f"[bench] rank {ctx.rank}: --gpus {a.gpus} but the process group has {ctx.world} rank(s)", file=sys.stderr, flush=True)
        os._exit(2)
    L = lib()
    dims = (a.size, a.size, a.size)
    pls = a.per_level_scale if a.per_level_scale > 0 else float(np.exp(np.log(a.size / 16.0) / max(a.levels - 1, 1)))
    os.environ.setdefault("VNR_AMD_INIT_SEED", "20240611")  # reproducible initial parameters (N > 1: rank 0's are broadcast anyway)
    kind = a.volume if a.volume != "auto" else ("vortex" if a.size <= 256 else "perlin")

    # ---- setup (untimed) -------------------------------------------------------------------------------------
    t_setup = time.perf_counter()
    if kind == "vortex":
        sv = api.vnrCreateSimpleVolume(syn.vortex_volume(a.size, seed=1234))
        volume_desc = f"{a.size}^3 synthetic vortex-tube field (seed 1234; stand-in for vorts1, SURVEY 8c)"
    else:
        sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
        volume_desc = f"{a.size}^3 synthetic Perlin fBm volume (seed 42)"
    cfg = syn.model_config(n_levels=a.levels, n_features=a.features, log2_hashmap_size=a.log2_hashmap_size, base_resolution=16,
                           n_hidden_layers=a.hidden_layers, per_level_scale=pls)
    # ground-truth macrocell (identical on every rank, so tiles compose exactly)
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
    info = api.neural_info(nv)
    check(L.vnrAmdSynchronize())
    # training: the model the frames are rendered from.  The first steps (allocations, first launches) are not timed, the last <= 64
    # steps are profiled kernel by kernel (HIP events between the phases, which cost a few microseconds each) and not timed either;
    # `train_ms_per_step` is the steady-state step in between.  The step count of the model is a.train_steps whatever the split.
    n_warm = min(100, a.train_steps // 10)
    n_tail = min(64, a.train_steps - n_warm)
    n_timed = a.train_steps - n_warm - n_tail
    if n_warm:
        dist.train_data_parallel(ctx, nv, n_warm, fast_mode=True)
    check(L.vnrAmdSynchronize())
    t_train = time.perf_counter()
    if n_timed:
        dist.train_data_parallel(ctx, nv, n_timed, fast_mode=True)
    check(L.vnrAmdSynchronize())
    train_ms = (time.perf_counter() - t_train) * 1e3 / max(n_timed, 1)
    check(L.vnrAmdNeuralVolumeSetTrainProfiling(nv.h, 1))
    if n_tail:
        dist.train_data_parallel(ctx, nv, n_tail, fast_mode=True)
    check(L.vnrAmdSynchronize())
    import ctypes as C
    phase_ms = (C.c_double * 5)()
    n_prof = C.c_int()
    check(L.vnrAmdNeuralVolumeGetTrainProfile(nv.h, phase_ms, C.byref(n_prof)))
    check(L.vnrAmdNeuralVolumeSetTrainProfiling(nv.h, 0))
    if n_timed == 0:   # (very short training runs, --train-steps < 75: nothing was left for the wall-clock leg; the profiled kernels stand in)
        train_ms = float(sum(phase_ms[i] for i in range(5)))
    train_loss = api.vnrNeuralVolumeGetTrainingLoss(nv)
    psnr = None
    if not a.no_psnr and ctx.rank == 0:
        psnr = api.vnrNeuralVolumeGetPSNR(nv)

    colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=a.opacity_scale)
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors)
    api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    cam = syn.oblique_camera(dims, distance_scale=a.camera_distance)
    camera = api.vnrCreateCamera()
    api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])

    def make_renderer(volume, device_output=True, profiling=True):
        rr = api.vnrCreateRenderer(volume)
        api.vnrRendererSetTransferFunction(rr, tfn)
        api.vnrRendererSetCamera(rr, camera)
        api.vnrRendererSetFramebufferSize(rr, (a.fb, a.fb))
        api.vnrRendererSetMode(rr, a.mode)
        if device_output:
            api.vnrRendererSetOutputAsDeviceFramebuffer(rr, True)
        api.vnrRendererSetProfiling(rr, profiling)
        return rr

    # HIP events around every launch of the evaluation kernel, on the stream it is launched on.  With more than one rank they stay
    # out of the timed region (two timed events per launch are 6 % of the frame time of a 1/8 share, docs/history/DESIGN_r01-r03.md 6) and an
    # un-timed leg behind it measures the launches instead.
    events_in_timed_region = not a.no_kernel_events and ctx.world == 1
    ren = make_renderer(nv, device_output=False, profiling=events_in_timed_region)
    sr = dist.ShardedRenderer(ctx, ren, a.fb, a.fb)
    setup_s = time.perf_counter() - t_setup

    # ---- first contact (N > 1): every collective this run uses, once, on patterned buffers, each with a deadline ---------------
    # A collective that hangs or returns other values is named on stderr and EVERY rank exits non-zero (the ranks agree over the
    # control plane, which does not depend on the transport under test); no JSON line is printed.  os._exit: a fresh process exit
    # that runs no finaliser against a stream that may hold a collective that will never complete (and no re-exec anywhere).
    self_test = None
    if ctx.distributed:
        ok, text = dist.self_test(deadline_s=float(os.environ.get("VNR_BENCH_SELFTEST_DEADLINE", "60")))
        all_ok = dist.all_reduce_host([1.0 if ok else 0.0], dist.MIN)[0] >= 1.0
        if not ok:
            print(f"[bench] rank {ctx.rank}: collective self-test FAILED: {text}", file=sys.stderr, flush=True)
        if not all_ok:
            if ok:
                print(f"[bench] rank {ctx.rank}: another rank's collective self-test failed; exiting", file=sys.stderr, flush=True)
            os._exit(3)
        self_test = text

    # ---- warm-up + timed region ---------------------------------------------------------------------------------
    for _ in range(a.warmup):
        sr.render()
    sr.flush()
    dist.barrier(ctx)
    t0 = time.perf_counter()
    samples = slots = 0
    infer_ms = 0.0
    union_ms = 0.0
    launches = iters = 0
    def add(st):
        nonlocal samples, slots, infer_ms, union_ms, launches, iters
        samples += st["n_samples"]; slots += st["n_reference_slots"]; infer_ms += st["infer_kernel_ms"]; union_ms += st["infer_union_ms"]
        launches += st["infer_kernel_launches"]; iters = st["n_iterations"]

    # a pipeline of depth one (vnrAmdRendererRenderPipelined): call k enqueues frame k, completes frame k - 1 and hands it out
    # (N > 1: all-gathered and assembled beside the rendering of frame k); the flush completes the last frame, so K frames are
    # rendered, completed (and gathered) inside the timed region
    for k in range(a.steps):
        sr.render()
        if k > 0:
            add(sr.completed_stats())
    sr.flush()
    add(sr.completed_stats())
    st = sr.completed_stats()
    dist.barrier(ctx)
    elapsed = time.perf_counter() - t0
    rays_hit = st["n_rays_hit"]
    samples_evt, evt_frames = samples, a.steps     # the frames the kernel events cover
    if ctx.world > 1 and not a.no_kernel_events:   # un-timed: the same frames with the events on
        api.vnrRendererSetProfiling(ren, True)
        keep = (samples, slots, iters)
        samples = slots = 0
        infer_ms = union_ms = 0.0
        launches = 0
        evt_frames = max(5, min(a.steps // 2, 30))
        for k in range(evt_frames):
            sr.render()
            if k > 0:
                add(sr.completed_stats())
        sr.flush()
        add(sr.completed_stats())
        samples_evt = samples
        samples, slots, iters = keep
        api.vnrRendererSetProfiling(ren, False)
        dist.barrier(ctx)
    # (N > 1, un-timed) where a rank's frame time goes, per rank, so that a bad scaling curve can be read from ONE record: the share alone
    # (vnrRender completes the rank's share, host-synchronous), then the gather (in-place all-gather + assembly, vnrAmdRendererGatherFrame)
    per_rank = None
    if ctx.world > 1:
        n_pr = max(5, min(a.steps // 2, 20))
        t_share = t_gather = 0.0
        for k in range(n_pr + 2):
            dist.barrier(ctx)
            t_a = time.perf_counter()
            api.vnrRender(ren)
            check(L.vnrAmdSynchronize())
            t_b = time.perf_counter()
            L.vnrAmdRendererGatherFrame(ren.h)
            t_c = time.perf_counter()
            if k >= 2:
                t_share += (t_b - t_a) * 1e3 / n_pr
                t_gather += (t_c - t_b) * 1e3 / n_pr
        def slot_vec(v):
            return [v if r == ctx.rank else 0.0 for r in range(ctx.world)]
        per_rank = {"share_ms": [round(x, 4) for x in dist.all_reduce_host(slot_vec(t_share), dist.SUM)],
                    "gather_ms": [round(x, 4) for x in dist.all_reduce_host(slot_vec(t_gather), dist.SUM)],
                    "train_step_ms": [round(x, 4) for x in dist.all_reduce_host(slot_vec(train_ms), dist.SUM)],
                    "train_exchange_ms": [round(x, 4) for x in dist.all_reduce_host(slot_vec(float(phase_ms[4])), dist.SUM)],
                    "what": f"per rank, un-timed leg of {n_pr} frames rendered one at a time: share_ms = vnrRender of the rank's share until its "
                            "kernels are done (host-synchronous), gather_ms = in-place all-gather + assembly behind it (vnrAmdRendererGatherFrame); "
                            "train_step_ms = wall per data-parallel step, train_exchange_ms = its last phase (the rank's 1/N optimizer slice + "
                            "waiting for reduce-scatter / all-gather of the gradient ranges)"}
    brick_state = api.neural_brick_image(nv)

    # ---- un-timed: the neural frame against the frame of the ground-truth volume (same camera / TFN / mode / macrocell) --------
    image = None
    if ctx.world == 1 and not a.no_psnr:
        def one_frame(volume):
            rr = make_renderer(volume, device_output=False, profiling=False)
            api.vnrRender(rr)
            return api.vnrRendererMapFrame(rr).astype(np.float64).copy()

        f_nn, f_gt = one_frame(nv), one_frame(sv)
        d = f_nn - f_gt
        mse = float((d[..., :3] ** 2).mean())
        image = {"what": "first frame of the neural volume vs the same frame marched through the ground-truth volume (rgb, peak 1)",
                 "psnr_db": round(10.0 * np.log10(1.0 / mse), 2) if mse > 0 else None,
                 "l2_per_pixel_mean": round(float(np.sqrt((d ** 2).sum(axis=2)).mean()), 6),
                 "l2_per_pixel_max": round(float(np.sqrt((d ** 2).sum(axis=2)).max()), 5)}

    # ---- un-timed extra legs (one GPU only) ------------------------------------------------------------------------------------
    # (1) the dominant kernel with the GPU to itself.  The default renderer runs two ray halves on two HIP streams, so a launch of
    # the fused kernel shares the GPU with the other half's kernels and its HIP-event duration says little about the kernel.  The
    # same frames on ONE stream (nothing else resident while the kernel runs) give the per-launch figure the roofline fraction is
    # meant to be.
    n_leg = max(5, min(a.steps // 2, 30))
    alone = None
    if ctx.world == 1 and not a.no_alone and os.environ.get("VNR_AMD_RENDER_HALVES", "2") != "1":
        os.environ["VNR_AMD_RENDER_HALVES"] = "1"
        ren1 = make_renderer(nv)
        del os.environ["VNR_AMD_RENDER_HALVES"]
        fps1, a_samples, a_ms, a_launches, _ = untimed_frames(ren1, n_leg)
        alone = {"frames": n_leg, "fps": round(fps1, 2), "samples": a_samples, "ms": a_ms, "launches": a_launches}
        del ren1
    # (2) the same frames WITHOUT the brick image: what an application that trains while it renders gets (every optimizer step
    # drops the image, so it never exists there); the image costs `inference_cache.brick_image_bytes` of HBM
    brick_off = None
    if ctx.world == 1 and not a.no_brick_off and brick_state["in_use"]:
        check(L.vnrAmdNeuralVolumeSetBrickImageMode(nv.h, 0))
        ren2 = make_renderer(nv)
        fps2, b_samples, b_ms, b_launches, b_union = untimed_frames(ren2, n_leg)
        del ren2
        os.environ["VNR_AMD_RENDER_HALVES"] = "1"
        ren3 = make_renderer(nv)
        del os.environ["VNR_AMD_RENDER_HALVES"]
        _, c_samples, c_ms, c_launches, _ = untimed_frames(ren3, max(5, n_leg // 2))
        del ren3
        brick_off = {"frames": n_leg, "fps": fps2, "samples": b_samples, "ms": b_ms, "launches": b_launches, "union": b_union,
                     "alone_samples": c_samples, "alone_ms": c_ms, "alone_launches": c_launches}
        check(L.vnrAmdNeuralVolumeSetBrickImageMode(nv.h, -1))

    # (3) what a budget for the image buys: the hashed levels are taken finest first while they fit, so a budget is a number of levels.
    # One un-timed leg per budget, each a fresh renderer on the same model: frames/s (two streams) and the evaluation kernel's union fraction.
    budget_table = None
    if ctx.world == 1 and not a.no_brick_table and not a.no_brick_off and brick_state["in_use"] and a.mode == 5:
        budget_table = []
        full = int(brick_state["bytes"])
        for gb in (0.25, 0.75, 1.7, 3.8):
            if gb * 2**30 >= full:
                continue
            api.neural_set_brick_budget(nv, int(gb * 2**30))
            check(L.vnrAmdNeuralVolumeSetBrickImageMode(nv.h, 1))     # build at the next launch
            rr = make_renderer(nv)
            fps_b, s_b, ms_b, l_b, u_b = untimed_frames(rr, max(5, n_leg // 2))
            del rr
            stb = api.neural_brick_image(nv)
            budget_table.append({"budget_gib": gb, "image_bytes": int(stb["bytes"]), "levels_in_image": [l for l in range(32) if stb["levels"] >> l & 1],
                                 "fps": round(fps_b, 2), "samples": s_b, "union_ms": u_b})
        api.neural_set_brick_budget(nv, 0)
        check(L.vnrAmdNeuralVolumeSetBrickImageMode(nv.h, -1))

    # (4) the reference's interactive loop (apps/int_dual_volume.cpp:631-672): every displayed frame is followed by
    # vnrNeuralVolumeTrain(nv, train_steps, /*fast_mode=*/false) -- the optimizer steps plus, per step, the macrocell's update from the step's samples
    # and, per call, the refresh of its max-opacity (core/network.cu:231-259, 770-779).  This is the "instant" in instantvnr.  Un-timed for `value`.
    interactive = interactive_leg(a, nv, make_renderer, brick_off) if ctx.world == 1 and not a.no_interactive and a.mode == 5 else None

    c5 = c5_leg(a, ctx) if a.c5 else None

    if ctx.distributed:
        elapsed = dist.all_reduce_host([elapsed], dist.MAX)[0]          # MAX over ranks of the timed region
        samples_all, slots_all, rays_f = dist.all_reduce_host([float(samples), float(slots), float(rays_hit)], dist.SUM)
        rays_hit = int(rays_f)
    else:
        samples_all, slots_all = float(samples), float(slots)

    if ctx.rank != 0:
        dist.barrier(ctx)
        return
    fps = a.steps / elapsed
    # the renderer counts SHADED samples; with gradient shading (mode 8) the network evaluates 4 coordinates for each of them,
    # and every rate below is per network evaluation
    evals_per_sample = 4 if a.mode in (8, 9) else 1
    shaded_samples_per_frame = int(samples_all / a.steps)
    samples *= evals_per_sample
    samples_all *= evals_per_sample
    bytes_per_sample = 12 + info["n_levels"] * 8 * info["n_features_per_level"] * 2 + 4
    in_pad = info["padded_width"]
    flops_per_sample = 2 * (in_pad * 64 + (info["n_hidden_layers"] - 1) * 64 * 64 + 64)
    # dominant kernel: fused hash-grid gather + MLP.  achieved = algorithmic bytes of the samples this rank's
    # launches processed / summed launch durations (HIP events on the render stream), i.e. per-launch average.
    samples_evt *= evals_per_sample
    achieved = (samples_evt * bytes_per_sample) / (infer_ms * 1e-3) / 1e9 if infer_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": "fused_infer_kernel (hash-grid gather + 3x64 MLP on MFMA)",
                "algorithmic_bytes_per_sample": bytes_per_sample, "flops_per_sample": flops_per_sample,
                "avg_launch_ms": round(infer_ms / max(launches, 1), 4), "launches": launches,
                "mfma_tflops": round(samples_evt * flops_per_sample / (infer_ms * 1e-3) / 1e12, 2) if infer_ms > 0 else 0.0}
    if ctx.world > 1:
        roofline["events"] = (f"rank 0's launches in an un-timed leg of {evt_frames} frames behind the timed region: with more than one rank the HIP events "
                              "around every launch stay out of the timed frames (they cost 6 % of a 1/8 share's frame time)")
    # The renderer runs the rays as 2 halves on 2 streams by default (march of one half overlaps inference of the other,
    # and the two halves' inference kernels overlap each other), so a launch's HIP-event duration includes time it shares
    # the GPU: `achieved`/`frac` (defined per launch) drop although the frame gets faster.  The frame-level figure below
    # does not depend on scheduling: algorithmic bytes of all live samples of a frame / frame time.
    halves = 1 if os.environ.get("VNR_AMD_RENDER_HALVES", "2") == "1" else 2
    if union_ms > 0:
        u = (samples_evt * bytes_per_sample) / (union_ms * 1e-3) / 1e9
        roofline["union"] = {"what": "algorithmic bytes of all launches / the time during which at least one launch of the kernel was "
                                     "running (union of the HIP-event intervals of both streams): overlapping launches count once",
                             "achieved": round(u, 1), "frac": round(u / HBM_PEAK_GBS, 4), "ms_per_frame": round(union_ms / evt_frames, 4)}
        # beside `frac` (per launch, HIP events of ONE stream: two streams' launches overlap and are counted twice there): the same bytes
        # over the time during which the kernel was running at all.  This is the figure to read the kernel by.
        roofline["union_frac"] = roofline["union"]["frac"]
    if alone and alone["ms"] > 0:
        ev = alone["samples"] * evals_per_sample
        a_gbs = ev * bytes_per_sample / (alone["ms"] * 1e-3) / 1e9
        roofline["alone"] = {"what": "same frames on ONE HIP stream, un-timed extra leg of this run: nothing else is resident while the kernel runs",
                             "achieved": round(a_gbs, 1), "frac": round(a_gbs / HBM_PEAK_GBS, 4),
                             "avg_launch_ms": round(alone["ms"] / max(alone["launches"], 1), 4), "launches": alone["launches"],
                             "msamples_per_s": round(ev / (alone["ms"] * 1e-3) / 1e6, 1), "frames": alone["frames"],
                             "fps_one_stream": alone["fps"],
                             "mfma_tflops": round(ev * flops_per_sample / (alone["ms"] * 1e-3) / 1e12, 2)}
    # the matrix cores' share of the kernel (north star: "MFMA utilisation on the MLP against the chip's peaks"): flops of this run's samples over
    # the union time, against the dense fp16 peak; the counter-based figure comes from a separate rocprofv3 --pmc pass (tools/r06_infer_bound.sh, pass mfma_on)
    MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense fp16 / bf16
    if union_ms > 0:
        tf = samples_evt * flops_per_sample / (union_ms * 1e-3) / 1e12
        roofline["mfma"] = {"tflops": round(tf, 1), "peak_tflops": MFMA_PEAK_TFLOPS, "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                            "what": "the MLP's flops (20 608 per sample for 3x64 on a 32-wide encoding) over the union of the launch intervals; the kernel is bound by "
                                    "its hash-grid gathers, the one dense contraction on the path is a tenth of it",
                            "util_by_counters": None,
                            "util_source": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) of a separate rocprofv3 --pmc pass (profiles/*_mfma_pmc.json); "
                                           "null unless that pass was made on the default workload, one GPU, with the sources the kernel is built from now"}
    roofline["concurrency"] = (f"{halves} ray halves on {halves} HIP streams; launch durations are per stream and overlap"
                               if halves == 2 else "1 stream: launches run alone")
    frame_gbs = (samples / a.steps) * bytes_per_sample / (elapsed / a.steps) / 1e9
    roofline["frame_algorithmic_gbs"] = round(frame_gbs, 1)
    roofline["frame_frac"] = round(frame_gbs / HBM_PEAK_GBS, 4)
    # HBM-side traffic of the dominant kernel: PMC counters cannot be collected inside this run (separate rocprofv3
    # --pmc passes, tools/run_pmc.sh); a committed result applies only to the exact default workload on one GPU with
    # the same stream configuration it was measured with.
    default_workload = (a.size, a.fb, a.levels, a.features, a.log2_hashmap_size, a.hidden_layers, a.per_level_scale,
                        a.train_steps, a.opacity_scale, a.camera_distance, a.mode, kind) == (1024, 1024, 16, 2, 22, 3, 0.0, 1500, 0.06, 1.1, 5, "perlin")
    # counter results come from separate passes: they enter the line only when the pass names the sources the kernel is compiled from NOW
    # (a kernel change otherwise leaves them silently stale: VERDICT r04 weak 7) and only for the workload they were measured on
    roofline["kernel_source_sha16"] = sources_sha16(EVAL_KERNEL_SOURCES)

    def counters_of(name):
        path = os.path.join(ROOT, "profiles", name)
        if not (default_workload and ctx.world == 1 and os.path.exists(path)):
            return None, None
        doc = json.load(open(path))
        fresh = doc.get("source_files") and doc.get("source_sha16") == sources_sha16(doc["source_files"])
        return (doc, path) if fresh else (None, path)

    mfma_doc, _ = counters_of("r06_mfma_pmc.json")
    if mfma_doc and "mfma" in roofline and brick_state["in_use"]:
        roofline["mfma"]["util_by_counters"] = mfma_doc["util_by_counters"]
        roofline["mfma"]["util_source"] += "; profiles/r06_mfma_pmc.json"
    pmc, pmc_path = counters_of("r06_pmc_traffic.json")
    if pmc and brick_state["in_use"]:
        leg = pmc["one_stream" if halves == 1 else "two_streams"]
        per_sample = leg.get("bytes_per_sample") or leg["traffic_over_algorithmic"] * bytes_per_sample
        roofline["traffic"] = round(per_sample * samples / max(launches, 1))
        roofline["traffic_unit"] = "bytes per launch (L2<->fabric reads x2-corrected + writes; includes Infinity-Cache hits)"
        roofline["algorithmic_bytes_per_launch"] = round(samples * bytes_per_sample / max(launches, 1))
        # the bound the counters point at: what actually crosses the fabric per second while the kernel runs
        if union_ms > 0:
            t_gbs = per_sample * samples_evt / (union_ms * 1e-3) / 1e9
            roofline["traffic_frac"] = {"what": "measured fabric bytes per sample x this run's samples / the union of the launch intervals / 8 TB/s: the kernel "
                                                "is no longer bound by bytes (L1 / texture-addresser latency, DESIGN 4.1); the algorithmic fraction flatters it",
                                        "achieved": round(t_gbs, 1), "frac": round(t_gbs / HBM_PEAK_GBS, 4), "bytes_per_sample": round(per_sample, 1)}
        roofline["traffic_note"] = (f"bytes per sample measured in separate rocprofv3 --pmc passes of the same frame, not in this run: {os.path.relpath(pmc_path, ROOT)} "
                                    "(traffic / algorithmic = %.2f; without the brick image: roofline.brick_off.traffic_bytes_per_sample)" % (per_sample / bytes_per_sample))
        if "alone" in roofline:
            one = pmc["one_stream"]
            ps1 = one.get("bytes_per_sample") or one["traffic_over_algorithmic"] * bytes_per_sample
            roofline["alone"]["traffic"] = round(ps1 * alone["samples"] / max(alone["launches"], 1))
    else:
        roofline["traffic_note"] = ("null: the committed PMC passes (profiles/r06_pmc_traffic.json) describe the default workload on one GPU with the brick image, for the "
                                    "kernel sources named by their source_sha16" + ("; this build's differ" if pmc_path and default_workload and ctx.world == 1 else ""))
    if brick_off:
        ev = brick_off["samples"] * evals_per_sample
        leg = {"what": "un-timed leg of this run with the brick image switched off: the configuration of an application that trains while it renders "
                       "(every optimizer step drops the image)",
               "fps": round(brick_off["fps"], 2), "frames": brick_off["frames"],
               "frame_frac": round((ev / brick_off["frames"]) * bytes_per_sample * brick_off["fps"] / 1e9 / HBM_PEAK_GBS, 4)}
        if brick_off["union"] > 0:
            leg["kernel_union_frac"] = round(ev * bytes_per_sample / (brick_off["union"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if brick_off["alone_ms"] > 0:
            ev1 = brick_off["alone_samples"] * evals_per_sample
            leg["kernel_alone_frac"] = round(ev1 * bytes_per_sample / (brick_off["alone_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            leg["kernel_alone_msamples_per_s"] = round(ev1 / (brick_off["alone_ms"] * 1e-3) / 1e6, 1)
        off, _ = counters_of("r06_pmc_traffic_brick_off.json")
        if off:
            leg["traffic_bytes_per_sample"] = {"two_streams": round(off["two_streams"]["bytes_per_sample"], 1), "one_stream": round(off["one_stream"]["bytes_per_sample"], 1),
                                               "note": "separate rocprofv3 --pmc passes of this frame with VNR_AMD_BRICK=0 (profiles/r06_pmc_traffic_brick_off.json), not this run"}
        roofline["brick_off"] = leg

    # ---- training step: algorithmic bytes per SURVEY 8(d) and the live per-kernel split -------------------------------------------
    B, Lv, F, n_params = 65536, info["n_levels"], info["n_features_per_level"], info["n_params"]
    fwd_bytes = bytes_per_sample * B
    bwd_bytes = Lv * 8 * F * 2 * 2 * B
    opt_bytes = n_params * (2 + 4 + 4 + 4) * 2
    step_bytes = fwd_bytes + bwd_bytes + opt_bytes
    overlap = os.environ.get("VNR_AMD_TRAIN_OVERLAP", "0") not in ("", "0") and not ctx.distributed   # (opt-in: INTEGRATION.md 6)
    names = ["forward (fused encode + MLP, keeps activations)", "loss + MLP backward (MFMA)",
             "weight gradients (MFMA, block partials summed in order)" + ("; on a side stream beside the grid backward: only the fork is left in this phase" if overlap else ""),
             ("grid backward (persistent packed-fp16 atomic scatter, beside it the dense levels' LDS scatter and the weight gradients; ends at the join)" if overlap else
              "grid backward (the dense coarse levels' LDS scatter, then the hashed levels' packed-fp16 atomic scatter)"),
             "optimizer (Adam, fp32 master, fp16 gradient)" + (" incl. waiting for the gradient exchange" if ctx.distributed else "")]
    kernel_ms = [float(phase_ms[i]) for i in range(5)]
    step_kernel_ms = sum(kernel_ms)
    # The step is not a byte sweep: its largest phase, the grid backward, is priced per memory-side atomic REQUEST (MI355X_MICROARCH.md
    # "Global float atomics": a 256-B wave instruction leaves L2 as four 64-B requests, one per ~50 ns and CU = 4 x 256 CUs / 50 ns ~ 20.5 G
    # requests/s chip-wide), so that phase is held against that rate.  Requests per step, from the level table: the atomic kernel's lanes
    # are (sample, x bit, feature pair) with the two x-neighbours of a corner pair adjacent, i.e. ONE request per (sample, level, yz corner)
    # for n_features <= 8; the dense coarse levels go through LDS tiles and flush at most entries x F x 2 B / 64 B requests per slice.
    ATOMIC_PEAK_GREQ = 4 * 256 / 50e-9 / 1e9
    plan = api.neural_grid_backward_plan(nv, B)   # the library's own layout and request count (environment overrides and level masking included: ADVICE r05)
    lds_levels, req_atomic, req_flush_max = plan["lds_levels"], plan["atomic_requests"], plan["flush_requests_at_most"]
    gb_ms = kernel_ms[3]
    greq = (req_atomic + req_flush_max) / (gb_ms * 1e-3) / 1e9 if gb_ms > 0 else 0.0
    atomic_doc, _ = counters_of("r06_train_atomic_pmc.json")
    train_roofline = {"bound": "memory-side atomic requests (the grid backward, the step's largest phase)", "unit": "G requests/s",
                      "peak": round(ATOMIC_PEAK_GREQ, 1), "achieved": round(greq, 2), "frac": round(greq / ATOMIC_PEAK_GREQ, 4),
                      "phase": "grid backward", "phase_ms": round(gb_ms, 4), "phase_share_of_step": round(gb_ms / max(step_kernel_ms, 1e-9), 3),
                      "requests_per_step": {"atomic_kernel": req_atomic, "lds_tile_flush_at_most": req_flush_max, "levels_through_lds": lds_levels,
                                            "what": "atomic kernel: 65 536 samples x 4 yz-corner pairs x levels, one 64-B request each; LDS tiles: slices x "
                                                    "level bytes / 64 (pairs nobody touched are skipped, so fewer)"},
                      "requests_by_counters": None if not atomic_doc else atomic_doc.get("requests_per_step"),
                      "counters_source": "TCC_EA0_ATOMIC_sum of grid_backward_kernel + grid_backward_lds_kernel, separate rocprofv3 --pmc pass (profiles/r06_train_atomic_pmc.json)",
                      "peak_source": "MI355X_MICROARCH.md 'Global float atomics': ~1.3 TB/s of added bytes = one 256-B wave instruction (four 64-B requests) per ~50 ns and CU",
                      "ms_per_step_wall": round(train_ms, 4), "ms_per_step_kernels": round(step_kernel_ms, 4), "profiled_steps": int(n_prof.value),
                      "kernels_ms": {n: round(v, 4) for n, v in zip(names, kernel_ms)},
                      "dense_sweep": {"what": "SURVEY 8(d)'s algorithmic bytes (528 B x B forward, L*8*F*2*2 B x B scatter, n_params x 14 B read + 14 B written) over the step: "
                                              "NOT a bound of this step, which skips untouched grid entries (a batch touches about an eighth) -- kept for continuity with rounds 1-4",
                                      "algorithmic_bytes_per_step": step_bytes, "gbs": round(step_bytes / (train_ms * 1e-3) / 1e9, 1),
                                      "frac_of_hbm_peak": round(step_bytes / (train_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}

    budget_rows = None
    if budget_table is not None:
        budget_rows = [{"budget_gib": 0, "image_bytes": 0, "levels_in_image": [], "fps": roofline.get("brick_off", {}).get("fps"),
                        "kernel_union_frac": roofline.get("brick_off", {}).get("kernel_union_frac")}]
        for row in budget_table:
            ev = row["samples"] * evals_per_sample
            budget_rows.append({"budget_gib": row["budget_gib"], "image_bytes": row["image_bytes"], "levels_in_image": row["levels_in_image"], "fps": row["fps"],
                                "kernel_union_frac": round(ev * bytes_per_sample / (row["union_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if row["union_ms"] > 0 else None})
        budget_rows.append({"budget_gib": "default", "image_bytes": int(brick_state["bytes"]), "levels_in_image": [l for l in range(32) if brick_state.get("levels", 0) >> l & 1],
                            "fps": round(fps, 2), "kernel_union_frac": roofline.get("union", {}).get("frac")})
    # `value` depends on the inference cache: say so where a truncated record still shows it
    off_fps = roofline.get("brick_off", {}).get("fps")
    cache_text = (f"inference cache (brick image) IN USE: {brick_state['bytes'] / 1e9:.2f} GB beside a {info['n_params'] * 2 / 1e6:.0f} MB model"
                  + (f", WITHOUT it (an application that trains while it renders) {off_fps:.1f} frames/s" if off_fps else "")
                  if brick_state["in_use"] else "inference cache (brick image) not in use")
    scaling = "strong"
    if ctx.distributed and ctx.transport != "rccl":   # the host-staged transport, as a fallback or because the environment chose it: never read as RCCL's curve
        scaling = f"strong-over-{ctx.transport} (NOT an RCCL measurement)"
    out = {
        "metric": "fps at 1024^2 on 1024^3 volume" if (a.size, a.fb, a.mode) == (1024, 1024, 5) else f"fps at {a.fb}^2 on {a.size}^3 volume, rendering mode {a.mode}" if a.mode != 5 else f"fps at {a.fb}^2 on {a.size}^3 volume",
        "value": round(fps, 3), "unit": "frames/s", "n_gpus": ctx.world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{workload_name(a)}: {volume_desc}, HashGrid L={a.levels} F={a.features} "
                               f"T=2^{a.log2_hashmap_size} base 16 per_level_scale {pls:.4f} + {a.hidden_layers}x64 FullyFusedMLP, "
                               f"{a.fb}x{a.fb} rendering mode {a.mode} ({'sample streaming' if a.mode == 5 else 'sample streaming with gradient shading' if a.mode == 8 else 'sample streaming, single-shade heuristic: camera pass + shadow pass' if a.mode == 11 else 'path tracing: one launch with the network inside the tracking loop (VNR_AMD_IN_SHADER=0: sample streaming)' if a.mode in (14, 15) else 'in-shader ray marching mode: the streaming path unless VNR_AMD_IN_SHADER=1'}), sampling rate 1, N_ITERS {os.environ.get('VNR_RM_N_ITERS', '24 (32 when a rank renders at most 262144 pixels)')}; {cache_text}",
                   "volume": f"{a.size}^3", "framebuffer": f"{a.fb}x{a.fb}", "n_params": info["n_params"],
                   "tfn": f"256-entry ramp-with-bumps, seed 7, opacity scale {a.opacity_scale}",
                   "camera": cam, "train_steps": a.train_steps, "batch": 65536,
                   "parallelism": (f"image tiles x{ctx.world} (interleaved 8-scanline tile rows) + one in-place all-gather per frame through libvnr_amd "
                                   f"(transport {ctx.transport}); training data parallel, fp16 gradient exchange") if ctx.distributed else "single GPU"},
        "inference_cache": {"what": "brick image: de-hashed copy of the hashed levels, built once the parameters have been left unchanged for 24 launches, "
                                    "dropped by every optimizer step; results are bit-identical with and without it",
                            "in_use": bool(brick_state["in_use"]), "brick_image_bytes": int(brick_state["bytes"]), "model_bytes": int(info["n_params"]) * 2,
                            "ratio": round(brick_state["bytes"] / (info["n_params"] * 2.0), 1), "build_ms": round(brick_state["build_ms"], 2),
                            "levels_in_image": [l for l in range(32) if brick_state.get("levels", 0) >> l & 1],
                            "policy": "budget = 1/16 of the device's memory (18 GB on MI355X), at most a quarter of the free memory; VNR_AMD_BRICK_MAX_GB / "
                                      "vnrAmdNeuralVolumeSetBrickImageBudget set it; levels are taken finest first",
                            "budget_table": budget_rows},
        "mlp_msamples_per_s": round(samples_all / elapsed / 1e6, 1),
        "mlp_msamples_per_s_kernel_only": round(samples_evt / (infer_ms * 1e-3) / 1e6, 1) if infer_ms > 0 else None,
        "samples_per_frame": int(samples_all / a.steps), "samples_per_hit_ray": round(samples_all / a.steps / max(rays_hit, 1), 1),
        "network_evaluations_per_shaded_sample": evals_per_sample, "shaded_samples_per_frame": shaded_samples_per_frame,
        "reference_slots_per_frame": int(slots_all / a.steps), "iterations_per_frame": iters, "rays_hit": rays_hit,
        "psnr_db": None if psnr is None else round(psnr, 2), "image_vs_ground_truth": image, "train_ms_per_step": round(train_ms, 3), "train_loss": round(train_loss, 5),
        "setup_s": round(setup_s, 1),
        "library_build": L.vnrAmdBuildId().decode(),   # md5 (12 digits) of the sources libvnr_amd.so was linked from (csrc/Makefile)
        "roofline": roofline,
        "train_roofline": train_roofline,
    }
    if interactive is not None:
        out["interactive"] = interactive
    if c5 is not None:
        out["c5"] = c5
    if ctx.distributed:
        # did RCCL see N ranks?  Two keys beside n_gpus answer it
        out["transport"] = ("rccl" if ctx.transport == "rccl" else
                            f"{ctx.transport}-fallback (the RCCL probe failed: transport_probe)" if transport_probe is not None else f"{ctx.transport} (chosen by VNR_AMD_DIST_TRANSPORT" + (": " + os.environ["VNR_BENCH_TRANSPORT_REASON"] if os.environ.get("VNR_BENCH_TRANSPORT_REASON") else "") + ")")
        out["rccl_ranks_seen"] = int(L.vnrAmdDistRcclRanksSeen())
        out["launcher"] = ("bench.py itself (bare `--gpus N`: a parent that never touches the GPU started one fresh process per rank)"
                           if os.environ.get("VNR_BENCH_LAUNCHED_BY") == "bench.py" else "external (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* came with the environment)")
    if per_rank is not None:
        out["per_rank"] = per_rank
        out["collective_self_test"] = self_test
        if transport_probe is not None:
            out["transport_probe"] = transport_probe
    if ctx.world == 1 and not a.no_cpu_baseline:
        mc = api.volume_macrocell(nv)
        out["cpu_baseline"] = cpu_baseline(sv, nv, info, dims, (colors, alphas), cam, a.fb, mc, pls, a.hidden_layers, a.log2_hashmap_size)
    print(json.dumps(out), flush=True)
    dist.barrier(ctx)


if __name__ == "__main__":
    try:
        main()
    finally:
        # every rank releases the communicator explicitly (ncclCommDestroy, the communication stream) while librccl and the HIP runtime
        # are still alive; nothing is left for exit-time destructors (csrc/dist.cpp ~Dist leaks an RCCL communicator rather than touch it)
        try:
            dist.finalize()
        except Exception as e:   # never turn a printed bench line into a failed run
            print(f"warning: vnrAmdDistFinalize: {e}", file=sys.stderr)
