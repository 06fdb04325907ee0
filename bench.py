#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native instantvnr hot path.

Metric (BASELINE.json): fps at 1024^2 on a 1024^3 volume + MLP Msamples/s; PSNR vs ground truth.
One "step" = one frame: sample-streaming ray march (rendering mode 5) of the trained neural volume
(HashGrid L=16 F=2 T=2^22 + 3x64 FullyFusedMLP) at 1024x1024; with N GPUs the image is sharded by interleaved
scanline blocks and gathered with one RCCL all_gather per frame (strong scaling: total work is fixed).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Untimed setup: generate the synthetic 1024^3 Perlin volume on the GPU, train the model (data parallel with an RCCL
gradient all-reduce when N > 1), build the renderer.  Rank 0 prints ONE JSON line.
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
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--size", type=int, default=1024, help="volume edge (C4: 1024)")
    p.add_argument("--fb", type=int, default=1024, help="framebuffer edge (C4: 1024)")
    p.add_argument("--levels", type=int, default=16)
    p.add_argument("--features", type=int, default=2)
    p.add_argument("--log2-hashmap-size", type=int, default=22)
    p.add_argument("--hidden-layers", type=int, default=3)
    p.add_argument("--per-level-scale", type=float, default=0.0, help="0 = finest level resolution equals the volume edge")
    p.add_argument("--train-steps", type=int, default=1500)
    p.add_argument("--opacity-scale", type=float, default=0.06)
    p.add_argument("--camera-distance", type=float, default=1.1, help="camera distance in volume edges (oblique view)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-psnr", action="store_true")
    p.add_argument("--no-alone", action="store_true", help="skip the un-timed one-stream leg (roofline.alone)")
    p.add_argument("--no-kernel-events", action="store_true", help="diagnostics: no HIP events around the evaluation kernel (roofline.achieved reads 0)")
    p.add_argument("--mode", type=int, default=5, choices=(5, 8, 11, 14),
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


def main():
    a = parse()
    if os.environ.get("VNR_BENCH_DUMP_AFTER"):  # diagnostics: where does a run that hangs under the profiler stand?
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VNR_BENCH_DUMP_AFTER"]), exit=False, file=sys.stderr)
    ctx = dist.init_from_env()
    if a.gpus != ctx.world:
        if ctx.rank == 0:
            print(f"warning: --gpus {a.gpus} but WORLD_SIZE={ctx.world}; using {ctx.world}", file=sys.stderr)
    L = lib()
    dims = (a.size, a.size, a.size)
    pls = a.per_level_scale if a.per_level_scale > 0 else float(np.exp(np.log(a.size / 16.0) / max(a.levels - 1, 1)))
    os.environ.setdefault("VNR_AMD_INIT_SEED", "20240611")  # identical initial parameters on every rank

    # ---- setup (untimed) -------------------------------------------------------------------------------------
    t_setup = time.perf_counter()
    sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
    cfg = syn.model_config(n_levels=a.levels, n_features=a.features, log2_hashmap_size=a.log2_hashmap_size, base_resolution=16,
                           n_hidden_layers=a.hidden_layers, per_level_scale=pls)
    # ground-truth macrocell (identical on every rank, so tiles compose exactly)
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
    check(L.vnrAmdNeuralVolumeSetSamplerSeed(nv.h, 1337, 0xda3e39cb94b95bdb + ctx.rank))
    info = api.neural_info(nv)
    check(L.vnrAmdSynchronize())
    t_train = time.perf_counter()
    dist.train_data_parallel(ctx, nv, a.train_steps, fast_mode=True)
    check(L.vnrAmdSynchronize())
    train_ms = (time.perf_counter() - t_train) * 1e3 / max(a.train_steps, 1)
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
    ren = api.vnrCreateRenderer(nv)
    api.vnrRendererSetTransferFunction(ren, tfn)
    api.vnrRendererSetCamera(ren, camera)
    api.vnrRendererSetFramebufferSize(ren, (a.fb, a.fb))
    api.vnrRendererSetMode(ren, a.mode)
    api.vnrRendererSetProfiling(ren, not a.no_kernel_events)  # HIP events around the fused encode+MLP kernel, on its own stream
    sr = dist.ShardedRenderer(ctx, ren, a.fb, a.fb)
    setup_s = time.perf_counter() - t_setup

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
    for _ in range(a.steps):
        sr.render()
        st = api.vnrRendererGetFrameStats(ren)
        samples += st["n_samples"]; slots += st["n_reference_slots"]; infer_ms += st["infer_kernel_ms"]; union_ms += st["infer_union_ms"]
        launches += st["infer_kernel_launches"]; iters = st["n_iterations"]
    sr.flush()   # N > 1: the gather of the last frame (ShardedRenderer pipelines render k with gather k - 1); every frame is rendered AND gathered inside the timed region
    dist.barrier(ctx)
    elapsed = time.perf_counter() - t0
    rays_hit = st["n_rays_hit"]

    # ---- un-timed: the neural frame against the frame of the ground-truth volume (same camera / TFN / mode / macrocell) --------
    image = None
    if ctx.world == 1 and not a.no_psnr:
        def one_frame(volume):
            rr = api.vnrCreateRenderer(volume)
            api.vnrRendererSetTransferFunction(rr, tfn)
            api.vnrRendererSetCamera(rr, camera)
            api.vnrRendererSetFramebufferSize(rr, (a.fb, a.fb))
            api.vnrRendererSetMode(rr, a.mode)
            api.vnrRender(rr)
            return api.vnrRendererMapFrame(rr).astype(np.float64).copy()

        f_nn, f_gt = one_frame(nv), one_frame(sv)
        d = f_nn - f_gt
        mse = float((d[..., :3] ** 2).mean())
        image = {"what": "first frame of the neural volume vs the same frame marched through the ground-truth volume (rgb, peak 1)",
                 "psnr_db": round(10.0 * np.log10(1.0 / mse), 2) if mse > 0 else None,
                 "l2_per_pixel_mean": round(float(np.sqrt((d ** 2).sum(axis=2)).mean()), 6),
                 "l2_per_pixel_max": round(float(np.sqrt((d ** 2).sum(axis=2)).max()), 5)}

    # ---- un-timed extra leg (one GPU only): the dominant kernel with the GPU to itself -------------------------------------
    # The default renderer runs two ray halves on two HIP streams, so a launch of the fused kernel shares the GPU with the
    # other half's kernels and its HIP-event duration says little about the kernel.  The same frames on ONE stream (nothing
    # else resident while the kernel runs) give the per-launch figure the roofline fraction is meant to be.
    alone = None
    if ctx.world == 1 and not a.no_alone and os.environ.get("VNR_AMD_RENDER_HALVES", "2") != "1":
        os.environ["VNR_AMD_RENDER_HALVES"] = "1"
        ren1 = api.vnrCreateRenderer(nv)
        del os.environ["VNR_AMD_RENDER_HALVES"]
        api.vnrRendererSetTransferFunction(ren1, tfn)
        api.vnrRendererSetCamera(ren1, camera)
        api.vnrRendererSetFramebufferSize(ren1, (a.fb, a.fb))
        api.vnrRendererSetMode(ren1, a.mode)
        api.vnrRendererSetOutputAsDeviceFramebuffer(ren1, True)
        api.vnrRendererSetProfiling(ren1, True)
        for _ in range(3):
            api.vnrRender(ren1); api.vnrRendererMapFrame(ren1)
        check(L.vnrAmdSynchronize())
        t1 = time.perf_counter()
        a_samples = a_launches = 0
        a_ms = 0.0
        n_alone = max(5, a.steps // 2)
        for _ in range(n_alone):
            api.vnrRender(ren1); api.vnrRendererMapFrame(ren1)
            s1 = api.vnrRendererGetFrameStats(ren1)
            a_samples += s1["n_samples"]; a_ms += s1["infer_kernel_ms"]; a_launches += s1["infer_kernel_launches"]
        check(L.vnrAmdSynchronize())
        alone = {"frames": n_alone, "fps": round(n_alone / (time.perf_counter() - t1), 2), "samples": a_samples, "ms": a_ms, "launches": a_launches}
        del ren1

    if ctx.distributed:
        import torch
        import torch.distributed as td
        mx = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        td.all_reduce(mx, op=td.ReduceOp.MAX)          # MAX over ranks of the timed region
        elapsed = float(mx[0])
        sm = torch.tensor([float(samples), float(slots), float(rays_hit)], dtype=torch.float64, device="cuda")
        td.all_reduce(sm, op=td.ReduceOp.SUM)
        samples_all, slots_all, rays_hit = float(sm[0]), float(sm[1]), int(sm[2])
    else:
        samples_all, slots_all = float(samples), float(slots)

    if ctx.rank != 0:
        return
    fps = a.steps / elapsed
    # the renderer counts SHADED samples; with gradient shading (mode 8) the network evaluates 4 coordinates for each of them,
    # and every rate below is per network evaluation
    evals_per_sample = 4 if a.mode == 8 else 1
    shaded_samples_per_frame = int(samples_all / a.steps)
    samples *= evals_per_sample
    samples_all *= evals_per_sample
    bytes_per_sample = 12 + info["n_levels"] * 8 * info["n_features_per_level"] * 2 + 4
    in_pad = info["padded_width"]
    flops_per_sample = 2 * (in_pad * 64 + (info["n_hidden_layers"] - 1) * 64 * 64 + 64)
    # dominant kernel: fused hash-grid gather + MLP.  achieved = algorithmic bytes of the samples this rank's
    # launches processed / summed launch durations (HIP events on the render stream), i.e. per-launch average.
    achieved = (samples * bytes_per_sample) / (infer_ms * 1e-3) / 1e9 if infer_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": "fused_infer_kernel (hash-grid gather + 3x64 MLP on MFMA)",
                "algorithmic_bytes_per_sample": bytes_per_sample, "flops_per_sample": flops_per_sample,
                "avg_launch_ms": round(infer_ms / max(launches, 1), 4), "launches": launches,
                "mfma_tflops": round(samples * flops_per_sample / (infer_ms * 1e-3) / 1e12, 2) if infer_ms > 0 else 0.0}
    # The renderer runs the rays as 2 halves on 2 streams by default (march of one half overlaps inference of the other,
    # and the two halves' inference kernels overlap each other), so a launch's HIP-event duration includes time it shares
    # the GPU: `achieved`/`frac` (defined per launch) drop although the frame gets faster.  The frame-level figure below
    # does not depend on scheduling: algorithmic bytes of all live samples of a frame / frame time.
    halves = 1 if os.environ.get("VNR_AMD_RENDER_HALVES", "2") == "1" else 2
    if union_ms > 0:
        u = (samples * bytes_per_sample) / (union_ms * 1e-3) / 1e9
        roofline["union"] = {"what": "timed region: algorithmic bytes of all launches / the time during which at least one launch of the kernel was "
                                     "running (union of the HIP-event intervals of both streams): overlapping launches count once",
                             "achieved": round(u, 1), "frac": round(u / HBM_PEAK_GBS, 4), "ms_per_frame": round(union_ms / a.steps, 4)}
    if alone and alone["ms"] > 0:
        ev = alone["samples"] * evals_per_sample
        a_gbs = ev * bytes_per_sample / (alone["ms"] * 1e-3) / 1e9
        roofline["alone"] = {"what": "same frames on ONE HIP stream, un-timed extra leg of this run: nothing else is resident while the kernel runs",
                             "achieved": round(a_gbs, 1), "frac": round(a_gbs / HBM_PEAK_GBS, 4),
                             "avg_launch_ms": round(alone["ms"] / max(alone["launches"], 1), 4), "launches": alone["launches"],
                             "msamples_per_s": round(ev / (alone["ms"] * 1e-3) / 1e6, 1), "frames": alone["frames"],
                             "fps_one_stream": alone["fps"],
                             "mfma_tflops": round(ev * flops_per_sample / (alone["ms"] * 1e-3) / 1e12, 2)}
    roofline["concurrency"] = (f"{halves} ray halves on {halves} HIP streams; launch durations are per stream and overlap"
                               if halves == 2 else "1 stream: launches run alone")
    frame_gbs = (samples / a.steps) * bytes_per_sample / (elapsed / a.steps) / 1e9
    roofline["frame_algorithmic_gbs"] = round(frame_gbs, 1)
    roofline["frame_frac"] = round(frame_gbs / HBM_PEAK_GBS, 4)
    # HBM-side traffic of the dominant kernel: PMC counters cannot be collected inside this run (separate rocprofv3
    # --pmc passes, tools/run_pmc.sh); a committed result applies only to the exact default workload on one GPU with
    # the same stream configuration it was measured with.
    default_workload = (a.size, a.fb, a.levels, a.features, a.log2_hashmap_size, a.hidden_layers, a.per_level_scale,
                        a.train_steps, a.opacity_scale, a.camera_distance, a.mode) == (1024, 1024, 16, 2, 22, 3, 0.0, 1500, 0.06, 1.1, 5)
    pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_l_pmc_traffic.json")
    brick = api.neural_brick_image(nv)["in_use"]
    if default_workload and ctx.world == 1 and brick and os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
        leg = pmc["one_stream" if halves == 1 else "two_streams"]
        roofline["traffic"] = round(leg["traffic_per_launch"])
        roofline["traffic_unit"] = "bytes per launch (L2<->fabric reads x2-corrected + writes; includes Infinity-Cache hits)"
        roofline["algorithmic_bytes_per_launch"] = round(samples * bytes_per_sample / max(launches, 1))
        roofline["traffic_note"] = ("measured in separate rocprofv3 --pmc passes of this command, not in this run: profiles/r01_l_pmc_traffic.json "
                                    "(traffic/algorithmic = %.2f; 2.09 before the brick image, profiles/r01_pmc_traffic.json)" % leg["traffic_over_algorithmic"])
        if "alone" in roofline:
            t1 = pmc["one_stream"].get("traffic_per_launch")
            roofline["alone"]["traffic"] = round(t1) if t1 else None
    else:
        roofline["traffic_note"] = "null: the committed PMC passes (profiles/r01_l_pmc_traffic.json) describe the default workload on one GPU with the brick image"
    out = {
        "metric": "fps at 1024^2 on 1024^3 volume" if (a.size, a.fb, a.mode) == (1024, 1024, 5) else f"fps at {a.fb}^2 on {a.size}^3 volume, rendering mode {a.mode}" if a.mode != 5 else f"fps at {a.fb}^2 on {a.size}^3 volume",
        "value": round(fps, 3), "unit": "frames/s", "n_gpus": ctx.world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"C4: {a.size}^3 synthetic Perlin fBm volume (seed 42), HashGrid L={a.levels} F={a.features} "
                               f"T=2^{a.log2_hashmap_size} base 16 per_level_scale {pls:.4f} + {a.hidden_layers}x64 FullyFusedMLP, "
                               f"{a.fb}x{a.fb} rendering mode {a.mode} ({'sample streaming' if a.mode == 5 else 'sample streaming with gradient shading' if a.mode == 8 else 'sample streaming, single-shade heuristic: camera pass + shadow pass' if a.mode == 11 else 'path tracing, sample streaming'}), sampling rate 1, N_ITERS {os.environ.get('VNR_RM_N_ITERS', '24 (32 when a rank renders at most 196608 pixels)')}",
                   "volume": f"{a.size}^3", "framebuffer": f"{a.fb}x{a.fb}", "n_params": info["n_params"],
                   "tfn": f"256-entry ramp-with-bumps, seed 7, opacity scale {a.opacity_scale}",
                   "camera": cam, "train_steps": a.train_steps, "batch": 65536,
                   "brick_image": {k: (round(v / 2**30, 2) if k == "bytes" else round(v, 2) if k == "build_ms" else v)
                                   for k, v in api.neural_brick_image(nv).items()} | {"unit": "bytes in GiB; built once, in the warm-up"},
                   "parallelism": f"image tiles x{ctx.world} (interleaved 8-scanline blocks) + RCCL all_gather" if ctx.world > 1 else "single GPU"},
        "mlp_msamples_per_s": round(samples_all / elapsed / 1e6, 1),
        "mlp_msamples_per_s_kernel_only": round(samples / (infer_ms * 1e-3) / 1e6, 1) if infer_ms > 0 else None,
        "samples_per_frame": int(samples_all / a.steps), "samples_per_hit_ray": round(samples_all / a.steps / max(rays_hit, 1), 1),
        "network_evaluations_per_shaded_sample": evals_per_sample, "shaded_samples_per_frame": shaded_samples_per_frame,
        "reference_slots_per_frame": int(slots_all / a.steps), "iterations_per_frame": iters, "rays_hit": rays_hit,
        "psnr_db": None if psnr is None else round(psnr, 2), "image_vs_ground_truth": image, "train_ms_per_step": round(train_ms, 3), "train_loss": round(train_loss, 5),
        "setup_s": round(setup_s, 1),
        "roofline": roofline,
    }
    if ctx.world == 1 and not a.no_cpu_baseline:
        mc = api.volume_macrocell(nv)
        out["cpu_baseline"] = cpu_baseline(sv, nv, info, dims, (colors, alphas), cam, a.fb, mc, pls, a.hidden_layers, a.log2_hashmap_size)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
