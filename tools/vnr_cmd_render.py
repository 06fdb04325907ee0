#!/usr/bin/env python3
"""vnr_cmd_render with the reference's command line (apps/batch_renderer.cpp:62-239) on top of libvnr_amd:

  --simple-volume <scene.json> | --neural-volume <params.json>   exactly one of the two (a scene document / a BSON parameter file)
  --tfn <scene.json>         the transfer function preset: camera (view.camera) and value range come from it       [required]
  --num-frames <int>         number of frames to render                                                            [required]
  --sampling-rate <float>    ray marching sampling rate                                                            [1]
  --density-scale <float>    path tracing density scale                                                            [1]
  --rendering-mode <int>     the vnrRenderMode enum (api.h:35-58): 4-12 ray marching, 13-15 path tracing                [0]
  --exp <name>               experiment name: <name>.csv (#, frame time, fps) and <name>-screenshot.*              [output]
  --camera-from / --camera-at / --camera-up   accepted and, like in the reference (batch_renderer.cpp:193), unused

Like the reference: 768 x 768 frame buffer, transfer function value range (0, 1), no denoiser, 5 warm-up frames, then --num-frames
timed calls of vnrRender, per-frame times in the log, "fps = num_frames / total time", the Summary block.

More than one GPU (new: the reference is single-GPU): one process per GPU with the torchrun environment (RANK, LOCAL_RANK, WORLD_SIZE,
MASTER_ADDR, MASTER_PORT); the frame is then rendered as interleaved tile rows, one share per rank, all-gathered over RCCL and
pipelined (vnrAmdRendererRenderPipelined); rank 0 writes the log, the screenshot and the Summary.

Two differences, both stated at run time:
  * the table of the --tfn preset is decoded in the reference by OVR's tfn module (tfn::loadTransferFunction), which is not part of
    the reference tree: --tfn-table <file> supplies what that module yields (resolution x RGBA; .npy, or JSON [[r,g,b,a], ...]);
    without it the tool stops with an explanation instead of rendering with an invented table;
  * the screenshot is a PNG (zlib is in the standard library; the reference writes a JPEG through stb_image_write), flipped
    vertically and quantised as saveJPG does (uint32(255.99 * clamp(v, 0, 1)))."""
import argparse
import json
import os
import struct
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, dist  # noqa: E402

# the vnrRenderMode enum (api.h:35-58), in the order of the reference tool's help text (batch_renderer.cpp:45-60)
RENDER_MODES = ["0-3 OptiX reference marcher (not implemented here)", "4 ray marching, decoding", "5 ray marching, sample streaming",
                "6 ray marching, in shader", "7 / 8 / 9 the same three with local illumination (gradient shading)",
                "10 / 11 / 12 the same three with the single-shade heuristic", "13 / 14 / 15 path tracing: decoding, sample streaming, in shader"]


def save_png(fname, pixels):
    """pixels [h, w, 4] float -> 8-bit RGBA PNG, first scanline at the bottom like stbi_flip_vertically_on_write(1)"""
    img = (np.float32(255.99) * np.clip(pixels, 0.0, 1.0)).astype(np.uint32).astype(np.uint8)[::-1]
    h, w = img.shape[:2]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(fname, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6))
                + chunk(b"IEND", b""))


def load_table(path):
    if path.endswith(".npy"):
        return np.load(path)
    with open(path) as f:
        return np.asarray(json.load(f), dtype=np.float32)


def main(argv=None):
    p = argparse.ArgumentParser(description="Commandline Volume Renderer")
    g = p.add_mutually_exclusive_group(required=True)
    g.add_argument("--simple-volume", metavar="filename", help="the simple volume to render")
    g.add_argument("--neural-volume", metavar="filename", help="the neural volume to render")
    for name, text in (("from", "from where we are looking"), ("at", "which point we are looking at"), ("up", "general up-vector")):
        p.add_argument(f"--camera-{name}", nargs=3, type=float, metavar="f", help=text)
    p.add_argument("--tfn", required=True, metavar="filename", help="the transfer function preset")
    p.add_argument("--tfn-table", default="", metavar="filename", help="the decoded table of the preset (resolution x RGBA)")
    p.add_argument("--num-frames", required=True, type=int, metavar="int", help="number of frames to render")
    p.add_argument("--sampling-rate", type=float, default=1.0, metavar="float", help="ray marching sampling rate")
    p.add_argument("--density-scale", type=float, default=1.0, metavar="float", help="path tracing density scale")
    p.add_argument("--rendering-mode", type=int, default=0, metavar="int", help="\n".join(RENDER_MODES))
    p.add_argument("--exp", default="output", metavar="std::string", help="experiment name")
    a = p.parse_args(argv)
    given = [x is not None for x in (a.camera_from, a.camera_at, a.camera_up)]
    if any(given) and not all(given):      # args::Group::Validators::AllOrNone
        p.error("--camera-from, --camera-at and --camera-up: all or none")

    ctx = dist.init_from_env()        # one rank: binds the GPU; more: meets the other ranks (RCCL)
    if a.simple_volume:
        volume = api.vnrCreateSimpleVolume(a.simple_volume, "GPU", False)
    else:
        volume = api.vnrCreateNeuralVolume(a.neural_volume)             # vnrLoadJsonBinary + vnrCreateNeuralVolume(params)

    camera = api.vnrCreateCamera()
    api.vnrCameraSet(camera, a.tfn)
    cam_from, cam_at, cam_up = api.vnrCameraGetPosition(camera), api.vnrCameraGetFocus(camera), api.vnrCameraGetUpVec(camera)

    tfn = api.vnrCreateTransferFunction(a.tfn, table=load_table(a.tfn_table) if a.tfn_table else None)
    api.vnrTransferFunctionSetValueRange(tfn, (0.0, 1.0))

    size = (768, 768)
    ren = api.vnrCreateRenderer(volume)
    api.vnrRendererSetTransferFunction(ren, tfn)
    api.vnrRendererSetCamera(ren, camera)
    api.vnrRendererSetFramebufferSize(ren, size)
    api.vnrRendererSetMode(ren, a.rendering_mode)
    api.vnrRendererSetDenoiser(ren, False)
    api.vnrRendererSetVolumeDensityScale(ren, a.density_scale)
    api.vnrRendererSetVolumeSamplingRate(ren, a.sampling_rate)

    timings = np.zeros(max(a.num_frames, 0))
    if not ctx.distributed:
        for _ in range(5):   # warm up
            api.vnrRender(ren)
        t_all = time.perf_counter()
        for i in range(a.num_frames):
            t0 = time.perf_counter()
            api.vnrRender(ren)
            timings[i] = (time.perf_counter() - t0) * 1e3
        total_s = time.perf_counter() - t_all
        pixels = api.vnrRendererMapFrame(ren)
    else:
        sr = dist.ShardedRenderer(ctx, ren, size[0], size[1])
        for _ in range(5):
            sr.render()
        sr.flush()
        dist.barrier(ctx)
        t_all = time.perf_counter()
        for i in range(a.num_frames):   # call i hands out frame i - 1, gathered; the flush hands out the last one
            t0 = time.perf_counter()
            sr.render()
            timings[i] = (time.perf_counter() - t0) * 1e3
        last = sr.flush()
        dist.barrier(ctx)
        total_s = time.perf_counter() - t_all
        pixels = sr.download(last)
        if ctx.rank != 0:
            dist.finalize()
            return 0

    with open(a.exp + ".csv", "w") as log:   # Logger: {"#", "frame time", "fps"}
        log.write("#,frame time,fps\n")
        for i, ms in enumerate(timings):
            log.write(f"{float(i)},{ms / 1000.0},{1000.0 / ms}\n")

    save_png(a.exp + "-screenshot.png", np.asarray(pixels).reshape(size[1], size[0], 4))

    vec = lambda v: "(" + ",".join(f"{x:g}" for x in v) + ")"
    print(f"Summary: {a.exp}")
    print(f"\tvolume: {a.simple_volume or a.neural_volume}")
    print(f"\t   tfn: {a.tfn}")
    print(f"\t   fps: {a.num_frames / total_s if total_s > 0 else 0.0}")
    print(f"\tdensity scale: {a.density_scale}")
    print(f"\tsampling rate: {a.sampling_rate}")
    print(f"\tcamera: {vec(cam_from)}")
    print(f"\t        {vec(cam_at)}")
    print(f"\t        {vec(cam_up)}")
    if ctx.distributed:
        print(f"\t  gpus: {ctx.world}")
        dist.finalize()
    return 0


if __name__ == "__main__":
    sys.exit(main())
