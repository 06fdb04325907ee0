#!/usr/bin/env python3
"""vnr_cmd_isosurface (apps/batch_isosurface.cpp:22-80): marching cubes of a simple volume (a scene document) or of a neural volume
(a params.json) at an isovalue; writes ./isosurface.obj like the reference.

  python tools/vnr_cmd_isosurface.py --simple-volume scene.json --iso 0.4
  python tools/vnr_cmd_isosurface.py --neural-volume params.json --isovalue 0.4

Exactly one of --simple-volume / --neural-volume (the reference's Xor group), --iso / --isovalue required."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api  # noqa: E402


def main():
    p = argparse.ArgumentParser(description="Commandline Volume Renderer")
    g = p.add_mutually_exclusive_group(required=True)
    g.add_argument("--simple-volume", metavar="filename", help="the simple volume to render")
    g.add_argument("--neural-volume", metavar="filename", help="the neural volume to render")
    p.add_argument("--iso", "--isovalue", dest="iso", type=float, required=True, metavar="float", help="iso-value")
    a = p.parse_args()
    volume = api.vnrCreateSimpleVolume(a.simple_volume, "GPU", False) if a.simple_volume else api.vnrCreateNeuralVolume(a.neural_volume)
    t = time.perf_counter()
    verts = api.vnrMarchingCube(volume, a.iso)
    print(f"Marching Cube Time = {time.perf_counter() - t:.6f}s")    # core/marching_cube.cu:431
    api.vnrSaveTriangles("isosurface.obj", verts)
    print(f"[info] {verts.shape[0] // 3} triangles -> isosurface.obj")


if __name__ == "__main__":
    main()
