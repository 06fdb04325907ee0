#!/usr/bin/env python3
"""vnr_view_model with the reference's command line (apps/view_model.cpp:39-160): prints what a params.json (BSON) holds and, with
--correct --dims x,y,z, adds the missing volume dims and writes params-corrected.json; --groundtruth <scene.json> also loads
the model next to the ground truth and prints PSNR / SSIM (needs a GPU).

  view_model.py <params.json> [--dims x,y,z] [--correct] [--groundtruth scene.json]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pretty_bytes(n):
    for unit, f in (("GB", 1 << 30), ("MB", 1 << 20), ("KB", 1 << 10)):
        if n >= f:
            return f"{n / f:.2f} {unit}"
    return f"{n} B"


def main(argv=None):
    import bson   # pymongo's codec; the library's own codec is byte-compatible (tests/test_cabi.py)
    p = argparse.ArgumentParser(description="Model Viewer")
    p.add_argument("volume", metavar="filename", help="the neural volume")
    p.add_argument("--groundtruth", metavar="filename", help="the ground truth volume")
    p.add_argument("--dims", metavar="vec3i", help="volume dimension: x,y,z")
    p.add_argument("--correct", action="store_true", help="correct model")
    a = p.parse_args(argv)
    root = bson.decode(open(a.volume, "rb").read())
    if "volume" in root:
        d = root["volume"]["dims"]
        print(f"[info] volume dims: ({d['x']}, {d['y']}, {d['z']})")
    else:
        print("[info] this file does not contain dimension data.")
        if a.correct and a.dims:
            x, y, z = (int(float(v)) for v in a.dims.replace(" ", ",").split(",") if v)
            root["volume"] = {"dims": {"x": x, "y": y, "z": z}}
    if "macrocell" in root:
        m = root["macrocell"]
        print(f"[info] use GT macrocell = {int(bool(m.get('groundtruth', False)))}")
        print(f"[info] macrocell dims = ({m['dims']['x']}, {m['dims']['y']}, {m['dims']['z']})")
        print(f"[info] macrocell spacing = ({m['spacings']['x']}, {m['spacings']['y']}, {m['spacings']['z']})")
        print(f"[info] macrocell data = {pretty_bytes(len(m['data']))}")
    else:
        print("[info] this file does not contain macrocell data.")
    if "model" in root:
        import json
        print("[info] model: " + json.dumps(root["model"], indent=2, sort_keys=True))
    else:
        print("[info] this file does not contain model information.")
    if "parameters" in root:
        print(f"[info] params = {pretty_bytes(len(root['parameters']['params_binary']))}")
    else:
        print("[info] this file does not contain model weights?!")
    if a.correct:
        open("params-corrected.json", "wb").write(bson.encode(root))
        print(f"Corrected model '{a.volume}' and saved it as 'params-corrected.json'.")
    if a.groundtruth:
        from instantvnr_amd import api
        api.check(api.lib().vnrAmdInit(-1))
        gt = api.vnrCreateSimpleVolume(a.groundtruth, "GPU", False)
        nv = api.vnrCreateNeuralVolume(root["model"], gt, True)
        api.vnrNeuralVolumeSetParams(nv, a.volume)
        print(f"[info] PSNR = {api.vnrNeuralVolumeGetPSNR(nv, True)}")
        print(f"[info] SSIM = {api.vnrNeuralVolumeGetSSIM(nv, True)}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
