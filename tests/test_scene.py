"""Scene documents (serializer.cpp:137-477): vnrCreateSimpleVolume(scene, mode), vnrCreateCamera(scene) / vnrCameraSet(self, scene)
and the value range of a scene's transfer function.  The expected values are worked out by hand from the reference's loader;
the host-only parts run without a GPU."""
import json
import os

import numpy as np
import pytest

from instantvnr_amd import api


def pos(cam):
    return tuple(float(v) for v in api.vnrCameraGetPosition(cam))


def vidi_scene(files, dims, type_name, volume_extra=None, version=None, **source_extra):
    src = [dict({"format": "REGULAR_GRID_RAW_BINARY", "fileName": f, "dimensions": {"x": dims[0], "y": dims[1], "z": dims[2]},
                 "type": type_name}, **source_extra) for f in files]
    sc = {"dataSource": src,
          "view": {"camera": {"eye": {"x": 10.0, "y": 20.0, "z": -300.0}, "center": {"x": 16.0, "y": 8.0, "z": 4.0},
                              "up": {"x": 0.0, "y": 1.0, "z": 0.0}, "fovy": 42.0},
                   "volume": dict({"transferFunction": {}}, **(volume_extra or {}))}}
    if version:
        sc["version"] = version
    return sc


def test_camera_from_scene_and_errors():
    sc = vidi_scene(["a.raw"], (32, 16, 8), "FLOAT", version="VIDI3D")
    cam = api.vnrCreateCamera(sc)
    # eye and center move by -dims / 2 (serializer.cpp:425-427); up and fovy are taken as they are
    assert pos(cam) == (10.0 - 16.0, 20.0 - 8.0, -300.0 - 4.0)
    assert tuple(api.vnrCameraGetFocus(cam)) == (0.0, 0.0, 0.0)
    assert tuple(api.vnrCameraGetUpVec(cam)) == (0.0, 1.0, 0.0)
    cam2 = api.vnrCreateCamera()
    api.vnrCameraSet(cam2, json.dumps(sc))          # JSON text
    assert pos(cam2) == pos(cam)
    # a DIVA scene leaves the camera untouched (the reference's TODO, serializer.cpp:464)
    cam3 = api.vnrCreateCamera()
    api.vnrCameraSet(cam3, (1, 2, 3), (0, 0, 0), (0, 0, 1))
    api.vnrCameraSet(cam3, {"version": "DIVA", "volume": {}})
    assert pos(cam3) == (1.0, 2.0, 3.0)
    with pytest.raises(api.VnrAmdError, match="unknown JSON configuration format"):
        api.vnrCreateCamera(dict(sc, version="OTHER"))
    with pytest.raises(api.VnrAmdError, match="expected to be an array"):
        api.vnrCreateCamera(dict(sc, dataSource={}))
    with pytest.raises(api.VnrAmdError):
        api.vnrCreateCamera({"view": {}})


def test_camera_from_scene_file(tmp_path):
    sc = vidi_scene(["a.raw"], (2, 4, 6), "BYTE")
    p = tmp_path / "scene.json"
    p.write_text("// a scene file may carry comments\n" + json.dumps(sc))
    cam = api.vnrCreateCamera(str(p))               # a string that is not JSON is a path (api.cpp:77-83)
    assert pos(cam) == (9.0, 18.0, -303.0)


@pytest.mark.parametrize("type_name,unnorm,expect", [
    ("UNSIGNED_BYTE", False, (255 * 0.25, 255 * 0.5)), ("BYTE", False, (127 * 0.25, 127 * 0.5)),
    ("UNSIGNED_SHORT", False, (65535 * 0.25, 65535 * 0.5)), ("SHORT", False, (32767 * 0.25, 32767 * 0.5)),
    ("UNSIGNED_INT", False, (float(np.float32(4294967295) * np.float32(0.25)), float(np.float32(4294967295) * np.float32(0.5)))),
    ("INT", False, (float(np.float32(2147483647) * np.float32(0.25)), float(np.float32(2147483647) * np.float32(0.5)))),
    ("FLOAT", False, (0.25, 0.5)), ("DOUBLE", False, (0.25, 0.5)), ("UNSIGNED_BYTE", True, (3.0, 200.0))])
def test_scene_value_range(type_name, unnorm, expect):
    extra = {"scalarMappingRange": {"minimum": 0.25, "maximum": 0.5}}
    if unnorm:  # the unnormalised range wins (serializer.cpp:212-216)
        extra["scalarMappingRangeUnnormalized"] = {"minimum": 3.0, "maximum": 200.0}
    sc = vidi_scene(["a.raw"], (4, 4, 4), type_name, extra)
    assert api.scene_value_range(sc) == pytest.approx(expect, rel=1e-7)


def test_scene_without_a_range_and_tfn_table():
    sc = vidi_scene(["a.raw"], (4, 4, 4), "FLOAT")
    assert api.scene_value_range(sc) is None
    assert api.scene_value_range(vidi_scene(["a.raw"], (4, 4, 4), "FLOAT", {"scalarMappingRange": {}})) == (0.0, 0.0)  # rangeFromJson
    with pytest.raises(api.VnrAmdError, match="tfn module"):
        api.vnrCreateTransferFunction(sc)


# ------------------------------------------------------------------------------------------------ volumes (GPU)
@pytest.mark.gpu
def test_simple_volume_from_vidi_scene_with_time_steps(oracle, tmp_path):
    rng = np.random.default_rng(0)
    dims = (24, 20, 18)
    steps = [rng.integers(0, 65535, dims[::-1], dtype=np.uint16) for _ in range(3)]
    files = []
    for i, s in enumerate(steps):
        f = tmp_path / f"t{i}.raw"
        with open(f, "wb") as h:
            h.write(b"\0" * 16)
            h.write(s.astype(">u2").tobytes())     # big endian on disk
        files.append(str(f))
    sc = vidi_scene(files, dims, "UNSIGNED_SHORT", {"scalarMappingRangeUnnormalized": {"minimum": 1000.0, "maximum": 60000.0}},
                    offset=16, endian="BIG_ENDIAN")
    sc["dataSource"][0]["fileName"] = [str(tmp_path / "missing.raw"), files[0]]   # first existing name wins (serializer.cpp:115-134)
    sv = api.vnrCreateSimpleVolume(sc, "GPU")
    assert api.vnrVolumeGetDims(sv) == dims and api.vnrSimpleVolumeGetNumberOfTimeSteps(sv) == 3
    coords = rng.uniform(0, 1, (2000, 3)).astype(np.float32)

    def norm(a):
        return np.clip((a.astype(np.float32) - np.float32(1000.0)) / (np.float32(60000.0) - np.float32(1000.0)), 0, 1).astype(np.float32)

    for t in (0, 2, 1, 1, 0):
        api.vnrSimpleVolumeSetCurrentTimeStep(sv, t)
        got = api.simple_volume_sample(sv, coords, nodal=False)
        assert np.array_equal(got, oracle.sample_volume(norm(steps[t]), coords, nodal=False))
        mc = api.volume_macrocell(sv)
        assert np.array_equal(mc["value_range"], oracle.macrocell_compute_implicit(norm(steps[t])))
    with pytest.raises(api.VnrAmdError, match="out of range"):
        api.vnrSimpleVolumeSetCurrentTimeStep(sv, 3)


@pytest.mark.gpu
def test_simple_volume_from_diva_scene_and_modes(oracle, tmp_path, monkeypatch):
    rng = np.random.default_rng(1)
    vol = rng.normal(0, 1, (6, 40, 260)).astype(np.float32)
    f = tmp_path / "v.raw"
    vol.tofile(f)
    sc = {"version": "DIVA", "volume": {"dims": {"x": 260, "y": 40, "z": 6}, "type": "FLOAT", "range": {"x": -1.0, "y": 1.5},
                                        "filename": str(f)}}
    monkeypatch.chdir(tmp_path)
    sv = api.vnrCreateSimpleVolume(sc, "GPU", True)               # save_loaded_volume -> ./reference.bin (neural_sampler.cu:101-108)
    want = np.clip((vol - np.float32(-1.0)) / np.float32(2.5), 0, 1).astype(np.float32)
    assert np.array_equal(np.fromfile(tmp_path / "reference.bin", np.float32).reshape(vol.shape), want)
    # the same scene out of core (reference defaults come from the environment, neural_sampler.cpp:1054-1062)
    monkeypatch.setenv("VNR_NUM_CONCURRENT_BLOCKS", "4")
    monkeypatch.setenv("VNR_NUM_BLOCKS", "12")
    ov = api.vnrCreateSimpleVolume(sc, "OUT_OF_CORE")
    info = api.out_of_core_info(ov)
    assert (info["n_concurrent_blocks"], info["n_blocks"]) == (4, 12) and info["file_dims"] == (260, 40, 6)
    blocks = api.out_of_core_blocks(ov)
    c, v = api.simple_volume_take_samples(ov, 512)
    r = oracle.pcg32_floats(5 * 512, 0, 1337, 0xda3e39cb94b95bdb)
    wc, wv, _ = oracle.OocSlabSet(vol, blocks).sample((-1.0, 1.5), r[:1536].reshape(512, 3), r[1536:2048], r[2048:])
    assert np.array_equal(c, wc) and np.array_equal(v, wv)
    # a shape without data, and the modes this build does not have
    nv = api.vnrCreateSimpleVolume(sc, "NOTHING")
    assert api.vnrVolumeGetDims(nv) == (260, 40, 6)
    with pytest.raises(api.VnrAmdError, match="not implemented"):
        api.vnrCreateSimpleVolume(sc, "VIRTUAL_MEMORY")
    with pytest.raises(api.VnrAmdError, match="unknown mode"):
        api.vnrCreateSimpleVolume(sc, "SOMETHING")
    with pytest.raises(api.VnrAmdError, match="cannot open"):
        api.vnrCreateSimpleVolume(dict(sc, volume=dict(sc["volume"], filename="/nonexistent.raw")), "GPU")


@pytest.mark.gpu
def test_vnr_cmd_train_command_line(tmp_path):
    """tools/vnr_cmd_train.py takes the reference's flags (apps/batch_trainer.cpp:30-70): scene document in, Summary block and
    ./params.json (BSON) out, which vnrCreateNeuralVolume(params) loads again"""
    import os
    import subprocess
    import sys
    from instantvnr_amd import synthetic as syn
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    vol = syn.analytic_volume(48)
    vol.astype(np.float32).tofile(tmp_path / "v.raw")
    scene = vidi_scene([str(tmp_path / "v.raw")], (48, 48, 48), "FLOAT")
    (tmp_path / "scene.json").write_text(json.dumps(scene))
    model = syn.model_config(n_levels=6, n_features=4, log2_hashmap_size=14, base_resolution=4, n_hidden_layers=2)
    (tmp_path / "model.json").write_text("// model file with a comment\n" + json.dumps(model))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "vnr_cmd_train.py"), "--volume", str(tmp_path / "scene.json"),
                          "--network", str(tmp_path / "model.json"), "--max-num-steps", "300", "--report", str(tmp_path / "log"),
                          "--quiet", "--train-macrocell"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    summary = dict(line.strip().split("=") for line in out.stdout.splitlines() if "=" in line)
    assert summary["STEP"] == "300" and float(summary["PSNR"]) > 30.0 and 0.5 < float(summary["SSIM"]) <= 1.0
    log = (tmp_path / "log.csv").read_text().splitlines()
    assert log[0] == "step,loss" and len(log) == 31 and log[-1].startswith("300,")
    nv = api.vnrCreateNeuralVolume(str(tmp_path / "params.json"))     # api.h:124: params.json with volume dims + model + parameters
    assert api.vnrVolumeGetDims(nv) == (48, 48, 48) and api.neural_info(nv)["n_levels"] == 6
    # resuming from it continues the step count? no: the reference's params.json carries no optimizer state; it must at least load
    out2 = subprocess.run([sys.executable, os.path.join(root, "tools", "vnr_cmd_train.py"), "--volume", str(tmp_path / "scene.json"),
                           "--network", str(tmp_path / "model.json"), "--resume", str(tmp_path / "params.json"), "--max-num-steps", "20",
                           "--quiet"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stderr[-2000:]
    assert float(dict(l.strip().split("=") for l in out2.stdout.splitlines() if "=" in l)["PSNR"]) > 30.0


def test_transfer_function_from_a_decoded_table():
    """the part of create_scene_vidi__tfn after OVR's tfn::loadTransferFunction (serializer.cpp:195-256): rgb -> colours, alpha at
    i / (resolution - 1), the end alphas below 0.01 forced to zero, the scene's value range"""
    sc = vidi_scene(["a.raw"], (8, 8, 8), "UNSIGNED_BYTE", volume_extra={"scalarMappingRange": {"minimum": 0.25, "maximum": 0.5}})
    table = np.array([[1, 0, 0, 0.005], [0, 1, 0, 0.5], [0, 0, 1, 0.2], [1, 1, 1, 0.02]], np.float32)
    t = api.vnrCreateTransferFunction(sc, table=table)
    from instantvnr_amd._lib import lib
    import ctypes as C
    nc, na = C.c_int(), C.c_int()
    assert lib().vnrAmdTransferFunctionGetSizes(t.h, C.byref(nc), C.byref(na)) == 0 and (nc.value, na.value) == (4, 4)
    rgb, xy, rng = np.zeros((4, 3), np.float32), np.zeros((4, 2), np.float32), np.zeros(2, np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    assert lib().vnrAmdTransferFunctionGet(t.h, fp(rgb), fp(xy), fp(rng)) == 0
    assert np.array_equal(rgb, table[:, :3])
    assert np.array_equal(xy[:, 0], np.array([0, 1, 2, 3], np.float32) / np.float32(3))
    assert np.array_equal(xy[:, 1], np.array([0.0, 0.5, 0.2, 0.02], np.float32))
    assert tuple(rng) == (0.25 * 255, 0.5 * 255)
    with pytest.raises(api.VnrAmdError, match="tfn::loadTransferFunction"):
        api.vnrCreateTransferFunction(sc)
    with pytest.raises(api.VnrAmdError, match="at least two"):
        api.vnrCreateTransferFunction(sc, table=table[:1])


def _read_png_rgba(path):
    import struct
    import zlib
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    o, w, h, data = 8, 0, 0, b""
    while o < len(b):
        n, tag = struct.unpack(">I", b[o:o + 4])[0], b[o + 4:o + 8]
        body = b[o + 8:o + 8 + n]
        assert zlib.crc32(tag + body) & 0xFFFFFFFF == struct.unpack(">I", b[o + 8 + n:o + 12 + n])[0]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
        if tag == b"IDAT":
            data += body
        o += 12 + n
    raw = np.frombuffer(zlib.decompress(data), np.uint8).reshape(h, 1 + 4 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 4)


@pytest.mark.gpu
def test_vnr_cmd_render_command_line(tmp_path):
    """tools/vnr_cmd_render.py takes the reference's flags (apps/batch_renderer.cpp:62-133): 768 x 768, 5 warm-up frames, the log, the
    screenshot (the frame the API renders, flipped and quantised like saveJPG) and the Summary block; for a scene and for params.json"""
    import os
    import subprocess
    import sys
    from instantvnr_amd import synthetic as syn
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "vnr_cmd_render.py")
    vol = syn.analytic_volume(48)
    vol.astype(np.float32).tofile(tmp_path / "v.raw")
    scene = vidi_scene([str(tmp_path / "v.raw")], (48, 48, 48), "FLOAT")
    scene["view"]["camera"] = {"eye": {"x": 100.0, "y": 90.0, "z": -60.0}, "center": {"x": 24.0, "y": 24.0, "z": 24.0},
                               "up": {"x": 0.0, "y": 1.0, "z": 0.0}, "fovy": 40.0}
    (tmp_path / "scene.json").write_text(json.dumps(scene))
    x = np.linspace(0, 1, 64, dtype=np.float32)
    table = np.stack([x, 1 - x, 0.5 + 0.5 * np.sin(6 * x), np.clip(1.5 * x - 0.2, 0, 1)], axis=1).astype(np.float32)
    np.save(tmp_path / "table.npy", table)
    base = [sys.executable, tool, "--tfn", str(tmp_path / "scene.json"), "--num-frames", "7", "--sampling-rate", "1.5"]
    # without the decoded table the tool stops and says why
    out = subprocess.run(base + ["--simple-volume", str(tmp_path / "scene.json")], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "tfn::loadTransferFunction" in out.stderr
    # the two volume flags exclude each other; one is required
    out = subprocess.run(base + ["--tfn-table", str(tmp_path / "table.npy")], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "required" in out.stderr
    out = subprocess.run(base + ["--tfn-table", str(tmp_path / "table.npy"), "--simple-volume", str(tmp_path / "scene.json"), "--rendering-mode", "5",
                                 "--exp", "gt"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    i = lines.index("Summary: gt")
    assert lines[i + 1] == f"\tvolume: {tmp_path / 'scene.json'}" and lines[i + 3].startswith("\t   fps: ") and float(lines[i + 3].split(":")[1]) > 1.0
    assert lines[i + 5] == "\tsampling rate: 1.5" and lines[i + 6] == "\tcamera: (76,66,-84)" and lines[i + 7].strip() == "(0,0,0)"
    log = (tmp_path / "gt.csv").read_text().splitlines()
    assert log[0] == "#,frame time,fps" and len(log) == 8 and log[7].startswith("6.0,")
    shot = _read_png_rgba(tmp_path / "gt-screenshot.png")
    # the same frame through the API: 5 + 7 frames accumulate (mode 5 jitters per frame)
    sv = api.vnrCreateSimpleVolume(str(tmp_path / "scene.json"), "GPU")
    cam = api.vnrCreateCamera(str(tmp_path / "scene.json"))
    tfn = api.vnrCreateTransferFunction(str(tmp_path / "scene.json"), table=table)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    r = api.vnrCreateRenderer(sv)
    api.vnrRendererSetTransferFunction(r, tfn)
    api.vnrRendererSetCamera(r, cam)
    api.vnrRendererSetFramebufferSize(r, (768, 768))
    api.vnrRendererSetMode(r, 5)
    api.vnrRendererSetVolumeSamplingRate(r, 1.5)
    for _ in range(12):
        api.vnrRender(r)
    frame = api.vnrRendererMapFrame(r).copy()
    want = (np.float32(255.99) * np.clip(frame, 0, 1)).astype(np.uint32).astype(np.uint8)[::-1]
    assert shot.shape == (768, 768, 4) and (shot[..., 3] > 0).mean() > 0.1
    assert np.array_equal(shot, want)
    # a neural volume from params.json
    cfg = syn.model_config(n_levels=4, n_features=4, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2)
    nv = api.vnrCreateNeuralVolume(cfg, sv, True)
    api.vnrNeuralVolumeTrain(nv, 200, True)
    api.vnrNeuralVolumeSerializeParams(nv, str(tmp_path / "params.json"))
    out = subprocess.run(base + ["--tfn-table", str(tmp_path / "table.npy"), "--neural-volume", str(tmp_path / "params.json"), "--rendering-mode", "5",
                                 "--exp", "nn"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    shot_nn = _read_png_rgba(tmp_path / "nn-screenshot.png")
    err = shot_nn[..., :3].astype(np.float32) - shot[..., :3].astype(np.float32)
    assert 10 * np.log10(255.0 ** 2 / float((err ** 2).mean())) > 20.0    # 200 steps of a 4-level model: the same picture, roughly


def test_every_key_of_the_reference_s_scene_reader_is_read_here_or_explained():
    """tools/scene_keys_vs_reference.py, where the reference's serializer.cpp is present: its JSON keys and enumeration strings against
    csrc/scene.cpp's"""
    import subprocess
    import sys
    if not os.path.exists("/root/reference/serializer.cpp"):
        pytest.skip("the reference's sources are not on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "scene_keys_vs_reference.py")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "UNEXPLAINED" not in out.stdout, out.stdout


@pytest.mark.gpu
def test_a_scene_that_claims_more_than_its_file_holds_is_refused_before_anything_is_allocated(tmp_path):
    """found by tests/test_gpu_fuzz.py: the loader sized (and zero-filled) its buffer from the DESCRIPTION and only then read the file, so a
    scene with one wrong dimension took the host's memory instead of failing; now the file's size is checked first"""
    f = tmp_path / "v.raw"
    f.write_bytes(bytes(16 + 20 * 12 * 9))
    for dims, offset, msg in [((4096, 4096, 4096), 16, "too short"), ((20, 12, 10), 16, "too short"), ((20, 12, 9), 17, "too short"),
                              ((-1, 12, 9), 16, "positive"), ((0, 12, 9), 16, "positive"), ((2 ** 31 - 1, 2 ** 31 - 1, 2 ** 31 - 1), 0, "too short")]:
        sc = vidi_scene([str(f)], dims, "UNSIGNED_BYTE", offset=offset)
        with pytest.raises(api.VnrAmdError, match=msg):
            api.vnrCreateSimpleVolume(sc, "GPU")
    sv = api.vnrCreateSimpleVolume(vidi_scene([str(f)], (20, 12, 9), "UNSIGNED_BYTE", offset=16), "GPU")
    assert api.vnrVolumeGetDims(sv) == (20, 12, 9)
