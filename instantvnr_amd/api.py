"""Python mirror of the reference's `vnr*` API (api.h:90-188) on top of the C-ABI (include/vnr_amd.h).

Same function names, argument meaning and error behaviour (errors raise, like the reference's
std::runtime_error).  JSON arguments may be dicts, JSON text, BSON bytes, or a path string (the reference
treats a JSON string value as a path: api.cpp:77-83).  This layer is host plumbing only — all compute runs in
libvnr_amd.so; there is no CPU fallback.
"""
import ctypes as C
import json

import numpy as np

from . import _lib
from ._lib import VnrAmdError, check, check_ptr, lib

JSON_TEXT, JSON_BSON, JSON_TEXT_FILE, JSON_BSON_FILE = 0, 1, 2, 3

VALUE_TYPES = {np.dtype(np.uint8): 0, np.dtype(np.int8): 1, np.dtype(np.uint16): 2, np.dtype(np.int16): 3,
               np.dtype(np.uint32): 4, np.dtype(np.int32): 5, np.dtype(np.float32): 8, np.dtype(np.float64): 12}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _vec(v, n=3):
    a = np.ascontiguousarray(v, dtype=np.float32).ravel()
    assert a.size == n
    return a


def _json_arg(j, params=False):
    """-> (buffer, size, format, keepalive)"""
    if isinstance(j, (dict, list)):
        b = json.dumps(j).encode()
        return b, len(b), JSON_TEXT
    if isinstance(j, (bytes, bytearray, memoryview)):
        b = bytes(j)
        return b, len(b), JSON_BSON
    if isinstance(j, str):
        s = j.lstrip()
        if s.startswith("{") or s.startswith("/") and s[1:2] in "/*":
            b = j.encode()
            return b, len(b), JSON_TEXT
        b = j.encode() + b"\0"
        return b, len(b), JSON_BSON_FILE if params else JSON_TEXT_FILE
    raise TypeError("unsupported JSON argument")


# ------------------------------------------------------------------------------------------------ device memory
class DeviceArray:
    """typed device buffer for hosts without a HIP binding (wraps vnrAmdMalloc/Memcpy)"""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = check_ptr(lib().vnrAmdMalloc(max(self.nbytes, 1)))

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        d.upload(a)
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes
        if self.nbytes:
            check(lib().vnrAmdMemcpyH2D(self.ptr, a.ctypes.data_as(C.c_void_p), self.nbytes))

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            check(lib().vnrAmdMemcpyD2H(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out

    def zero(self):
        check(lib().vnrAmdMemset(self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr:
            lib().vnrAmdFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------ handles
class _Handle:
    _release = None

    def __init__(self, h):
        self.h = check_ptr(h)

    def release(self):
        if self.h and self._release:
            getattr(lib(), self._release)(self.h)
        self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class vnrVolume(_Handle):
    _release = "vnrAmdReleaseVolume"
    _keep = None


class vnrRenderer(_Handle):
    _release = "vnrAmdReleaseRenderer"
    _keep = None


class vnrTransferFunction(_Handle):
    _release = "vnrAmdReleaseTransferFunction"


class vnrCamera(_Handle):
    _release = "vnrAmdReleaseCamera"


# ------------------------------------------------------------------------------------------------ json (api.h:90-96)
def _convert(data, size, fmt_in, fmt_out):
    out = C.c_void_p()
    n = C.c_size_t()
    check(lib().vnrAmdJsonConvert(data, size, fmt_in, fmt_out, C.byref(out), C.byref(n)))
    b = C.string_at(out, n.value)
    lib().vnrAmdFreeHost(out)
    return b


def vnrCreateJsonText(filename):
    f = filename.encode() + b"\0"
    return json.loads(_convert(f, len(f), JSON_TEXT_FILE, JSON_TEXT))


def vnrCreateJsonBinary(filename):
    """returns the BSON document as bytes (binary members cannot live in a Python dict losslessly)"""
    f = filename.encode() + b"\0"
    return _convert(f, len(f), JSON_BSON_FILE, JSON_BSON)


def vnrSaveJsonText(j, filename):
    b, n, f = _json_arg(j)
    check(lib().vnrAmdJsonSave(b, n, f, filename.encode(), JSON_TEXT))


def vnrSaveJsonBinary(j, filename):
    b, n, f = _json_arg(j, params=True)
    check(lib().vnrAmdJsonSave(b, n, f, filename.encode(), JSON_BSON))


def json_to_bson(j):
    b, n, f = _json_arg(j)
    return _convert(b, n, f, JSON_BSON)


def bson_to_json_text(b):
    return _convert(bytes(b), len(b), JSON_BSON, JSON_TEXT).decode()


# ------------------------------------------------------------------------------------------------ camera (api.h:103-110)
def vnrCreateCamera(scene=None):
    """api.h:103-104: vnrCreateCamera() / vnrCreateCamera(scene json or path)"""
    cam = vnrCamera(lib().vnrAmdCreateCamera())
    if scene is not None:
        vnrCameraSet(cam, scene)
    return cam


def vnrCameraSet(cam, frm, at=None, up=None, fovy=None):
    """api.h:105-106: vnrCameraSet(self, from, at, up) / vnrCameraSet(self, scene)"""
    if at is None:
        b, n, f = _json_arg(frm)
        check(lib().vnrAmdCameraSetFromScene(cam.h, b, n, f))
        return
    check(lib().vnrAmdCameraSet(cam.h, _fp(_vec(frm)), _fp(_vec(at)), _fp(_vec(up))))
    if fovy is not None:
        check(lib().vnrAmdCameraSetFovy(cam.h, float(fovy)))


def _cam_get(cam):
    f, a, u = (np.zeros(3, np.float32) for _ in range(3))
    fov = C.c_float()
    check(lib().vnrAmdCameraGet(cam.h, _fp(f), _fp(a), _fp(u), C.byref(fov)))
    return f, a, u, fov.value


def vnrCameraGetPosition(cam):
    return _cam_get(cam)[0]


def vnrCameraGetFocus(cam):
    return _cam_get(cam)[1]


def vnrCameraGetUpVec(cam):
    return _cam_get(cam)[2]


# ------------------------------------------------------------------------------------------------ volumes (api.h:117-148)
def vnrCreateSimpleVolume(data, mode_or_range=None, save_loaded_volume=False, value_range=None):
    """api.h:117 vnrCreateSimpleVolume(scene, mode, save_loaded_volume): `data` is a scene document (dict / JSON text / path)
    and mode "GPU", "OUT_OF_CORE" or "NOTHING"; or, an extension for hosts that hold the voxels already, `data` is a numpy
    array [z, y, x] (x fastest) and the second argument an optional (lo, hi) value range"""
    if not isinstance(data, np.ndarray):
        b, n, f = _json_arg(data)
        mode = "GPU" if mode_or_range is None else mode_or_range
        return vnrVolume(lib().vnrAmdCreateSimpleVolumeFromScene(b, n, f, mode.encode(), 1 if save_loaded_volume else 0))
    if value_range is None:
        value_range = mode_or_range
    a = np.ascontiguousarray(data)
    if a.dtype not in VALUE_TYPES:
        raise VnrAmdError("unknown data type")
    dims = (C.c_int * 3)(a.shape[2], a.shape[1], a.shape[0])
    lo, hi = (1.0, 0.0) if value_range is None else value_range
    return vnrVolume(lib().vnrAmdCreateSimpleVolumeFromMemory(a.ctypes.data_as(C.c_void_p), dims, VALUE_TYPES[a.dtype], lo, hi))


def vnrCreateSimpleVolumeFromRawFile(filename, dims, dtype, offset=0, big_endian=False, value_range=None):
    d = (C.c_int * 3)(*[int(v) for v in dims])
    lo, hi = (1.0, 0.0) if value_range is None else value_range
    return vnrVolume(lib().vnrAmdCreateSimpleVolumeFromRawFile(filename.encode(), d, VALUE_TYPES[np.dtype(dtype)], offset,
                                                               1 if big_endian else 0, lo, hi))


def vnrSimpleVolumeGetNumberOfTimeSteps(v):
    n = lib().vnrAmdSimpleVolumeGetNumberOfTimeSteps(v.h)
    if n < 0:
        raise VnrAmdError(_lib.last_error())
    return n


def vnrSimpleVolumeSetCurrentTimeStep(v, index):
    check(lib().vnrAmdSimpleVolumeSetCurrentTimeStep(v.h, int(index)))


def scene_value_range(scene):
    """the value range a scene maps its transfer function to, or None (serializer.cpp:212-256)"""
    b, n, f = _json_arg(scene)
    r = np.zeros(2, np.float32)
    st = lib().vnrAmdSceneGetValueRange(b, n, f, _fp(r))
    if st == 2:
        return None
    check(st)
    return float(r[0]), float(r[1])


def vnrCreateSimpleVolumeOutOfCore(filename, dims, dtype, value_range, offset=0, n_concurrent_blocks=0, n_blocks=0):
    """vnrCreateSimpleVolume(scene, "OUT_OF_CORE") (api.cpp:145-158): the volume stays in `filename`; 0 block counts = the
    reference's defaults / environment variables (neural_sampler.cpp:1054-1062)"""
    d = (C.c_int * 3)(*[int(v) for v in dims])
    lo, hi = value_range
    return vnrVolume(check_ptr(lib().vnrAmdCreateSimpleVolumeOutOfCore(str(filename).encode(), d, VALUE_TYPES[np.dtype(dtype)], offset,
                                                                       lo, hi, n_concurrent_blocks, n_blocks)))


def out_of_core_info(v):
    info = _lib.OutOfCoreInfo()
    check(lib().vnrAmdSimpleVolumeOutOfCoreInfo(v.h, C.byref(info)))
    return {"file_dims": tuple(info.file_dims), "block_dims": tuple(info.block_dims),
            "block_index_space": tuple(info.block_index_space), "n_blocks": info.n_blocks,
            "n_concurrent_blocks": info.n_concurrent_blocks, "block_size_aligned": info.block_size_aligned,
            "bytes_read": info.bytes_read}


def out_of_core_blocks(v):
    """block index (y, z) of every resident slot, as the next take_samples call will see them"""
    n = out_of_core_info(v)["n_blocks"]
    a = np.zeros((n, 2), np.int32)
    check(lib().vnrAmdSimpleVolumeOutOfCoreBlocks(v.h, a.ctypes.data_as(C.POINTER(C.c_int)), n))
    return a


def vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=4.0):
    d = (C.c_int * 3)(*[int(v) for v in dims])
    return vnrVolume(lib().vnrAmdCreateSimpleVolumePerlin(d, seed, octaves, base_frequency))


def vnrCreateNeuralVolume(config, groundtruth_or_dims=None, online_macrocell_construction=True):
    """the three overloads of api.h:122-124: (config, groundtruth[, online]), (config, dims), (params)"""
    L = lib()
    if groundtruth_or_dims is None:
        b, n, f = _json_arg(config, params=True)
        return vnrVolume(L.vnrAmdCreateNeuralVolumeFromParams(b, n, f))
    b, n, f = _json_arg(config)
    if isinstance(groundtruth_or_dims, vnrVolume):
        v = vnrVolume(L.vnrAmdCreateNeuralVolume(b, n, f, groundtruth_or_dims.h, 1 if online_macrocell_construction else 0))
        v._keep = groundtruth_or_dims
        return v
    d = (C.c_int * 3)(*[int(x) for x in groundtruth_or_dims])
    return vnrVolume(L.vnrAmdCreateNeuralVolumeFromDims(b, n, f, d))


def vnrNeuralVolumeSetModel(v, config):
    b, n, f = _json_arg(config)
    check(lib().vnrAmdNeuralVolumeSetModel(v.h, b, n, f))


def vnrNeuralVolumeSetParams(v, params):
    b, n, f = _json_arg(params, params=True)
    check(lib().vnrAmdNeuralVolumeSetParams(v.h, b, n, f))


def _dbl(x):
    if x == -1.0 and _lib.last_error():
        pass
    return x


def vnrNeuralVolumeGetPSNR(v, verbose=False):
    r = lib().vnrAmdNeuralVolumeGetPSNR(v.h, 1 if verbose else 0)
    if r == -1.0:
        raise VnrAmdError(_lib.last_error())
    return r


def vnrNeuralVolumeGetSSIM(v, verbose=False):
    r = lib().vnrAmdNeuralVolumeGetSSIM(v.h, 1 if verbose else 0)
    if r == -1.0:
        raise VnrAmdError(_lib.last_error())
    return r


def vnrNeuralVolumeGetTestingLoss(v):
    r = lib().vnrAmdNeuralVolumeGetTestingLoss(v.h)
    if r == -1.0:
        raise VnrAmdError(_lib.last_error())
    return r


def vnrNeuralVolumeGetTrainingLoss(v):
    return lib().vnrAmdNeuralVolumeGetTrainingLoss(v.h)


def vnrNeuralVolumeGetTrainingStep(v):
    return lib().vnrAmdNeuralVolumeGetTrainingStep(v.h)


def vnrNeuralVolumeGetNumberOfBlobs(v):
    return lib().vnrAmdNeuralVolumeGetNumberOfBlobs(v.h)


def vnrNeuralVolumeTrain(v, steps, fast_mode):
    check(lib().vnrAmdNeuralVolumeTrain(v.h, int(steps), 1 if fast_mode else 0))


def vnrNeuralVolumeDecodeProgressive(v):
    """decodes the next blob of 16 z-slices into the dense volume rendering modes 4 / 7 march (api.h:137)"""
    check(lib().vnrAmdNeuralVolumeDecodeProgressive(v.h))


def vnrNeuralVolumeDecodeInference(v, filename):
    check(lib().vnrAmdNeuralVolumeDecodeInference(v.h, filename.encode()))


def vnrNeuralVolumeDecodeReference(v, filename):
    check(lib().vnrAmdNeuralVolumeDecodeReference(v.h, filename.encode()))


def neural_decoded_volume(v, dims):
    """the decoded dense volume as numpy [z, y, x] (AMD extension; None before the first decode)"""
    p = lib().vnrAmdNeuralVolumeDecodedDeviceData(v.h)
    if not p:
        return None
    out = np.empty((dims[2], dims[1], dims[0]), dtype=np.float32)
    check(lib().vnrAmdMemcpyD2H(out.ctypes.data_as(C.c_void_p), p, out.nbytes))
    return out


def vnrNeuralVolumeSerializeParams(v, filename=None):
    """with a filename: writes BSON params.json; without: returns the BSON bytes"""
    if filename is not None:
        check(lib().vnrAmdNeuralVolumeSerializeParamsToFile(v.h, filename.encode()))
        return None
    out = C.c_void_p()
    n = C.c_size_t()
    check(lib().vnrAmdNeuralVolumeSerializeParams(v.h, C.byref(out), C.byref(n)))
    b = C.string_at(out, n.value)
    lib().vnrAmdFreeHost(out)
    return b


def vnrVolumeSetClippingBox(v, lower, upper):
    check(lib().vnrAmdVolumeSetClippingBox(v.h, _fp(_vec(lower)), _fp(_vec(upper))))


def vnrVolumeSetScaling(v, scale):
    check(lib().vnrAmdVolumeSetScaling(v.h, _fp(_vec(scale))))


def vnrVolumeGetDims(v):
    d = (C.c_int * 3)()
    check(lib().vnrAmdVolumeGetDims(v.h, d))
    return tuple(d)


def vnrVolumeGetValueRange(v):
    r = np.zeros(2, np.float32)
    check(lib().vnrAmdVolumeGetValueRange(v.h, _fp(r)))
    return tuple(r)


# ------------------------------------------------------------------------------------------------ isosurface (core/marching_cube.cuh:6-8)
def vnrMarchingCube(v, isovalue):
    """vnrMarchingCube(volume, isovalue, &ptr, &size, false): -> float32 [n_vertices, 3], three vertices per triangle, voxel units"""
    p = C.POINTER(C.c_float)()
    n = C.c_size_t()
    check(lib().vnrAmdMarchingCube(v.h, float(isovalue), C.byref(p), C.byref(n), 0))
    if n.value == 0:
        return np.zeros((0, 3), np.float32)
    out = np.ctypeslib.as_array(p, shape=(n.value, 3)).copy()
    lib().vnrAmdFreeHost(p)
    return out


def vnrSaveTriangles(filename, vertices):
    """vnrSaveTriangles(filename, ptr, size): Wavefront OBJ"""
    a = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    check(lib().vnrAmdSaveTriangles(str(filename).encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0]))


# ------------------------------------------------------------------------------------------------ tfn (api.h:154-162)
def vnrCreateTransferFunction(scene=None, table=None):
    """api.h:154-155.  vnrCreateTransferFunction(scene) decodes the scene's transfer function with OVR's tfn module
    (tfn::loadTransferFunction, serializer.cpp:192-193), which is not part of the reference tree.  `table` is what that module
    yields (tfn::TransferFunctionCore::data(): resolution x RGBA): given it, this does the rest of create_scene_vidi__tfn
    (serializer.cpp:195-256): colours = rgb, alphas at i / (resolution - 1), end alphas below 0.01 forced to zero, value range from
    the scene.  A scene without a table is refused with an explanation rather than rendered with an invented table."""
    if scene is not None and table is None:
        raise VnrAmdError("vnrCreateTransferFunction(scene): the transfer-function table of a scene is decoded by OVR's tfn module "
                          "(tfn::loadTransferFunction), which is outside the reference tree; pass the decoded RGBA table (table=) or "
                          "set colours / alphas explicitly (the scene's value range: scene_value_range)")
    t = vnrTransferFunction(lib().vnrAmdCreateTransferFunction())
    if table is not None:
        rgba = np.asarray(table, dtype=np.float32).reshape(-1, 4)
        if rgba.shape[0] < 2:
            raise VnrAmdError("vnrCreateTransferFunction: a transfer-function table needs at least two RGBA entries")
        alpha = np.stack([np.arange(rgba.shape[0], dtype=np.float32) / np.float32(rgba.shape[0] - 1), rgba[:, 3]], axis=1)
        for i in (0, -1):
            if alpha[i, 1] < 0.01:
                alpha[i, 1] = 0.0
        vnrTransferFunctionSetColor(t, rgba[:, :3])
        vnrTransferFunctionSetAlpha(t, alpha)
        if scene is not None:
            rng = scene_value_range(scene)
            if rng is not None:
                vnrTransferFunctionSetValueRange(t, rng)
    return t


def vnrTransferFunctionSetColor(t, colors):
    c = np.ascontiguousarray(colors, dtype=np.float32).reshape(-1, 3)
    check(lib().vnrAmdTransferFunctionSetColor(t.h, _fp(c), c.shape[0]))


def vnrTransferFunctionSetAlpha(t, alphas):
    """alphas: [n,2] (position, alpha) like the reference's vec2f list, or [n] alpha values"""
    a = np.asarray(alphas, dtype=np.float32)
    if a.ndim == 1:
        a = np.stack([np.linspace(0, 1, a.size, dtype=np.float32), a], axis=1)
    a = np.ascontiguousarray(a)
    check(lib().vnrAmdTransferFunctionSetAlpha(t.h, _fp(a), a.shape[0]))


def vnrTransferFunctionSetValueRange(t, rng):
    check(lib().vnrAmdTransferFunctionSetValueRange(t.h, float(rng[0]), float(rng[1])))


# ------------------------------------------------------------------------------------------------ renderer (api.h:168-178)
def vnrCreateRenderer(v):
    r = vnrRenderer(lib().vnrAmdCreateRenderer(v.h))
    r._keep = v
    r._size = (0, 0)
    r._device_output = False
    return r


def vnrRendererSetFramebufferSize(r, size):
    check(lib().vnrAmdRendererSetFramebufferSize(r.h, int(size[0]), int(size[1])))
    r._size = (int(size[0]), int(size[1]))


def vnrRendererSetTransferFunction(r, t):
    check(lib().vnrAmdRendererSetTransferFunction(r.h, t.h))


def vnrRendererSetCamera(r, cam):
    check(lib().vnrAmdRendererSetCamera(r.h, cam.h))


def vnrRendererSetMode(r, mode):
    check(lib().vnrAmdRendererSetMode(r.h, int(mode)))


def vnrRendererSetDenoiser(r, flag):
    check(lib().vnrAmdRendererSetDenoiser(r.h, 1 if flag else 0))


def vnrRendererSetVolumeSamplingRate(r, v):
    check(lib().vnrAmdRendererSetVolumeSamplingRate(r.h, float(v)))


def vnrRendererSetVolumeDensityScale(r, v):
    check(lib().vnrAmdRendererSetVolumeDensityScale(r.h, float(v)))


def vnrRendererResetAccumulation(r):
    check(lib().vnrAmdRendererResetAccumulation(r.h))


def vnrRender(r):
    check(lib().vnrAmdRender(r.h))


def vnrRendererMapFrame(r):
    """host frame as a numpy view [h, w, 4] (valid until two frames later), or the raw device pointer"""
    p = check_ptr(lib().vnrAmdRendererMapFrame(r.h))
    if r._device_output:
        return p
    w, h = r._size
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(h, w, 4))


# misc (api.h:185-188) -----------------------------------------------------------------------------------------------
def vnrMemoryQuery():
    """-> (bytes used by the renderer, bytes used by the network engine)"""
    r, n = C.c_size_t(), C.c_size_t()
    lib().vnrAmdMemoryQuery(C.byref(r), C.byref(n))
    return r.value, n.value


def vnrFreeTemporaryGPUMemory():
    lib().vnrAmdFreeTemporaryGPUMemory()


# AMD extensions -------------------------------------------------------------------------------------------------
def vnrRendererSetPixelRange(r, lo, hi):
    check(lib().vnrAmdRendererSetPixelRange(r.h, int(lo), int(hi)))


def vnrRendererSetPixelInterleave(r, block_pixels, n_parts, part):
    check(lib().vnrAmdRendererSetPixelInterleave(r.h, int(block_pixels), int(n_parts), int(part)))


def vnrRendererSetOutputAsDeviceFramebuffer(r, flag):
    check(lib().vnrAmdRendererSetOutputAsDeviceFramebuffer(r.h, 1 if flag else 0))
    r._device_output = bool(flag)


def vnrRendererSetProfiling(r, flag):
    check(lib().vnrAmdRendererSetProfiling(r.h, 1 if flag else 0))


def vnrRendererGetFrameStats(r):
    s = _lib.FrameStats()
    check(lib().vnrAmdRendererGetFrameStats(r.h, C.byref(s)))
    return {k: getattr(s, k) for k, _ in s._fields_}


def renderer_schedule(r):
    """which schedule the last sample-streaming frame ran (vnrAmdRendererDebugSchedule)"""
    out = (C.c_int * 4)()
    check(lib().vnrAmdRendererDebugSchedule(r.h, out))
    return {"n_iters": out[0], "n_parts": out[1], "fused_pack": bool(out[2]), "decoupled": bool(out[3])}


def neural_brick_image(v):
    """state of the de-hashed inference copy of the hashed levels (csrc/network.h)"""
    u, b, ms = C.c_int(), C.c_size_t(), C.c_float()
    check(lib().vnrAmdNeuralVolumeBrickImageInfo(v.h, C.byref(u), C.byref(b), C.byref(ms)))
    builds, after, tier, small = C.c_uint64(), C.c_uint(), C.c_int(), C.c_uint64()
    check(lib().vnrAmdNeuralVolumeBrickImagePolicy(v.h, C.byref(builds), C.byref(after), C.byref(tier), C.byref(small)))
    return {"in_use": bool(u.value), "bytes": b.value, "build_ms": ms.value, "levels": int(lib().vnrAmdNeuralVolumeBrickImageLevels(v.h)),
            "builds": int(builds.value), "launches_before_next_build": int(after.value), "tier": int(tier.value), "small_builds": int(small.value)}


def neural_set_brick_budget(v, n_bytes):
    """budget of the de-hashed inference copy in bytes (0: the default policy); the next launches rebuild it"""
    check(lib().vnrAmdNeuralVolumeSetBrickImageBudget(v.h, int(n_bytes)))


def neural_info(v):
    vals = [C.c_int() for _ in range(5)]
    n = C.c_uint64()
    check(lib().vnrAmdNeuralVolumeGetInfo(v.h, *[C.byref(x) for x in vals], C.byref(n)))
    keys = ["n_levels", "n_features_per_level", "padded_width", "n_neurons", "n_hidden_layers"]
    d = {k: x.value for k, x in zip(keys, vals)}
    d["n_params"] = n.value
    kind = [C.c_int() for _ in range(6)]
    check(lib().vnrAmdNeuralVolumeGetModelKind(v.h, *[C.byref(x) for x in kind]))
    d.update({k: x.value for k, x in zip(["activation", "output_activation", "grid_type", "interpolation", "mfma_kernels", "mfma_training_kernels"], kind)})
    return d


def neural_grid_backward_plan(v, batch=65536):
    """the training step's grid-backward layout and its memory-side atomic requests, from the library (vnrAmdNeuralVolumeGridBackwardPlan)"""
    u32, u64 = (C.c_uint32 * 4)(), (C.c_uint64 * 2)()
    check(lib().vnrAmdNeuralVolumeGridBackwardPlan(v.h, int(batch), u32, u64))
    return {"n_levels": u32[0], "lds_levels": u32[1], "tile_entries": u32[2], "lds_blocks": u32[3], "atomic_requests": int(u64[0]), "flush_requests_at_most": int(u64[1])}


def neural_level_table(v):
    """the hash grid's levels as the library laid them out: list of dicts(res, entries, offset, kind) (kind: 0 dense, 1 hash, 2-4 tiled)"""
    a = [np.zeros(32, np.uint32) for _ in range(4)]
    n = lib().vnrAmdNeuralVolumeLevelTable(v.h, 32, *[x.ctypes.data_as(C.c_void_p) for x in a])
    if n < 0:
        check(1)
    return [{"res": int(a[0][l]), "entries": int(a[1][l]), "offset": int(a[2][l]), "kind": int(a[3][l])} for l in range(n)]


def neural_set_params_fp16(v, params):
    p = np.ascontiguousarray(params).view(np.uint16)
    check(lib().vnrAmdNeuralVolumeSetParamsFP16(v.h, p.ctypes.data_as(C.c_void_p), p.size))


def neural_get_params_fp16(v):
    n = neural_info(v)["n_params"]
    p = np.empty(n, dtype=np.uint16)
    check(lib().vnrAmdNeuralVolumeGetParamsFP16(v.h, p.ctypes.data_as(C.c_void_p), n))
    return p.view(np.float16)


def neural_inference(v, coords):
    """coords [n,3] float32 (host) -> values [n] float32 (host), through the fused HIP kernel"""
    c = DeviceArray.from_numpy(np.ascontiguousarray(coords, dtype=np.float32))
    o = DeviceArray((c.shape[0],), np.float32)
    check(lib().vnrAmdNeuralVolumeInference(v.h, c.shape[0], c.ptr, o.ptr, None))
    check(lib().vnrAmdSynchronize())
    return o.numpy()


def neural_encode(v, coords):
    info = neural_info(v)
    c = DeviceArray.from_numpy(np.ascontiguousarray(coords, dtype=np.float32))
    o = DeviceArray((c.shape[0], info["padded_width"]), np.uint16)
    check(lib().vnrAmdNeuralVolumeEncode(v.h, c.shape[0], c.ptr, o.ptr, None))
    check(lib().vnrAmdSynchronize())
    return o.numpy().view(np.float16)


def volume_macrocell(v):
    dims = (C.c_int * 3)()
    sp = (C.c_float * 3)()
    vr, mo = C.c_void_p(), C.c_void_p()
    check(lib().vnrAmdVolumeGetMacrocell(v.h, dims, sp, C.byref(vr), C.byref(mo)))
    n = dims[0] * dims[1] * dims[2]
    value_range = np.empty((dims[2], dims[1], dims[0], 2), np.float32)
    max_opacity = np.empty((dims[2], dims[1], dims[0]), np.float32)
    check(lib().vnrAmdMemcpyD2H(value_range.ctypes.data_as(C.c_void_p), vr, n * 8))
    check(lib().vnrAmdMemcpyD2H(max_opacity.ctypes.data_as(C.c_void_p), mo, n * 4))
    return {"dims": tuple(dims), "spacings": np.array(list(sp), np.float32), "value_range": value_range,
            "max_opacity": max_opacity}


def simple_volume_sample(v, coords, nodal):
    c = DeviceArray.from_numpy(np.ascontiguousarray(coords, dtype=np.float32))
    o = DeviceArray((c.shape[0],), np.float32)
    check(lib().vnrAmdSimpleVolumeSample(v.h, c.shape[0], c.ptr, o.ptr, 1 if nodal else 0, None))
    check(lib().vnrAmdSynchronize())
    return o.numpy()


def simple_volume_take_samples_grid(v, origin, size):
    n = int(size[0]) * int(size[1]) * int(size[2])
    c = DeviceArray((n, 3), np.float32)
    o = DeviceArray((n,), np.float32)
    check(lib().vnrAmdSimpleVolumeTakeSamplesGrid(v.h, (C.c_int * 3)(*[int(x) for x in origin]), (C.c_int * 3)(*[int(x) for x in size]),
                                                  c.ptr, o.ptr, None))
    check(lib().vnrAmdSynchronize())
    return c.numpy(), o.numpy()


def simple_volume_take_samples(v, n, lower=(0, 0, 0), upper=(1, 1, 1)):
    c = DeviceArray((n, 3), np.float32)
    o = DeviceArray((n,), np.float32)
    check(lib().vnrAmdSimpleVolumeTakeSamples(v.h, n, _fp(_vec(lower)), _fp(_vec(upper)), c.ptr, o.ptr, None))
    check(lib().vnrAmdSynchronize())
    return c.numpy(), o.numpy()


def neural_forward_backward(v, coords, targets):
    """forward + backward on a host batch; returns the (loss-scaled x128) gradient blob, which the library keeps in half precision, as float32 numpy"""
    c = DeviceArray.from_numpy(np.ascontiguousarray(coords, dtype=np.float32))
    t = DeviceArray.from_numpy(np.ascontiguousarray(targets, dtype=np.float32))
    check(lib().vnrAmdNeuralVolumeForwardBackward(v.h, c.shape[0], c.ptr, t.ptr))
    return neural_gradients(v)


def neural_gradients(v):
    n = C.c_size_t()
    p = check_ptr(lib().vnrAmdNeuralVolumeGradients(v.h, C.byref(n)))
    check(lib().vnrAmdSynchronize())
    out = np.empty(n.value, np.float32)
    check(lib().vnrAmdMemcpyD2H(out.ctypes.data_as(C.c_void_p), p, n.value * 4))
    return out


def neural_train_begin(v):
    check(lib().vnrAmdNeuralVolumeTrainBegin(v.h))


def neural_train_end(v, grad_scale=1.0, fast_mode=True):
    check(lib().vnrAmdNeuralVolumeTrainEnd(v.h, float(grad_scale), 1 if fast_mode else 0))
