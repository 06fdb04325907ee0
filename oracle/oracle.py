"""ctypes/numpy front-end of the CPU oracle (oracle/vnr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by instantvnr_amd.  PARITY UNPINNED (see
vnr_oracle.h).
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvnr_oracle.so")
MAX_LEVELS = 32


def build(force=False):
    src = os.path.join(_HERE, "vnr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class GridConfig(C.Structure):
    _fields_ = [("n_levels", C.c_uint32), ("n_features", C.c_uint32),
                ("log2_hashmap_size", C.c_uint32), ("base_resolution", C.c_uint32),
                ("per_level_scale", C.c_float), ("interpolation", C.c_uint32),
                ("quantize_threshold", C.c_float), ("max_level", C.c_float), ("grid_type", C.c_uint32)]


class GridLayout(C.Structure):
    _fields_ = [("offsets", C.c_uint32 * (MAX_LEVELS + 1)), ("scale", C.c_float * MAX_LEVELS),
                ("resolution", C.c_uint32 * MAX_LEVELS)]


class Tfn(C.Structure):
    _fields_ = [("colors", C.POINTER(C.c_float)), ("n_colors", C.c_int),
                ("alphas", C.POINTER(C.c_float)), ("n_alphas", C.c_int),
                ("range_lo", C.c_float), ("range_hi", C.c_float), ("range_rcp_norm", C.c_float)]


class Scene(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("frame_index", C.c_int),
                ("cam_from", C.c_float * 3), ("cam_at", C.c_float * 3), ("cam_up", C.c_float * 3),
                ("fovy", C.c_float), ("xfm", C.c_float * 12),
                ("vol_dims", C.c_int * 3), ("bbox_lo", C.c_float * 3), ("bbox_hi", C.c_float * 3),
                ("sampling_rate", C.c_float),
                ("mc_dims", C.c_int * 3), ("mc_spacings", C.c_float * 3),
                ("mc_max_opacity", C.POINTER(C.c_float)),
                ("tfn", Tfn), ("pixel_lo", C.c_uint32), ("pixel_hi", C.c_uint32),
                ("shading_mode", C.c_int), ("light_dir", C.c_float * 3), ("density_scale", C.c_float),
                ("il_block", C.c_uint32), ("il_parts", C.c_uint32), ("il_part", C.c_uint32)]


class RenderStats(C.Structure):
    _fields_ = [("n_samples", C.c_uint64), ("n_slots", C.c_uint64),
                ("n_iterations", C.c_uint32), ("n_rays_hit", C.c_uint32)]


VALUE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float))

_lib = None
_lock = threading.Lock()


def lib():
    global _lib
    with _lock:
        if _lib is None:
            build()
            L = C.CDLL(_SO)
            L.vnro_f32_to_f16.restype = C.c_uint16
            L.vnro_f32_to_f16.argtypes = [C.c_float]
            L.vnro_f16_to_f32.restype = C.c_float
            L.vnro_f16_to_f32.argtypes = [C.c_uint16]
            L.vnro_grid_make_layout.restype = C.c_uint32
            L.vnro_grid_index.restype = C.c_uint32
            L.vnro_mlp_n_params.restype = C.c_size_t
            L.vnro_mlp_n_params.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
            L.vnro_tex3d.restype = C.c_float
            L.vnro_sample_volume.restype = C.c_float
            L.vnro_lcg_next.restype = C.c_float
            L.vnro_pcg32_next_uint.restype = C.c_uint32
            L.vnro_pcg32_next_float.restype = C.c_float
            L.vnro_dda_trace.restype = C.c_size_t
            L.vnro_psnr.restype = C.c_double
            L.vnro_ooc_sample.restype = C.c_size_t
            _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------- fp16
def f32_to_f16_bits(x):
    L = lib()
    x = _f32(x).ravel()
    return np.array([L.vnro_f32_to_f16(float(v)) for v in x], dtype=np.uint16)


# --------------------------------------------------------------------------- grid
ACTIVATIONS = {"None": 0, "ReLU": 1, "Exponential": 2, "Sigmoid": 3, "Squareplus": 4, "Softplus": 5}
GRID_TYPES = {"Hash": 0, "Dense": 1, "Tiled": 2}


def act_code(activation=1, output_activation=0):
    """the `activation` argument of the C functions: hidden activation | output activation << 8 (names or codes)"""
    a = ACTIVATIONS.get(activation, activation)
    o = ACTIVATIONS.get(output_activation, output_activation)
    return int(a) | (int(o) << 8)


def grid_config(n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale=2.0,
                interpolation=0, quantize_threshold=0.0, max_level=1000.0, grid_type=0):
    """interpolation: 0 Linear, 1 Smoothstep, 2 Nearest; grid_type: 0 Hash, 1 Dense, 2 Tiled (or the names)"""
    return GridConfig(n_levels, n_features, log2_hashmap_size, base_resolution,
                      float(per_level_scale), interpolation, float(quantize_threshold), float(max_level),
                      int(GRID_TYPES.get(grid_type, grid_type)))


def grid_layout(cfg):
    lay = GridLayout()
    total = lib().vnro_grid_make_layout(C.byref(cfg), C.byref(lay))
    n = cfg.n_levels
    return {"total_entries": int(total), "offsets": np.array(lay.offsets[:n + 1], dtype=np.uint32),
            "scale": np.array(lay.scale[:n], dtype=np.float32),
            "resolution": np.array(lay.resolution[:n], dtype=np.uint32)}


def grid_index(hashmap_size, resolution, p):
    arr = (C.c_uint32 * 3)(*[int(v) & 0xFFFFFFFF for v in p])
    return int(lib().vnro_grid_index(C.c_uint32(hashmap_size), C.c_uint32(resolution), arr))


def padded_width(cfg):
    return ((cfg.n_levels * cfg.n_features + 15) // 16) * 16


def grid_encode(cfg, table_f16_bits, coords):
    """table: uint16 view of fp16 table; coords [n,3] fp32 -> uint16 [n, padded]"""
    coords = _f32(coords)
    n = coords.shape[0]
    pw = padded_width(cfg)
    out = np.zeros((n, pw), dtype=np.uint16)
    table = np.ascontiguousarray(table_f16_bits, dtype=np.uint16)
    lib().vnro_grid_encode(C.byref(cfg), _p(table, C.c_uint16), _p(coords, C.c_float), C.c_size_t(n),
                           _p(out, C.c_uint16), C.c_uint32(pw))
    return out


def mlp_n_params(in_width, width, n_hidden_matmuls):
    return int(lib().vnro_mlp_n_params(in_width, width, n_hidden_matmuls))


def mlp_forward(weights_bits, in_width, width, n_hidden_matmuls, x_bits, activation=1, acc_mode=0,
                want_activations=False):
    w = np.ascontiguousarray(weights_bits, dtype=np.uint16)
    x = np.ascontiguousarray(x_bits, dtype=np.uint16)
    n = x.shape[0]
    out = np.zeros(n, dtype=np.float32)
    act = np.zeros((n_hidden_matmuls + 1, n, width), dtype=np.uint16) if want_activations else None
    lib().vnro_mlp_forward(_p(w, C.c_uint16), C.c_uint32(in_width), C.c_uint32(width),
                           C.c_uint32(n_hidden_matmuls), C.c_int(activation), C.c_int(acc_mode),
                           _p(x, C.c_uint16), C.c_size_t(n), _p(out, C.c_float),
                           _p(act, C.c_uint16) if act is not None else None)
    return (out, act) if want_activations else out


def network_inference(cfg, width, n_hidden_layers, params_bits, coords, activation=1, acc_mode=0):
    params = np.ascontiguousarray(params_bits, dtype=np.uint16)
    coords = _f32(coords)
    n = coords.shape[0]
    out = np.zeros(n, dtype=np.float32)
    lib().vnro_network_inference(C.byref(cfg), C.c_uint32(width), C.c_uint32(n_hidden_layers),
                                 C.c_int(activation), C.c_int(acc_mode), _p(params, C.c_uint16),
                                 _p(coords, C.c_float), C.c_size_t(n), _p(out, C.c_float))
    return out


def network_inference_mt(cfg, width, n_hidden_layers, params_bits, coords, activation=1, acc_mode=0, n_threads=None):
    """network_inference over all host threads: the C function is scalar and re-entrant, ctypes releases the GIL during the call, so
    the coordinates are cut into chunks that a pool of threads pulls from (same values as one call: samples are independent)"""
    params = np.ascontiguousarray(params_bits, dtype=np.uint16)
    coords = _f32(coords)
    n = coords.shape[0]
    n_threads = n_threads or (os.cpu_count() or 1)
    if n_threads <= 1 or n < 4096:
        return network_inference(cfg, width, n_hidden_layers, params, coords, activation, acc_mode)
    out = np.zeros(n, dtype=np.float32)
    chunk = max(1024, min(65536, n // (4 * n_threads) + 1))
    starts = iter(range(0, n, chunk))
    lk = threading.Lock()
    L = lib()

    def runner():
        while True:
            with lk:
                a = next(starts, None)
            if a is None:
                return
            b = min(n, a + chunk)
            c = coords[a:b]
            o = out[a:b]
            L.vnro_network_inference(C.byref(cfg), C.c_uint32(width), C.c_uint32(n_hidden_layers), C.c_int(activation), C.c_int(acc_mode),
                                     _p(params, C.c_uint16), _p(c, C.c_float), C.c_size_t(b - a), _p(o, C.c_float))

    ts = [threading.Thread(target=runner) for _ in range(n_threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    return out


def n_params(cfg, width, n_hidden_layers):
    lay = grid_layout(cfg)
    return mlp_n_params(padded_width(cfg), width, n_hidden_layers - 1) + lay["total_entries"] * cfg.n_features


# --------------------------------------------------------------------------- volume
def _dims(d):
    return (C.c_int * 3)(int(d[0]), int(d[1]), int(d[2]))


def sample_volume(vol, coords, nodal):
    """vol: [z,y,x] fp32 (x fastest); coords [n,3] in [0,1]"""
    vol = _f32(vol)
    dims = _dims(vol.shape[::-1])
    coords = _f32(coords)
    out = np.zeros(coords.shape[0], dtype=np.float32)
    lib().vnro_sample_volume_batch(_p(vol, C.c_float), dims, _p(coords, C.c_float),
                                   C.c_size_t(coords.shape[0]), C.c_int(1 if nodal else 0), _p(out, C.c_float))
    return out


# --------------------------------------------------------------------------- tfn
class TfnHolder:
    """keeps numpy buffers alive for a vnro_tfn"""

    def __init__(self, colors_rgb, alphas, range_lo=0.0, range_hi=1.0):
        c = _f32(colors_rgb).reshape(-1, 3)
        self.colors = np.ascontiguousarray(np.concatenate([c, np.ones((c.shape[0], 1), np.float32)], axis=1))
        self.alphas = _f32(alphas).ravel().copy()
        self.c = Tfn(_p(self.colors, C.c_float), self.colors.shape[0], _p(self.alphas, C.c_float),
                     self.alphas.shape[0], range_lo, range_hi,
                     float(np.float32(1.0) / (np.float32(range_hi) - np.float32(range_lo))))


def tfn_sample(tfn, values):
    values = _f32(values).ravel()
    rgb = (C.c_float * 3)()
    a = C.c_float()
    out = np.zeros((values.shape[0], 4), dtype=np.float32)
    L = lib()
    for i, v in enumerate(values):
        L.vnro_tfn_sample(C.byref(tfn.c), C.c_float(float(v)), rgb, C.byref(a))
        out[i] = (rgb[0], rgb[1], rgb[2], a.value)
    return out


# --------------------------------------------------------------------------- macrocell
def macrocell_shape(vol_dims):
    mc = (C.c_int * 3)()
    sp = (C.c_float * 3)()
    lib().vnro_macrocell_shape(_dims(vol_dims), mc, sp)
    return tuple(mc), np.array(list(sp), dtype=np.float32)


def macrocell_compute_implicit(vol):
    vol = _f32(vol)
    vd = vol.shape[::-1]
    mc, _ = macrocell_shape(vd)
    vr = np.zeros((mc[2], mc[1], mc[0], 2), dtype=np.float32)
    lib().vnro_macrocell_compute_implicit(_p(vol, C.c_float), _dims(vd), _dims(mc), _p(vr, C.c_float))
    return vr


def macrocell_update_explicit(value_range, vol_dims, coords, values):
    coords = _f32(coords)
    values = _f32(values)
    mc = value_range.shape[:3][::-1]
    lib().vnro_macrocell_update_explicit(_p(coords, C.c_float), _p(values, C.c_float), C.c_size_t(coords.shape[0]),
                                         _dims(vol_dims), _dims(mc), _p(value_range, C.c_float))
    return value_range


def macrocell_max_opacity(tfn, value_range):
    n = value_range.size // 2
    out = np.zeros(value_range.shape[:3], dtype=np.float32)
    lib().vnro_macrocell_max_opacity(C.byref(tfn.c), _p(value_range, C.c_float), C.c_size_t(n), _p(out, C.c_float))
    return out


# --------------------------------------------------------------------------- rng
def lcg_floats(v0, v1, n):
    class Lcg(C.Structure):
        _fields_ = [("state", C.c_uint32)]
    r = Lcg()
    L = lib()
    L.vnro_lcg_init(C.byref(r), C.c_uint32(v0), C.c_uint32(v1))
    return np.array([L.vnro_lcg_next(C.byref(r)) for _ in range(n)], dtype=np.float32)


class Pcg32(C.Structure):
    _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64)]


def pcg32(initstate=0x853c49e6748fea9b, initseq=0xda3e39cb94b95bdb):
    r = Pcg32()
    lib().vnro_pcg32_seed(C.byref(r), C.c_uint64(initstate), C.c_uint64(initseq))
    return r


# --------------------------------------------------------------------------- rendering
def default_transform(dims):
    """object->world = translate(-dims/2) * scale(dims)  (ref: core/network.cu:569)"""
    d = np.asarray(dims, dtype=np.float32)
    return np.array([d[0], 0, 0, 0, d[1], 0, 0, 0, d[2], -d[0] / 2, -d[1] / 2, -d[2] / 2], dtype=np.float32)


class SceneHolder:
    def __init__(self, width, height, vol_dims, tfn, mc_max_opacity, cam_from, cam_at=(0, 0, 0), cam_up=(0, 1, 0),
                 fovy=60.0, frame_index=1, sampling_rate=1.0, bbox=((0, 0, 0), (1, 1, 1)), xfm=None,
                 pixel_range=None, shading_mode=0, light_dir=None, density_scale=1.0, interleave=None):
        """shading_mode: 0 NO_SHADING (rendering modes 4 / 5), 1 GRADIENT_SHADING (modes 7 / 8).
        light_dir: LaunchParams::light_directional_dir; default = the reference's (0.7, 0.9, 0.4) after the flip of
        renderer.cpp:98-101 (negated when it points along the view direction)."""
        self.tfn = tfn
        self.mc = np.ascontiguousarray(mc_max_opacity, dtype=np.float32)
        mc_dims, mc_sp = macrocell_shape(vol_dims)
        assert tuple(self.mc.shape[::-1]) == tuple(mc_dims), (self.mc.shape, mc_dims)
        s = Scene()
        s.width, s.height, s.frame_index = width, height, frame_index
        s.cam_from[:] = [float(v) for v in cam_from]
        s.cam_at[:] = [float(v) for v in cam_at]
        s.cam_up[:] = [float(v) for v in cam_up]
        s.fovy = fovy
        x = default_transform(vol_dims) if xfm is None else _f32(xfm)
        s.xfm[:] = [float(v) for v in x]
        s.vol_dims[:] = [int(v) for v in vol_dims]
        s.bbox_lo[:] = [float(v) for v in bbox[0]]
        s.bbox_hi[:] = [float(v) for v in bbox[1]]
        s.sampling_rate = sampling_rate
        s.mc_dims[:] = list(mc_dims)
        s.mc_spacings[:] = [float(v) for v in mc_sp]
        s.mc_max_opacity = _p(self.mc, C.c_float)
        s.tfn = tfn.c
        pr = pixel_range or (0, width * height)
        s.pixel_lo, s.pixel_hi = pr
        # interleave = (block, parts, part): a rank's share of a tile-sharded frame (streaming marcher only), global pixel indices kept
        s.il_block, s.il_parts, s.il_part = interleave or (0, 0, 0)
        s.shading_mode = int(shading_mode)
        s.density_scale = float(density_scale)
        s.light_dir[:] = [float(v) for v in (flipped_light_dir(cam_from, cam_at) if light_dir is None else light_dir)]
        self.c = s


DEFAULT_LIGHT_DIR = (0.7, 0.9, 0.4)   # LaunchParams::light_directional_dir, core/instantvnr_types.h:148


def flipped_light_dir(cam_from, cam_at, light_dir=DEFAULT_LIGHT_DIR):
    """renderer.cpp:98-101: `if (dot(camera.direction, light_dir) > 0) light_dir *= -1` (fp32).  The reference flips the
    stored member in place every frame; starting from the default that is this stateless rule except when the dot
    product is exactly zero."""
    d = _f32(cam_at) - _f32(cam_from)
    d = d / np.float32(np.sqrt(np.float32((d * d).sum(dtype=np.float32))))
    L = _f32(light_dir)
    return tuple(float(v) for v in (-L if np.float32((d * L).sum(dtype=np.float32)) > 0 else L))


def shade_scivis_light(ray_dir, normal, albedo, light_dir):
    out = (C.c_float * 3)()
    f = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    lib().vnro_shade_scivis_light(f(ray_dir), f(normal), f(albedo), f(light_dir), out)
    return np.array(list(out), dtype=np.float32)


def render_streaming(scene, value_fn, n_iters=16, accumulation=None):
    """value_fn(coords[n,3] float32) -> values[n] float32"""
    s = scene.c
    npx = s.width * s.height
    acc = np.zeros((npx, 4), dtype=np.float32) if accumulation is None else accumulation
    frame = np.zeros((npx, 4), dtype=np.float32)
    stats = RenderStats()

    def _cb(_user, cptr, n, vptr):
        coords = np.ctypeslib.as_array(cptr, shape=(n, 3))
        vals = np.ctypeslib.as_array(vptr, shape=(n,))
        vals[:] = value_fn(coords)

    cb = VALUE_FN(_cb)
    lib().vnro_render_streaming(C.byref(s), C.c_int(n_iters), cb, None, _p(acc, C.c_float), _p(frame, C.c_float),
                                C.byref(stats))
    st = {"n_samples": stats.n_samples, "n_slots": stats.n_slots, "n_iterations": stats.n_iterations,
          "n_rays_hit": stats.n_rays_hit}
    return frame.reshape(s.height, s.width, 4), acc, st


def render_pathtracing(scene, value_fn, accumulation=None):
    """rendering mode 14 (sample-streaming path tracer); value_fn(coords[n,3] float32) -> values[n] float32"""
    s = scene.c
    npx = s.width * s.height
    acc = np.zeros((npx, 4), dtype=np.float32) if accumulation is None else accumulation
    frame = np.zeros((npx, 4), dtype=np.float32)
    stats = RenderStats()

    def _cb(_user, cptr, n, vptr):
        coords = np.ctypeslib.as_array(cptr, shape=(n, 3))
        vals = np.ctypeslib.as_array(vptr, shape=(n,))
        vals[:] = value_fn(coords)

    cb = VALUE_FN(_cb)
    lib().vnro_render_pathtracing(C.byref(s), cb, None, _p(acc, C.c_float), _p(frame, C.c_float), C.byref(stats))
    st = {"n_samples": stats.n_samples, "n_slots": stats.n_slots, "n_iterations": stats.n_iterations, "n_rays_hit": stats.n_rays_hit}
    return frame.reshape(s.height, s.width, 4), acc, st


def render_pathtracing_monolithic(scene, vol, accumulation=None):
    """rendering mode 13: the path tracer in one loop per pixel on a dense volume"""
    s = scene.c
    vol = _f32(vol)
    npx = s.width * s.height
    acc = np.zeros((npx, 4), dtype=np.float32) if accumulation is None else accumulation
    frame = np.zeros((npx, 4), dtype=np.float32)
    lib().vnro_render_pathtracing_monolithic(C.byref(s), _p(vol, C.c_float), C.c_int(0), C.c_int(s.height), _p(acc, C.c_float),
                                             _p(frame, C.c_float))
    return frame.reshape(s.height, s.width, 4), acc


def render_monolithic(scene, vol, accumulation=None, n_threads=1, rows=None):
    s = scene.c
    vol = _f32(vol)
    npx = s.width * s.height
    acc = np.zeros((npx, 4), dtype=np.float32) if accumulation is None else accumulation
    frame = np.zeros((npx, 4), dtype=np.float32)
    L = lib()

    def work(lo, hi):
        L.vnro_render_monolithic(C.byref(s), _p(vol, C.c_float), C.c_int(lo), C.c_int(hi), _p(acc, C.c_float),
                                 _p(frame, C.c_float))

    if n_threads <= 1 and rows is None:
        work(0, s.height)
    else:
        # scanline blocks pulled from a shared queue; ctypes releases the GIL during the call
        if rows is None:
            rows = [(r, min(r + 4, s.height)) for r in range(0, s.height, 4)]
        else:
            rows = [(r, min(r + 2, hi)) for lo, hi in rows for r in range(lo, hi, 2)]
        it = iter(rows)
        lk = threading.Lock()

        def runner():
            while True:
                with lk:
                    rg = next(it, None)
                if rg is None:
                    return
                work(*rg)

        ts = [threading.Thread(target=runner) for _ in range(max(1, n_threads))]
        [t.start() for t in ts]
        [t.join() for t in ts]
    return frame.reshape(s.height, s.width, 4), acc


def dda_trace(org, dir_, t_min, t_max, grid, max_cells=4096):
    cells = np.zeros((max_cells, 3), dtype=np.int32)
    ts = np.zeros((max_cells, 2), dtype=np.float32)
    o = (C.c_float * 3)(*[float(v) for v in org])
    d = (C.c_float * 3)(*[float(v) for v in dir_])
    n = lib().vnro_dda_trace(o, d, C.c_float(t_min), C.c_float(t_max), _dims(grid), _p(cells, C.c_int),
                             _p(ts, C.c_float), C.c_size_t(max_cells))
    n = min(int(n), max_cells)
    return cells[:n], ts[:n]


# --------------------------------------------------------------------------- metrics
def grid_coords(lower, size, rdims):
    n = int(size[0]) * int(size[1]) * int(size[2])
    out = np.zeros((n, 3), dtype=np.float32)
    r = (C.c_float * 3)(*[float(v) for v in rdims])
    lib().vnro_generate_grid_coords(_dims(lower), _dims(size), r, _p(out, C.c_float))
    return out


def psnr(pred, ref, ref_min=None, ref_max=None):
    pred = _f32(pred).ravel()
    ref = _f32(ref).ravel()
    lo = float(ref.min()) if ref_min is None else ref_min
    hi = float(ref.max()) if ref_max is None else ref_max
    return float(lib().vnro_psnr(_p(pred, C.c_float), _p(ref, C.c_float), C.c_size_t(pred.size), C.c_float(lo),
                                 C.c_float(hi)))


def mssim(pred, ref, data_range=1.0, win=7, k1=0.01, k2=0.03):
    """get_mssim / compute_ssim<7> (core/network.cu:474-549, :70-127): mean SSIM of `pred` against `ref` (volumes [z, y, x]
    sampled at the voxel centres) over the interior, one win^3 UNIFORM window per voxel, sample covariance
    (cov_norm = NP / (NP - 1)), C1 = (k1 R)^2, C2 = (k2 R)^2.  float64 (the reference sums the fp32 map in fp32)."""
    x = np.asarray(ref, dtype=np.float64)
    y = np.asarray(pred, dtype=np.float64)
    assert x.shape == y.shape and min(x.shape) >= win

    def box(a):  # mean over every win^3 window ("valid"), via cumulative sums along each axis
        for ax in range(3):
            c = np.cumsum(a, axis=ax)
            c = np.concatenate([np.zeros_like(np.take(c, [0], axis=ax)), c], axis=ax)
            hi = np.take(c, np.arange(win, c.shape[ax]), axis=ax)
            lo = np.take(c, np.arange(0, c.shape[ax] - win), axis=ax)
            a = (hi - lo) / win
        return a

    ux, uy, uxx, uyy, uxy = box(x), box(y), box(x * x), box(y * y), box(x * y)
    n_p = win ** 3
    cov_norm = n_p / (n_p - 1.0)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return float(s.mean())


# --------------------------------------------------------------------------- out-of-core training sampler
OOC_VALUE_TYPES = {np.dtype(np.uint8): 0, np.dtype(np.int8): 1, np.dtype(np.uint16): 2, np.dtype(np.int16): 3,
                   np.dtype(np.uint32): 4, np.dtype(np.int32): 5, np.dtype(np.float32): 8, np.dtype(np.float64): 12}


class OocGeometry(C.Structure):
    _fields_ = [("dims", C.c_int * 3), ("type", C.c_int), ("elem", C.c_uint32), ("block_dims", C.c_int * 3),
                ("ghost_dims", C.c_int * 3), ("index_space", C.c_int * 3), ("block_size_aligned", C.c_uint64)]


class OocBlock(C.Structure):
    _fields_ = [("index", C.c_int * 3), ("offset", C.c_uint64), ("length", C.c_uint64), ("bounds_lo", C.c_int * 3),
                ("bounds_hi", C.c_int * 3), ("ghost_lo", C.c_int * 3), ("ghost_hi", C.c_int * 3)]


def ooc_geometry(dims, dtype):
    g = OocGeometry()
    if lib().vnro_ooc_geometry_make(_dims(dims), C.c_int(OOC_VALUE_TYPES[np.dtype(dtype)]), C.byref(g)) != 0:
        raise ValueError("unsupported data type")
    return g


class OocSlabSet:
    """the resident slab set of RandomBuffer for a volume held in memory: `vol` is [z, y, x] in its file type,
    `block_index_yz` one (y, z) block index per slot"""

    def __init__(self, vol, block_index_yz):
        self.vol = np.ascontiguousarray(vol)
        self.g = ooc_geometry(self.vol.shape[::-1], self.vol.dtype)
        idx = np.asarray(block_index_yz, dtype=np.int64).reshape(-1, 2)
        self.n = idx.shape[0]
        self.blocks = (OocBlock * self.n)()
        self.data = np.zeros(self.n * self.g.block_size_aligned, np.uint8)
        file_p = self.vol.ctypes.data_as(C.POINTER(C.c_uint8))
        for i in range(self.n):
            bi = (C.c_int * 3)(0, int(idx[i, 0]), int(idx[i, 1]))
            dst = self.data[i * self.g.block_size_aligned:]
            if lib().vnro_ooc_load_block(C.byref(self.g), file_p, bi, C.byref(self.blocks[i]), _p(dst, C.c_uint8)) != 0:
                raise ValueError(f"[aio] invalid block index {tuple(idx[i])}")

    def sample(self, value_range, r_coords, r_bidx, r_vidx, lower=(0, 0, 0), upper=(1, 1, 1)):
        rc, rb, rv = _f32(r_coords).reshape(-1), _f32(r_bidx), _f32(r_vidx)
        n = rb.shape[0]
        coords = np.zeros((n, 3), np.float32)
        values = np.zeros(n, np.float32)
        lo3 = (C.c_float * 3)(*[float(v) for v in lower])
        hi3 = (C.c_float * 3)(*[float(v) for v in upper])
        bad = lib().vnro_ooc_sample(C.byref(self.g), self.blocks, _p(self.data, C.c_uint8), C.c_uint64(self.n),
                                    C.c_float(value_range[0]), C.c_float(value_range[1]), _p(rc, C.c_float), _p(rb, C.c_float),
                                    _p(rv, C.c_float), C.c_size_t(n), lo3, hi3, _p(coords, C.c_float), _p(values, C.c_float))
        return coords, values, int(bad)


def ooc_sample_grid(vol, value_range, origin, size, spacing):
    vol = np.ascontiguousarray(vol)
    g = ooc_geometry(vol.shape[::-1], vol.dtype)
    n = int(size[0]) * int(size[1]) * int(size[2])
    values = np.zeros(n, np.float32)
    lib().vnro_ooc_sample_grid(C.byref(g), vol.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_float(value_range[0]),
                               C.c_float(value_range[1]), _dims(origin), _dims(size),
                               (C.c_float * 3)(*[float(v) for v in spacing]), _p(values, C.c_float))
    return values


def pcg32_floats(n, offset=0, initstate=1337, initseq=0xda3e39cb94b95bdb):
    """n consecutive floats of the sampler's pcg32 stream starting at stream position `offset`"""
    r = pcg32(initstate, initseq)
    lib().vnro_pcg32_advance(C.byref(r), C.c_int64(offset))
    out = np.zeros(n, np.float32)
    L = lib()
    for i in range(n):
        out[i] = L.vnro_pcg32_next_float(C.byref(r))
    return out
