"""ctypes binding of the CPU restatement (oracle/rm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (raytracing_amd/) must never
import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librm_oracle.so")

NODE_DTYPE = np.dtype([("type", "<u4"), ("o", "<i4", (4,)), ("f", "<f4", (5,))])
assert NODE_DTYPE.itemsize == 40

POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
INFO_DTYPE = np.dtype([("Depth", "<u8"), ("LightSource", POINT), ("StartPos", POINT),
                       ("StartDir", POINT), ("StartWVec", POINT), ("StartHVec", POINT),
                       ("_pad", "<u4")])
assert INFO_DTYPE.itemsize == 72

NTYPES = 21
COUNTER_FIELDS = (["dist_calls"] * NTYPES + ["norm_calls"] * NTYPES +
                  ["root_dist_primary", "root_dist_shadow", "root_norm", "hits",
                   "lum_out_of_range", "hit_colour_disagree", "flops"])
TYPE_NAMES = ["portanta_sfero", "sfero", "kubo", "cilindro", "ebeno",
              "kunigajo_2", "kunigajo_3", "kunigajo_4", "komunajo_2", "komunajo_3", "komunajo_4",
              "komplemento", "glata_kunigajo_2", "glata_komunajo_2",
              "movo", "rotacioX", "rotacioY", "rotacioZ", "rotacioQ", "senfina_ripeto"]

BAZO_LP64, BAZO_LLP64, BAZO_SCE = 0, 1, 2


class Point(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Ray(C.Structure):
    _fields_ = [("p", Point), ("d", Point)]


def build(force=False):
    """Compile librm_oracle.so (gcc -O2 -ffp-contract=off)."""
    src = os.path.join(_HERE, "rm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.rmo_decode_bazo.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.rmo_decode_bazo.restype = C.c_int
        L.rmo_encode_sce.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.rmo_encode_sce.restype = None
        L.rmo_validate.argtypes = [C.c_void_p, C.c_size_t]
        L.rmo_validate.restype = C.c_int
        L.rmo_dist.argtypes = [C.c_void_p, C.c_size_t, Point, C.c_void_p]
        L.rmo_dist.restype = C.c_float
        L.rmo_norm.argtypes = [C.c_void_p, C.c_size_t, Point, C.c_void_p]
        L.rmo_norm.restype = Point
        L.rmo_set_ray.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_longlong, C.c_longlong]
        L.rmo_set_ray.restype = Ray
        L.rmo_render_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                      C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rmo_render_rows.restype = None
        L.rmo_render.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rmo_render.restype = C.c_double
        L.rmo_render_rows_windowed.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                               C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rmo_render_rows_windowed.restype = None
        L.rmo_render_windowed.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rmo_render_windowed.restype = C.c_double
        L.rmo_default_info.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rmo_default_info.restype = None
        L.rmo_camera_script.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.rmo_camera_script.restype = C.c_size_t
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def decode_bazo(raw, layout=BAZO_SCE):
    raw = np.frombuffer(bytes(raw), dtype=np.uint8)
    assert raw.size % 48 == 0, "a bazo table is a whole number of 48-byte records"
    n = raw.size // 48
    nodes = np.zeros(n, dtype=NODE_DTYPE)
    rc = lib().rmo_decode_bazo(_ptr(raw), n, layout, _ptr(nodes))
    if rc != 0:
        raise ValueError("rmo_decode_bazo failed: %d" % rc)
    return nodes


def load_sce(path):
    with open(path, "rb") as f:
        return decode_bazo(f.read(), BAZO_SCE)


def encode_sce(nodes):
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    out = np.zeros(nodes.size * 48, dtype=np.uint8)
    lib().rmo_encode_sce(_ptr(nodes), nodes.size, _ptr(out))
    return out.tobytes()


def validate(nodes):
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    return lib().rmo_validate(_ptr(nodes), nodes.size)


def default_info(light_steps=1, phi_steps=0):
    """main.cpp default camera; light_steps=1 is 'frame 0' of SURVEY.md 8d."""
    info = np.zeros(1, dtype=INFO_DTYPE)
    lib().rmo_default_info(_ptr(info), light_steps, phi_steps)
    return info


EVENTS = {"frame": 0, "left": 1, "right": 2, "space": 3, "pause": 4, "resume": 5, "wheel": 6, "pan": 7, "drag": 8}


def camera_script(lines):
    """The rays_info of every frame of a scripted run of the reference's event loop
    (rmo_camera_script; main.cpp:144-270).  `lines`: the text of an rm_render --script file,
    one event per line: frame | left | right | space | pause | resume | wheel N | pan DX DY | drag DX DY."""
    ev = []
    for ln in lines:
        tok = ln.split("#")[0].split()
        if not tok:
            continue
        args = [int(v) for v in tok[1:]] + [0, 0]
        ev.append((EVENTS[tok[0]], args[0], args[1]))
    ev = np.ascontiguousarray(ev, dtype=np.int32).reshape(-1, 3)
    frames = sum(1 for e in ev if e[0] == 0)
    out = np.zeros(max(frames, 1), dtype=INFO_DTYPE)
    got = lib().rmo_camera_script(_ptr(ev), len(ev), _ptr(out), frames)
    assert got == frames
    return out[:frames]


def dist(nodes, idx, p):
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    return float(lib().rmo_dist(_ptr(nodes), idx, Point(*[float(v) for v in p]), None))


def norm(nodes, idx, p):
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    r = lib().rmo_norm(_ptr(nodes), idx, Point(*[float(v) for v in p]), None)
    return (r.x, r.y, r.z)


def dist_many(nodes, idx, pts):
    """float32 bit patterns of dist at each point (pts: [m,3] float32)."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty(len(pts), dtype=np.float32)
    L = lib()
    for i, p in enumerate(pts):
        out[i] = L.rmo_dist(_ptr(nodes), idx, Point(float(p[0]), float(p[1]), float(p[2])), None)
    return out


def norm_many(nodes, idx, pts):
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty((len(pts), 3), dtype=np.float32)
    L = lib()
    for i, p in enumerate(pts):
        r = L.rmo_norm(_ptr(nodes), idx, Point(float(p[0]), float(p[1]), float(p[2])), None)
        out[i] = (r.x, r.y, r.z)
    return out


def set_ray(width, height, info, x, y):
    r = lib().rmo_set_ray(width, height, _ptr(info), x, y)
    return np.array([r.p.x, r.p.y, r.p.z, r.d.x, r.d.y, r.d.z], dtype=np.float32)


def counters_dict(c):
    d = {"dist_calls": {}, "norm_calls": {}}
    for i, name in enumerate(TYPE_NAMES):
        if c[i]:
            d["dist_calls"][name] = int(c[i])
        if c[NTYPES + i]:
            d["norm_calls"][name] = int(c[NTYPES + i])
    for k, name in enumerate(COUNTER_FIELDS[2 * NTYPES:]):
        d[name] = int(c[2 * NTYPES + k])
    return d


BOUND_DTYPE = np.dtype([("kind", "<u4"), ("c", "<f4", (3,)), ("u", "<f4", (3,)), ("r", "<f4")])   # rm_scene_bound
assert BOUND_DTYPE.itemsize == 32


def bound_record(kind, c=(0, 0, 0), u=(0, 0, 0), r=0.0):
    b = np.zeros(1, dtype=BOUND_DTYPE)
    b["kind"], b["c"], b["u"], b["r"] = kind, c, u, r
    return b


def render(nodes, width, height, info, threads=1, want_hit=True, want_evals=False,
           want_counters=False, rows=None, window=None):
    """Render a frame (or rows=(y0,y1) of it, single-threaded) on the CPU.

    window: an rm_scene_bound record (BOUND_DTYPE) - the product's march restated (rmo_render_windowed: primary
    rays marched only inside their ray window, hits that face away from the light shaded without their shadow
    march; same pixels, fewer evaluations); None: the reference's march.
    Returns dict(rgba[H,W,4] u8, hit[H,W] u8, evals[H,W] u16, counters, seconds)."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    info = np.ascontiguousarray(info, dtype=INFO_DTYPE)
    if validate(nodes) != 0:
        raise ValueError("scene failed validation")
    rgba = np.zeros((height, width, 4), dtype=np.uint8)
    hit = np.zeros((height, width), dtype=np.uint8) if want_hit else None
    evals = np.zeros((height, width), dtype=np.uint16) if want_evals else None
    cnt = np.zeros(len(COUNTER_FIELDS), dtype=np.uint64) if want_counters else None
    if window is not None:
        window = np.ascontiguousarray(window, dtype=BOUND_DTYPE)
    if rows is None:
        secs = lib().rmo_render_windowed(_ptr(nodes), nodes.size, width, height, _ptr(info), _ptr(window), threads,
                                         _ptr(rgba), _ptr(hit), _ptr(evals), _ptr(cnt))
    else:
        import time
        t0 = time.perf_counter()
        lib().rmo_render_rows_windowed(_ptr(nodes), nodes.size, width, height, _ptr(info), _ptr(window), rows[0], rows[1],
                                       _ptr(rgba), _ptr(hit), _ptr(evals), _ptr(cnt))
        secs = time.perf_counter() - t0
    return {"rgba": rgba, "hit": hit, "evals": evals,
            "counters": counters_dict(cnt) if cnt is not None else None, "seconds": secs}


def ppm_bytes(rgba):
    """P6 PPM of the RGB channels (the CPU-only config 1 dump)."""
    h, w = rgba.shape[:2]
    return b"P6\n%d %d\n255\n" % (w, h) + rgba[:, :, :3].tobytes()
