"""ctypes binding of libraymarching_amd.so (include/raymarching_amd.h).

This is a driver for tests and bench.py, not a renderer: every call goes
straight through the C ABI to the HIP kernels.  There is no CPU fallback; if the
shared library is missing or no HIP device is present the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RM_LIB_PATH") or os.path.join(_HERE, "libraymarching_amd.so")      # RM_LIB_PATH: A/B runs of two builds on one box

NODE_DTYPE = np.dtype([("type", "<u4"), ("o", "<i4", (4,)), ("f", "<f4", (5,))])
POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
INFO_DTYPE = np.dtype([("Depth", "<u8"), ("LightSource", POINT), ("StartPos", POINT),
                       ("StartDir", POINT), ("StartWVec", POINT), ("StartHVec", POINT),
                       ("_pad", "<u4")])
assert NODE_DTYPE.itemsize == 40 and INFO_DTYPE.itemsize == 72

BAZO_LP64, BAZO_LLP64, BAZO_SCE = 0, 1, 2

# every symbol include/raymarching_amd.h declares
EXPORTS = [
    "rm_decode_bazo", "rm_load_sce", "rm_save_sce", "rm_create", "rm_destroy", "rm_set_stream",
    "rm_set_output", "rm_set_primitives", "rm_set_nodes", "rm_set_info", "rm_set_bands", "rm_set_band_pattern",
    "rm_local_rows", "rm_enable_hit_mask", "rm_process", "rm_process_async", "rm_synchronize",
    "rm_read_pixels", "rm_read_hit_mask", "rm_device_framebuffer", "rm_get_scene_stats",
    "rm_flatten_scene", "rm_set_evaluator", "rm_status_string",
    "rm_eval_points", "rm_eval_rays", "rm_debug_wave_stats", "rm_set_boost_age", "rm_set_heavy_first", "rm_selftest_math", "rm_set_short_math", "rm_set_culling", "rm_set_jit", "rm_jit_log", "rm_jit_check",
    "rm_set_luminance_output", "rm_assemble_luminance", "rm_assemble_luminance_batch", "rm_set_batch", "rm_set_info_batch",
    "rm_process_static", "rm_debug_set_watchdog", "rm_scene_bound_of", "rm_set_ray_windows", "rm_read_pixels_of_frame", "rm_debug_last_check",
]


class SceneStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("nodes", "reachable", "program_slots", "value_slots",
                                          "tree_depth", "lds_bytes", "kernel_vgprs", "kernel_scratch", "jit",
                                          "exact_divisors")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class SceneBound(C.Structure):
    """rm_scene_bound (include/rm_types.h)."""
    _fields_ = [("kind", C.c_uint32), ("c", C.c_float * 3), ("u", C.c_float * 3), ("r", C.c_float)]

    def as_dict(self):
        return {"kind": int(self.kind), "c": [float(v) for v in self.c], "u": [float(v) for v in self.u], "r": float(self.r)}


BOUND_NONE, BOUND_SPHERE, BOUND_CYLINDER = 0, 1, 2


class RaymarchError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__("%s failed: %d (%s)" % (what, status, status_string(status)))


_lib = None


def lib():
    """Load the C-ABI library; raises OSError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
        # PyTorch wheels bundle their own libamdhip64.so with the same SONAME
        # (libamdhip64.so.7) as /opt/rocm's.  Two HIP runtimes in one process do not
        # coexist (the second one finds no GPU), so when torch is installed let it
        # load its runtime first: the loader then binds this library to that copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        sig = {
            "rm_decode_bazo": [vp, sz, i32, vp],
            "rm_load_sce": [C.c_char_p, vp, sz, C.POINTER(sz)],
            "rm_save_sce": [C.c_char_p, vp, sz],
            "rm_create": [i32, sz, sz, C.POINTER(vp)],
            "rm_destroy": [vp],
            "rm_set_stream": [vp, vp],
            "rm_set_output": [vp, vp, sz],
            "rm_set_luminance_output": [vp, vp, sz],
            "rm_set_batch": [vp, sz, sz, sz],
            "rm_set_info_batch": [vp, vp, sz],
            "rm_assemble_luminance": [vp, vp, sz, sz, vp, sz, vp],
            "rm_assemble_luminance_batch": [vp, vp, sz, sz, sz, sz, vp, sz, sz, vp],
            "rm_set_primitives": [vp, vp, sz, i32],
            "rm_set_nodes": [vp, vp, sz],
            "rm_set_info": [vp, vp],
            "rm_set_bands": [vp, sz, i32, i32],
            "rm_set_band_pattern": [vp, sz, vp, sz, i32, i32],
            "rm_local_rows": [vp, C.POINTER(sz)],
            "rm_enable_hit_mask": [vp, i32],
            "rm_process": [vp, C.POINTER(C.c_float)],
            "rm_process_static": [vp, vp, vp, sz, C.POINTER(C.c_float)],
            "rm_process_async": [vp],
            "rm_synchronize": [vp],
            "rm_read_pixels": [vp, vp, sz],
            "rm_read_hit_mask": [vp, vp, sz],
            "rm_read_pixels_of_frame": [vp, sz, vp, sz],
            "rm_device_framebuffer": [vp, C.POINTER(vp), C.POINTER(sz)],
            "rm_get_scene_stats": [vp, C.POINTER(SceneStats)],
            "rm_flatten_scene": [vp, sz, C.POINTER(SceneStats), C.c_char_p, sz],
            "rm_set_evaluator": [vp, i32],
            "rm_set_boost_age": [vp, C.c_uint],
            "rm_debug_set_watchdog": [vp, C.c_ulonglong],
            "rm_debug_last_check": [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)],
            "rm_set_heavy_first": [vp, i32],
            "rm_selftest_math": [vp, vp, sz, C.POINTER(C.c_ulonglong), vp],
            "rm_set_short_math": [vp, i32],
            "rm_set_culling": [vp, i32],
            "rm_set_ray_windows": [vp, i32],
            "rm_scene_bound_of": [vp, sz, C.POINTER(SceneBound)],
            "rm_eval_points": [vp, i32, sz, vp, sz, vp, vp],
            "rm_eval_rays": [vp, vp, sz, vp],
            "rm_debug_wave_stats": [vp, vp, sz, C.POINTER(sz)],
            "rm_set_jit": [vp, i32],
            "rm_jit_log": [vp, C.c_char_p, sz],
            "rm_jit_check": [vp, sz, C.c_char_p, i32, C.c_char_p, sz],
        }
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = i32
        L.rm_status_string.argtypes = [i32]
        L.rm_status_string.restype = C.c_char_p
        _lib = L
    return _lib


def status_string(status):
    return lib().rm_status_string(int(status)).decode()


def _check(status, what):
    if status != 0:
        raise RaymarchError(status, what)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def decode_bazo(raw, layout=BAZO_SCE):
    raw = np.frombuffer(bytes(raw), dtype=np.uint8)
    if raw.size == 0 or raw.size % 48:
        raise ValueError("a bazo table is a whole number of 48-byte records")
    nodes = np.zeros(raw.size // 48, dtype=NODE_DTYPE)
    _check(lib().rm_decode_bazo(_ptr(raw), nodes.size, layout, _ptr(nodes)), "rm_decode_bazo")
    return nodes


def load_sce(path):
    n = C.c_size_t(0)
    _check(lib().rm_load_sce(os.fsencode(path), None, 0, C.byref(n)), "rm_load_sce")
    nodes = np.zeros(n.value, dtype=NODE_DTYPE)
    _check(lib().rm_load_sce(os.fsencode(path), _ptr(nodes), nodes.size, C.byref(n)), "rm_load_sce")
    return nodes


def save_sce(path, nodes):
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    _check(lib().rm_save_sce(os.fsencode(path), _ptr(nodes), nodes.size), "rm_save_sce")


def flatten_scene(nodes, listing_cap=1 << 20):
    """Host-only flattening: returns (stats dict, program listing)."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    st = SceneStats()
    buf = C.create_string_buffer(listing_cap)
    _check(lib().rm_flatten_scene(_ptr(nodes), nodes.size, C.byref(st), buf, listing_cap), "rm_flatten_scene")
    return st.as_dict(), buf.value.decode()


def scene_bound(nodes):
    """Host-only: the volume outside of which the scene's distance function is provably large (rm_scene_bound_of)."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    b = SceneBound()
    _check(lib().rm_scene_bound_of(_ptr(nodes), nodes.size, C.byref(b)), "rm_scene_bound_of")
    return b


def jit_check(nodes, arch="gfx950", want_source=False, cap=1 << 20, assume_exact_divisors=False):
    """Host-only: generate + compile a scene's code with hiprtc.  Returns (status, source or log)."""
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    buf = C.create_string_buffer(cap)
    rc = lib().rm_jit_check(_ptr(nodes), nodes.size, arch.encode(), (1 if want_source else 0) | (2 if assume_exact_divisors else 0), buf, cap)
    return rc, buf.value.decode(errors="replace")


class Raymarching:
    """One rm_ctx: the counterpart of `cuda::raymarching` (Include/cuda_rays.cuh:14-46)."""

    def __init__(self, width, height, device=0):
        self.width, self.height, self.device = int(width), int(height), int(device)
        self._ctx = C.c_void_p()
        _check(lib().rm_create(self.device, self.width, self.height, C.byref(self._ctx)), "rm_create")

    def close(self):
        if self._ctx:
            lib().rm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene / camera
    def set_nodes(self, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        _check(lib().rm_set_nodes(self._ctx, _ptr(nodes), nodes.size), "rm_set_nodes")

    def set_primitives(self, raw, layout=BAZO_SCE):
        raw = np.frombuffer(bytes(raw), dtype=np.uint8)
        _check(lib().rm_set_primitives(self._ctx, _ptr(raw), raw.size // 48, layout), "rm_set_primitives")

    def set_info(self, info):
        info = np.ascontiguousarray(info, dtype=INFO_DTYPE)
        _check(lib().rm_set_info(self._ctx, _ptr(info)), "rm_set_info")

    # -- configuration
    def set_stream(self, stream_handle):
        _check(lib().rm_set_stream(self._ctx, C.c_void_p(stream_handle)), "rm_set_stream")

    def set_output(self, device_ptr, pitch_bytes):
        _check(lib().rm_set_output(self._ctx, C.c_void_p(device_ptr), pitch_bytes), "rm_set_output")

    def set_batch(self, frames, rgba_frame_stride=0, luminance_frame_stride=0):
        """`frames` frames per launch, their tiles in one pool (rm_set_batch)."""
        _check(lib().rm_set_batch(self._ctx, frames, rgba_frame_stride, luminance_frame_stride), "rm_set_batch")

    def set_info_batch(self, infos):
        """One camera block per frame of the batch (an INFO_DTYPE array)."""
        infos = np.ascontiguousarray(infos, dtype=INFO_DTYPE)
        _check(lib().rm_set_info_batch(self._ctx, infos.ctypes.data_as(C.c_void_p), infos.size), "rm_set_info_batch")

    def set_luminance_output(self, device_ptr, pitch_bytes):
        """One byte per pixel besides the RGBA8 frame (a quarter of the bytes to gather); 0/None switches it off."""
        _check(lib().rm_set_luminance_output(self._ctx, C.c_void_p(device_ptr or None), pitch_bytes), "rm_set_luminance_output")

    def assemble_luminance(self, parts_ptr, rank_stride, part_pitch, rgba_ptr, pitch_bytes, stream=None):
        """RGBA8 frame from every rank's luminance plane (this context's image size and band layout)."""
        _check(lib().rm_assemble_luminance(self._ctx, C.c_void_p(parts_ptr), rank_stride, part_pitch, C.c_void_p(rgba_ptr),
                                           pitch_bytes, C.c_void_p(stream or None)), "rm_assemble_luminance")

    def set_bands(self, band_rows, rank, nranks, owner=None):
        """owner: the rank of every band of a run (rm_set_band_pattern); None: band b to rank b % nranks."""
        if owner is None:
            _check(lib().rm_set_bands(self._ctx, band_rows, rank, nranks), "rm_set_bands")
        else:
            table = (C.c_ubyte * len(owner))(*owner)
            _check(lib().rm_set_band_pattern(self._ctx, band_rows, table, len(owner), rank, nranks), "rm_set_band_pattern")

    def assemble_luminance_batch(self, parts_ptr, rank_stride, part_frame_stride, part_pitch, frames, rgba_ptr, pitch_bytes,
                                 rgba_frame_stride, stream=None):
        """The frames of a launch rebuilt in one kernel (frame f: part_frame_stride * f into every part,
        rgba_frame_stride * f into the output)."""
        _check(lib().rm_assemble_luminance_batch(self._ctx, C.c_void_p(parts_ptr), rank_stride, part_frame_stride, part_pitch, frames,
                                                 C.c_void_p(rgba_ptr), pitch_bytes, rgba_frame_stride, C.c_void_p(stream or None)),
               "rm_assemble_luminance_batch")

    def local_rows(self):
        n = C.c_size_t(0)
        _check(lib().rm_local_rows(self._ctx, C.byref(n)), "rm_local_rows")
        return n.value

    def enable_hit_mask(self, enable=True):
        _check(lib().rm_enable_hit_mask(self._ctx, 1 if enable else 0), "rm_enable_hit_mask")

    def set_jit(self, mode):
        _check(lib().rm_set_jit(self._ctx, mode), "rm_set_jit")

    def jit_log(self):
        buf = C.create_string_buffer(1 << 16)
        _check(lib().rm_jit_log(self._ctx, buf, 1 << 16), "rm_jit_log")
        return buf.value.decode(errors="replace")

    def set_heavy_first(self, enable=True):
        _check(lib().rm_set_heavy_first(self._ctx, 1 if enable else 0), "rm_set_heavy_first")

    def selftest_math(self, divisors=()):
        """(sqrt mismatches, [division mismatches per divisor]) over all 2^32 bit patterns."""
        d = np.ascontiguousarray(divisors, dtype=np.float32)
        sq = C.c_ulonglong(0)
        bad = np.zeros(max(1, d.size), dtype=np.uint64)
        _check(lib().rm_selftest_math(self._ctx, _ptr(d) if d.size else None, d.size, C.byref(sq), _ptr(bad)),
               "rm_selftest_math")
        return sq.value, [int(x) for x in bad[:d.size]]

    def set_culling(self, enable=True):
        _check(lib().rm_set_culling(self._ctx, 1 if enable else 0), "rm_set_culling")

    def set_ray_windows(self, enable=True):
        _check(lib().rm_set_ray_windows(self._ctx, 1 if enable else 0), "rm_set_ray_windows")

    def set_short_math(self, enable=True):
        _check(lib().rm_set_short_math(self._ctx, 1 if enable else 0), "rm_set_short_math")

    def set_boost_age(self, trips):
        _check(lib().rm_set_boost_age(self._ctx, trips), "rm_set_boost_age")

    def set_watchdog(self, ticks):
        _check(lib().rm_debug_set_watchdog(self._ctx, ticks), "rm_debug_set_watchdog")

    def last_check(self):
        """(tiles reported finished, tiles launched, watchdog flag) of the context's last completion check."""
        d, e, f = C.c_uint(0), C.c_uint(0), C.c_uint(0)
        _check(lib().rm_debug_last_check(self._ctx, C.byref(d), C.byref(e), C.byref(f)), "rm_debug_last_check")
        return d.value, e.value, f.value

    def set_evaluator(self, which):
        _check(lib().rm_set_evaluator(self._ctx, which), "rm_set_evaluator")

    def scene_stats(self):
        st = SceneStats()
        _check(lib().rm_get_scene_stats(self._ctx, C.byref(st)), "rm_get_scene_stats")
        return st.as_dict()

    # -- rendering
    def process(self):
        """Render one frame and wait; returns the kernel time in ms."""
        ms = C.c_float(0)
        _check(lib().rm_process(self._ctx, C.byref(ms)), "rm_process")
        return ms.value

    def process_static(self, info, device_ptr, pitch_bytes):
        """One frame into a caller-owned device buffer through the entry point that allocates nothing; kernel ms."""
        info = np.ascontiguousarray(info, dtype=INFO_DTYPE)
        ms = C.c_float(0)
        _check(lib().rm_process_static(self._ctx, _ptr(info), C.c_void_p(device_ptr), pitch_bytes, C.byref(ms)), "rm_process_static")
        return ms.value

    def process_async(self):
        _check(lib().rm_process_async(self._ctx), "rm_process_async")

    def synchronize(self):
        _check(lib().rm_synchronize(self._ctx), "rm_synchronize")

    def read_pixels(self):
        rows = self.local_rows()
        out = np.zeros((rows, self.width, 4), dtype=np.uint8)
        _check(lib().rm_read_pixels(self._ctx, _ptr(out), self.width * 4), "rm_read_pixels")
        return out

    def read_pixels_of_frame(self, frame):
        """Frame `frame` of the last launch of several frames (rm_set_batch), from whichever planes it rendered into."""
        out = np.zeros((self.local_rows(), self.width, 4), dtype=np.uint8)
        _check(lib().rm_read_pixels_of_frame(self._ctx, frame, _ptr(out), self.width * 4), "rm_read_pixels_of_frame")
        return out

    def read_hit_mask(self):
        rows = self.local_rows()
        out = np.zeros((rows, self.width), dtype=np.uint8)
        _check(lib().rm_read_hit_mask(self._ctx, _ptr(out), self.width), "rm_read_hit_mask")
        return out

    def device_framebuffer(self):
        p, pitch = C.c_void_p(), C.c_size_t(0)
        _check(lib().rm_device_framebuffer(self._ctx, C.byref(p), C.byref(pitch)), "rm_device_framebuffer")
        return p.value, pitch.value

    def wave_stats(self):
        """[waves, 8] uint32: loop trips, normal passes, ticks, ticks in normal passes, start, end (clock low words), tiles, longest tile's trips."""
        n = C.c_size_t(0)
        _check(lib().rm_debug_wave_stats(self._ctx, None, 0, C.byref(n)), "rm_debug_wave_stats")
        out = np.zeros((n.value, 8), dtype=np.uint32)
        _check(lib().rm_debug_wave_stats(self._ctx, _ptr(out), n.value, C.byref(n)), "rm_debug_wave_stats")
        return out

    # -- known-answer hooks
    def eval_points(self, pts, idx=0, which=1, want_norm=True):
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
        dist = np.zeros(len(pts), dtype=np.float32)
        norm = np.zeros((len(pts), 3), dtype=np.float32) if want_norm else None
        _check(lib().rm_eval_points(self._ctx, which, idx, _ptr(pts), len(pts), _ptr(dist), _ptr(norm)),
               "rm_eval_points")
        return dist, norm

    def eval_rays(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
        out = np.zeros((len(xy), 6), dtype=np.float32)
        _check(lib().rm_eval_rays(self._ctx, _ptr(xy), len(xy), _ptr(out)), "rm_eval_rays")
        return out
