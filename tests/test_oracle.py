"""CPU suite: the restatement (oracle/) against the outputs of the reference's own
march recorded in BASELINE.md section 2, and against the committed fixtures."""
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

import kat_scenes
from oracle import oracle as O

# BASELINE.md section 2 (reference's kernel source run host-side during the survey)
REFERENCE_PPM_SHA256 = {
    "chair": "f5c226d07e78e48b447e0456fc75922d8fe8f0794e9569cfc7b5efabfea489ec",
    "infinity": "369094d61c29b3a505f77cde384c40af3c8b427acc609ecc5ca01086ac6a59e8",
}
REFERENCE_HITS = {("chair", 640, 480): 259549, ("infinity", 640, 480): 307136,
                  ("chair", 1920, 1080): 296418, ("infinity", 1920, 1080): 2073420}
REFERENCE_MEAN_ROOT_EVALS = {("chair", 640, 480): 34.2, ("infinity", 640, 480): 58.6,
                             ("chair", 1920, 1080): 16.9, ("infinity", 1920, 1080): 56.2}
REFERENCE_NODE_CALLS_PER_PIXEL = {("chair", 640, 480): 385.6, ("infinity", 640, 480): 293.2,
                                  ("chair", 1920, 1080): 188.0, ("infinity", 1920, 1080): 281.1}
THREADS = os.cpu_count() or 4


@pytest.mark.parametrize("name", ["chair", "infinity"])
def test_pinned_by_reference_frame_640x480(stock_scenes, name):
    """config 1: the PPM of the 640x480 frame hashes to the reference's own output."""
    info = O.default_info(1, 0)
    r = O.render(stock_scenes[name], 640, 480, info, threads=THREADS, want_counters=True, want_evals=True)
    assert hashlib.sha256(O.ppm_bytes(r["rgba"])).hexdigest() == REFERENCE_PPM_SHA256[name]
    assert int(r["hit"].sum()) == REFERENCE_HITS[(name, 640, 480)]
    assert round(float(r["evals"].mean()), 1) == REFERENCE_MEAN_ROOT_EVALS[(name, 640, 480)]
    calls = sum(r["counters"]["dist_calls"].values()) / (640 * 480)
    assert round(calls, 1) == REFERENCE_NODE_CALLS_PER_PIXEL[(name, 640, 480)]
    assert r["counters"]["lum_out_of_range"] == 0 and r["counters"]["hit_colour_disagree"] == 0


@pytest.mark.parametrize("name", ["chair", "infinity"])
def test_pinned_by_reference_counts_1080p(stock_scenes, name):
    info = O.default_info(1, 0)
    r = O.render(stock_scenes[name], 1920, 1080, info, threads=THREADS, want_counters=True, want_evals=True)
    assert int(r["hit"].sum()) == REFERENCE_HITS[(name, 1920, 1080)]
    assert round(float(r["evals"].mean()), 1) == REFERENCE_MEAN_ROOT_EVALS[(name, 1920, 1080)]
    calls = sum(r["counters"]["dist_calls"].values()) / (1920 * 1080)
    assert round(calls, 1) == REFERENCE_NODE_CALLS_PER_PIXEL[(name, 1920, 1080)]
    # and the committed fixture of the same frame
    with open(os.path.join(os.path.dirname(__file__), "golden", "frames.json")) as f:
        want = json.load(f)["%s_1920x1080" % name]
    assert hashlib.sha256(r["rgba"].tobytes()).hexdigest() == want["rgba_sha256"]
    assert r["counters"]["flops"] == want["counters"]["flops"]
    packed = np.packbits(r["hit"].astype(bool), axis=None).tobytes()
    with open(os.path.join(os.path.dirname(__file__), "golden", "hit_%s_1920x1080.z" % name), "rb") as f:
        assert zlib.decompress(f.read()) == packed


def test_hand_derived_known_answers():
    """DIST and NORM at points where the reference's formulas (Source/cuda_kernels.cu:51-592, cited per case in
    kat_scenes.ANALYTIC) have exact closed forms worked out by hand: every node type, both gates of the smooth
    combinators, every tie rule of the normals.  The one check of the restatement against something that is neither
    itself nor the survey's hashes - and the only one for the node types the stock scenes do not use."""
    cases = kat_scenes.analytic_cases()
    assert {kat_scenes.ANALYTIC[i][0][0] for i in range(len(cases))} == set(O.TYPE_NAMES)
    for i, (table, p, want_d, want_n) in enumerate(cases):
        got_d = np.float32(O.dist(table, 0, p))
        got_n = np.array(O.norm(table, 0, p), dtype=np.float32)
        assert got_d == want_d, "case %d (%s at %s): DIST %r, by hand %r" % (i, kat_scenes.ANALYTIC[i][0][0], p, got_d, want_d)
        assert np.array_equal(got_n, want_n), "case %d (%s at %s): NORM %s, by hand %s" % (i, kat_scenes.ANALYTIC[i][0][0], p, got_n, want_n)


def test_node_kat_fixture(golden_dir):
    """Node-level known answers frozen in node_kat.npz still hold (compiler / edit drift)."""
    kat = np.load(os.path.join(golden_dir, "node_kat.npz"))
    pts = kat["points"]
    assert np.array_equal(pts, kat_scenes.points())
    for name, table in kat_scenes.tables().items():
        assert O.validate(table) == 0
        assert np.array_equal(O.dist_many(table, 0, pts).view(np.uint32), kat[name + ".dist"]), name
        assert np.array_equal(O.norm_many(table, 0, pts).view(np.uint32), kat[name + ".norm"]), name


def test_set_rays_fixture(golden_dir):
    kat = np.load(os.path.join(golden_dir, "set_rays_kat.npz"))
    keys = sorted(k[:-3] for k in kat.files if k.endswith(".xy"))
    assert keys
    for key in keys:
        w, h = (int(v) for v in key.split("_")[0].split("x"))
        info = kat[key + ".info"].view(O.INFO_DTYPE)
        got = np.stack([O.set_ray(w, h, info, int(x), int(y)) for x, y in kat[key + ".xy"]])
        assert np.array_equal(got.view(np.uint32), kat[key + ".rays"])


def test_sweep_fixture(golden_dir):
    """The 120 camera blocks of the 8K sweep (config 5) follow main.cpp's own float steps."""
    sweep = np.fromfile(os.path.join(golden_dir, "sweep_rays_info.bin"), dtype=O.INFO_DTYPE)
    assert len(sweep) == 120
    for k in (0, 1, 59, 119):
        assert sweep[k].tobytes() == O.default_info(k + 1, k)[0].tobytes()
    # frame 0 is the default camera with the light one degree on
    assert sweep[0]["StartDir"]["x"] == np.float32(2.0) ** np.float32(-6.1)
    assert sweep[0]["StartDir"]["y"] == 0 and sweep[0]["StartHVec"]["z"] == -sweep[0]["StartDir"]["x"]


def test_camera_script(golden_dir):
    """rmo_camera_script (the reference's event loop, main.cpp:144-270, as a scripted path): a right-drag of
    one pixel per frame is the committed sweep; every other event against the closed forms of its lines."""
    sweep = np.fromfile(os.path.join(golden_dir, "sweep_rays_info.bin"), dtype=O.INFO_DTYPE)
    lines = ["frame"] + ["drag 1 0", "frame"] * 119
    assert O.camera_script(lines).tobytes() == sweep.tobytes()
    f32 = np.float32
    base = O.default_info(0, 0)[0]
    # pause holds the light, left and right undo each other only up to rounding, resume lets it turn again
    got = O.camera_script(["pause", "frame", "left", "frame", "resume", "frame"])
    assert got[0]["LightSource"] == base["LightSource"]
    assert got[1]["LightSource"] == O.default_info(1, 0)[0]["LightSource"]
    assert got[2]["LightSource"] == O.default_info(2, 0)[0]["LightSource"]
    # wheel: scale *= 2^(0.1 y) (:220); space: scale = 2^-2.1 and everything else back (:173-184)
    got = O.camera_script(["wheel 3", "pan 5 -2", "frame", "space", "frame"])
    def pow2(e):                                                  # powf(2.f, e): the double result, rounded once
        return f32(2.0 ** float(f32(e)))
    scale = f32(pow2(-6.1) * pow2(f32(3) * f32(0.1)))
    assert got[0]["StartDir"]["x"] == scale and got[0]["StartWVec"]["y"] == -scale
    S = f32(f32(2.0) / scale)                                     # :232-238, WVec = (0,-scale,0), HVec = (0,0,-scale)
    assert got[0]["StartPos"]["y"] == f32(f32(0) - f32(f32(f32(5) * -scale) * S))
    assert got[0]["StartPos"]["z"] == f32(f32(0) - f32(f32(f32(-2) * -scale) * S))
    assert got[1]["StartDir"]["x"] == pow2(-2.1) and got[1]["StartPos"]["y"] == 0 and got[1]["StartPos"]["z"] == 0
    # theta is clamped to +-pi/2 (:241-244)
    got = O.camera_script(["drag 0 400", "frame", "drag 0 -900", "frame"])
    assert got[0]["StartDir"]["z"] == pow2(-6.1)                  # scale * sinf(pi/2)
    assert got[1]["StartDir"]["z"] == -got[0]["StartDir"]["z"]


def test_sce_codec_round_trip(stock_scenes, golden_dir):
    """Appendix B: decode the stock files, re-encode, decode again."""
    for name, nodes in stock_scenes.items():
        again = O.decode_bazo(O.encode_sce(nodes), O.BAZO_SCE)
        assert again.tobytes() == nodes.tobytes()
    chair = stock_scenes["chair"]
    assert [int(t) for t in chair["type"]] == [0x205, 0x200, 0x204, 0x103, 0x3, 0x106, 0x100, 0x200, 0x2, 0x200, 0x2]
    assert list(chair[3]["o"][:2]) == [1, 2] and list(chair[6]["o"][:2]) == [1, 3]
    assert chair[2]["f"][0].view(np.uint32) == 0x3F1F21D5          # q_w, appendix B
    inf = stock_scenes["infinity"]
    assert [int(t) for t in inf["type"]] == [0x106, 0x205, 0x205, 0x205, 0x2] and inf[4]["f"][0] == 65.0


def test_validate_rejects_bad_tables(stock_scenes):
    bad = stock_scenes["chair"].copy()
    bad[10]["type"] = 0                               # nenio leaf: a null call in the reference
    assert O.validate(bad) != 0
    bad = stock_scenes["chair"].copy()
    bad[9]["o"][0] = 5                                # link past the end of the table
    assert O.validate(bad) != 0
    bad = stock_scenes["chair"].copy()
    bad[1]["o"][0] = 0                                # self reference: infinite recursion in the reference
    assert O.validate(bad) != 0


def test_hypot_convention():
    """length_2 == glibc hypotf bit for bit (rm_oracle.h arithmetic conventions)."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.hypotf.restype = ctypes.c_float
    libm.hypotf.argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(3)
    xy = (rng.standard_normal((4000, 2)) * rng.choice([1e-3, 1.0, 50.0, 1e4], size=(4000, 1))).astype(np.float32)
    table = kat_scenes.Tree(("cilindro", (0.0, 1e9))).table()     # dist = hypot(x, y) - 0 inside the slab
    for x, y in xy[:1500]:
        want = np.float32(libm.hypotf(float(x), float(y)))
        got = np.float32(O.dist(table, 0, (x, y, 0.0)))
        # exterior branch re-applies hypot to (max(q.x,0), max(q.y,0)) = (want, 0): still `want`
        assert got == want


def test_hand_derived_whole_pixels():
    """The march and shading loop (Source/cuda_kernels.cu:619-717) against answers worked out on paper
    (kat_scenes.PIXELS): the pixel, the hit flag, and how many root evaluations each loop made - the only pin of
    that loop beyond the survey's two frame hashes.  A 9x9 frame whose 81 rays are the same ray."""
    import kat_scenes
    for name, table, info, rgba, hit, n_primary, n_shadow in kat_scenes.pixel_cases(O.INFO_DTYPE):
        r = O.render(table, 9, 9, info, want_evals=True, want_counters=True)
        assert (r["rgba"] == rgba).all(), "%s: pixel %s, by hand %s" % (name, r["rgba"][4, 4], rgba)
        assert (r["hit"] == hit).all(), name
        assert r["counters"]["root_dist_primary"] == 81 * n_primary, (name, r["counters"]["root_dist_primary"] / 81)
        assert r["counters"]["root_dist_shadow"] == 81 * n_shadow, (name, r["counters"]["root_dist_shadow"] / 81)
        assert (r["evals"] == n_primary + n_shadow).all(), name
