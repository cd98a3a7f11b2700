"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
restatement (oracle/) on the same inputs.  Bar: bit-exact.

The kernels use IEEE fp32 add/mul/div/sqrt without contraction in the
reference's operation order, so no tolerance is needed or allowed here:
frames must be byte-identical (north_star allows <=1 LSB per channel and
counted grazing-pixel mask flips; we measure 0 of either).
"""
import hashlib
import os

import numpy as np
import pytest

import kat_scenes
from oracle import oracle as O
from raytracing_amd import api, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# sha256 of the P6 PPM the reference's march produced host-side
GOLDEN_PPM = {
    "chair": "f5c226d07e78e48b447e0456fc75922d8fe8f0794e9569cfc7b5efabfea489ec",
    "infinity": "369094d61c29b3a505f77cde384c40af3c8b427acc609ecc5ca01086ac6a59e8",
}
GOLDEN_HITS = {("chair", 640, 480): 259549, ("infinity", 640, 480): 307136,
               ("chair", 1920, 1080): 296418, ("infinity", 1920, 1080): 2073420}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def differences(got, want):
    """Where two frames differ, for an assertion message."""
    bad = np.argwhere(got != want)
    return "%d of %d values differ, first at %s: got %s, want %s" % (
        len(bad), want.size, bad[:8].tolist(), [int(got[tuple(i)]) for i in bad[:8]], [int(want[tuple(i)]) for i in bad[:8]])


def assert_same_frame(rgba, hit, ref, label):
    """RGBA8 frame AND hit mask against the oracle's, both compared before anything is asserted, so that a
    failure says whether the pixels, the mask or both were wrong; got and want go to a file under
    gpurun_out/ (which travels back from the GPU box) for the post-mortem."""
    ok_rgba = np.array_equal(rgba, ref["rgba"])
    ok_hit = hit is None or np.array_equal(hit, ref["hit"])
    if ok_rgba and ok_hit:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "mismatch_%s.npz" % "".join(ch if ch.isalnum() else "_" for ch in str(label)))
    np.savez_compressed(path, got_rgba=rgba, want_rgba=ref["rgba"], got_hit=hit if hit is not None else np.zeros(0, np.uint8),
                        want_hit=ref["hit"])
    msg = ["%s: frame differs from the oracle (dump: %s)" % (label, path)]
    msg.append("rgba " + ("identical" if ok_rgba else differences(rgba, ref["rgba"])))
    if hit is not None:
        msg.append("hit mask " + ("identical" if ok_hit else differences(hit, ref["hit"])))
        if not ok_hit:               # a tile that nobody wrote shows as an 8x8-aligned block: say which tiles
            ys, xs = np.nonzero(hit != ref["hit"])
            msg.append("8x8 tiles holding mask differences: %s" % sorted(set(zip((ys // 8).tolist(), (xs // 8).tolist())))[:16])
    raise AssertionError("; ".join(msg))


def same_floats(a, b):
    """bit-identical, except that +0 and -0 compare equal (the program
    re-associates exact min/max chains, which can only change the sign of a zero)
    and that any NaN equals any NaN (an invalid operation yields 0xFFC00000 on
    x86 and 0x7FC00000 on gfx950; e.g. the normalised normal on a cylinder's axis)"""
    fa, fb = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    a, b = bits(a), bits(b)
    return np.all((a == b) | (((a | b) & 0x7FFFFFFF) == 0) | (np.isnan(fa) & np.isnan(fb)))


# the three device evaluators of DIST(root, p): per-scene generated code (hiprtc),
# the wave-uniform distance program, the per-lane tree walker -> (jit mode, evaluator)
PATHS = {"jit": (2, 0), "jitprog": (3, 0), "program": (0, 0), "walker": (0, 1)}


def gpu_render(nodes, w, h, info, path="jit"):
    jit, evaluator = PATHS[path]
    with api.Raymarching(w, h) as rm:
        rm.set_jit(jit)
        rm.set_nodes(nodes)
        assert rm.scene_stats()["jit"] == (1 if jit else 0)
        rm.set_info(info)
        rm.enable_hit_mask(True)
        rm.set_evaluator(evaluator)
        ms = rm.process()
        return rm.read_pixels(), rm.read_hit_mask(), ms


@pytest.mark.parametrize("name", sorted(kat_scenes.SCENES))
def test_node_kat(name):
    """DIST and NORM of every node type / flattener path at seeded points."""
    table = kat_scenes.tables()[name]
    pts = kat_scenes.points()
    want_d = O.dist_many(table, 0, pts)
    want_n = O.norm_many(table, 0, pts)
    with api.Raymarching(64, 64) as rm:
        rm.set_jit(2)
        rm.set_nodes(table)
        d_prog, _ = rm.eval_points(pts, 0, which=0, want_norm=False)
        d_walk, n_walk = rm.eval_points(pts, 0, which=1, want_norm=True)
        d_jit, n_jit = rm.eval_points(pts, 0, which=2, want_norm=True)
        rm.set_jit(3)                              # the distance program compiled, normals by the walker
        rm.set_nodes(table)
        d_jp, n_jp = rm.eval_points(pts, 0, which=2, want_norm=True)
    assert same_floats(d_jp, want_d), "compiled distance program DIST differs from the oracle"
    assert same_floats(n_jp, want_n), "compiled distance program NORM differs from the oracle"
    assert same_floats(d_walk, want_d), "tree walker DIST differs from the oracle"
    assert same_floats(d_prog, want_d), "distance program DIST differs from the oracle"
    assert same_floats(n_walk, want_n), "tree walker NORM differs from the oracle"
    assert same_floats(d_jit, want_d), "generated code DIST differs from the oracle"
    assert same_floats(n_jit, want_n), "generated code NORM differs from the oracle"


def test_hand_derived_known_answers():
    """The hand-derived closed forms of kat_scenes.ANALYTIC (worked out from the reference's source text, exact
    in fp32) on every device evaluator: the device against the paper, not against the oracle."""
    for i, (table, p, want_d, want_n) in enumerate(kat_scenes.analytic_cases()):
        with api.Raymarching(64, 64) as rm:
            rm.set_jit(2)
            rm.set_nodes(table)
            pts = p.reshape(1, 3)
            got = {"program": (rm.eval_points(pts, 0, which=0, want_norm=False)[0], None),
                   "walker": rm.eval_points(pts, 0, which=1, want_norm=True),
                   "generated": rm.eval_points(pts, 0, which=2, want_norm=True)}
            rm.set_jit(3)
            rm.set_nodes(table)
            got["compiled program"] = rm.eval_points(pts, 0, which=2, want_norm=True)
        for name, (d, n) in got.items():
            label = "case %d (%s at %s) on the %s" % (i, kat_scenes.ANALYTIC[i][0][0], p, name)
            assert same_floats(d, [want_d]), "%s: DIST %r, by hand %r" % (label, d, want_d)
            if n is not None:
                assert same_floats(n[0], want_n), "%s: NORM %s, by hand %s" % (label, n[0], want_n)


def test_hand_derived_whole_pixels():
    """kat_scenes.PIXELS on the device: the march and shading loop (Source/cuda_kernels.cu:619-717) of every
    kernel - generated code, compiled distance program, interpreter, tree walker - against pixels worked out on
    paper, not against the oracle: a miss by distance, a lit floor, an occluded hit, a start inside a solid,
    near-surface samples that march on, and both step caps.  With ray windows off the tiles' march-loop trips
    must also be the root evaluations counted by hand (every lane of a tile is the same ray, so a tile makes
    primary + shadow trips; none of the latter where the surface faces away from the light)."""
    for name, table, info, rgba, hit, n_primary, n_shadow in kat_scenes.pixel_cases(api.INFO_DTYPE):
        for path in sorted(PATHS):
            got, mask, _ = gpu_render(table, 9, 9, info, path)
            assert (got == rgba).all(), "%s on %s: pixel %s, by hand %s" % (name, path, got[4, 4], rgba)
            assert (mask == hit).all(), "%s on %s: hit flag" % (name, path)
        for jit in (2, 0):
            with api.Raymarching(9, 9) as rm:
                rm.set_jit(jit)
                rm.set_ray_windows(False)
                rm.set_nodes(table)
                rm.set_info(info)
                stats = rm.wave_stats()
                assert int(stats[:, 6].sum()) == 4, "four tiles"
                device_shadow = 0 if name in kat_scenes.PIXELS_UNLIT else n_shadow     # (an unlit hit is shaded without its shadow march)
                assert int(stats[:, 0].sum()) == 4 * (n_primary + device_shadow), (name, jit, int(stats[:, 0].sum()) / 4)
                assert (rm.read_pixels() == rgba).all(), name


def test_node_kat_inner_nodes():
    """The walker can start at any reachable node (sub-tree evaluation)."""
    table = kat_scenes.tables()["nested_frames"]
    pts = kat_scenes.points(64)
    with api.Raymarching(64, 64) as rm:
        rm.set_nodes(table)
        for idx in range(len(table)):
            d, n = rm.eval_points(pts, idx, which=1)
            assert same_floats(d, O.dist_many(table, idx, pts)), idx
            assert same_floats(n, O.norm_many(table, idx, pts)), idx


@pytest.mark.parametrize("seed", range(int(os.environ.get("RM_FUZZ_SEEDS", "24"))))
def test_random_trees(seed):
    """Fuzz: a seeded random tree over all 20 node types (kat_scenes.random_tree), DIST and NORM at seeded
    points on every device evaluator and a small frame on the three kernels, against the oracle."""
    table = scenes.Tree(kat_scenes.random_tree(1000 + seed, depth=3 + seed % 4)).table()
    pts = kat_scenes.points(96, seed=seed)
    want_d, want_n = O.dist_many(table, 0, pts), O.norm_many(table, 0, pts)
    with api.Raymarching(64, 64) as rm:
        rm.set_jit(2)
        rm.set_nodes(table)
        d_prog, _ = rm.eval_points(pts, 0, which=0, want_norm=False)
        d_walk, n_walk = rm.eval_points(pts, 0, which=1, want_norm=True)
        d_jit, n_jit = rm.eval_points(pts, 0, which=2, want_norm=True)
        rm.set_jit(3)
        rm.set_nodes(table)
        d_jp, n_jp = rm.eval_points(pts, 0, which=2, want_norm=True)
    for name, got, want in (("distance program DIST", d_prog, want_d), ("walker DIST", d_walk, want_d),
                            ("walker NORM", n_walk, want_n), ("generated DIST", d_jit, want_d),
                            ("generated NORM", n_jit, want_n), ("compiled program DIST", d_jp, want_d),
                            ("compiled program NORM", n_jp, want_n)):
        assert same_floats(got, want), "%s differs from the oracle (%d nodes)" % (name, len(table))
    info = O.default_info(2, seed % 8)
    info["StartPos"] = (-45.0, 2.0, 1.0)
    for v in ("StartWVec", "StartHVec"):           # four times the default field of view: the whole scene in a small frame
        for a in "xyz":
            info[v][a] *= np.float32(4)
    ref = O.render(table, 160, 96, info)
    for path in ("jit", "jitprog", "program"):
        rgba, hit, _ = gpu_render(table, 160, 96, info, path)
        assert_same_frame(rgba, hit, ref, "fuzz_%d_%s" % (seed, path))


@pytest.mark.parametrize("seed", range(int(os.environ.get("RM_CARVE_SEEDS", "16"))))
def test_random_carved_solids(seed):
    """Fuzz of the generated code's sphere_skip (jit.cpp): solids with spheres cut out of them (kat_scenes.random_carved),
    a frame from a seeded camera and light on the three kernels, DIST and NORM at seeded points, against the oracle."""
    table = scenes.Tree(kat_scenes.random_carved(500 + seed)).table()
    pts = kat_scenes.points(96, seed=100 + seed)
    want_d, want_n = O.dist_many(table, 0, pts), O.norm_many(table, 0, pts)
    with api.Raymarching(64, 64) as rm:
        rm.set_jit(2)
        rm.set_nodes(table)
        d_jit, n_jit = rm.eval_points(pts, 0, which=2, want_norm=True)
    assert same_floats(d_jit, want_d) and same_floats(n_jit, want_n), "generated DIST / NORM differ from the oracle"
    info = O.default_info(1 + seed % 7, (seed * 5) % 23)
    info["StartPos"] = (-42.0 + seed % 5, 2.0 - seed % 3, 1.0 + seed % 4)
    for v in ("StartWVec", "StartHVec"):
        for a in "xyz":
            info[v][a] *= np.float32(3)
    ref = O.render(table, 192, 112, info)
    for path in ("jit", "jitprog", "program"):
        rgba, hit, _ = gpu_render(table, 192, 112, info, path)
        assert_same_frame(rgba, hit, ref, "carved_%d_%s" % (seed, path))


@pytest.mark.parametrize("seed", range(int(os.environ.get("RM_LATTICE_SEEDS", "16"))))
def test_random_lattices(seed):
    """Fuzz of the generated code's cell_cache (jit.cpp): scenes that are mostly senfina_ripeto (kat_scenes.random_lattice),
    frames from a near and a far camera (cells beyond the 127 the cache keeps) on the three kernels, DIST and NORM at
    seeded points near and far, against the oracle."""
    table = scenes.Tree(kat_scenes.random_lattice(900 + seed)).table()
    pts = np.concatenate([kat_scenes.points(64, seed=200 + seed, span=150.0),
                          kat_scenes.points(32, seed=300 + seed, span=30000.0)])
    want_d, want_n = O.dist_many(table, 0, pts), O.norm_many(table, 0, pts)
    with api.Raymarching(64, 64) as rm:
        rm.set_jit(2)
        rm.set_nodes(table)
        d_jit, n_jit = rm.eval_points(pts, 0, which=2, want_norm=True)
    assert same_floats(d_jit, want_d) and same_floats(n_jit, want_n), "generated DIST / NORM differ from the oracle"
    for far in (False, True):
        info = O.default_info(1 + seed % 5, (seed * 7) % 19)
        info["StartPos"] = (-9000.0 - 31 * seed, 40.0, 7000.0 + seed) if far else (-42.0 + seed % 5, 2.0 - seed % 3, 1.0 + seed % 4)
        ref = O.render(table, 160, 96, info, threads=min(16, os.cpu_count() or 4))
        for path in ("jit", "program"):
            rgba, hit, _ = gpu_render(table, 160, 96, info, path)
            assert_same_frame(rgba, hit, ref, "lattice_%d_%s_%s" % (seed, "far" if far else "near", path))


def test_set_rays_kat():
    """kernel_SetRays (Source/cuda_kernels.cu:730-762) fused into the kernel."""
    for (w, h, phi) in [(640, 480, 0), (1920, 1080, 37), (7680, 4320, 119), (33, 17, 5)]:
        info = O.default_info(1, phi)
        rng = np.random.default_rng(w)
        xy = np.stack([rng.integers(0, w, 256), rng.integers(0, h, 256)], axis=1).astype(np.int32)
        xy[:4] = [[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1]]
        with api.Raymarching(w, h) as rm:
            rm.set_info(info)
            got = rm.eval_rays(xy)
        want = np.stack([O.set_ray(w, h, info, int(x), int(y)) for x, y in xy])
        assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("name", ["chair", "infinity"])
@pytest.mark.parametrize("path", sorted(PATHS))
def test_stock_frame_640x480(stock_scenes, name, path):
    """SURVEY.md 8d config 1 on the GPU: byte-identical frame, golden sha256, hit count."""
    info = O.default_info(1, 0)
    rgba, hit, _ = gpu_render(stock_scenes[name], 640, 480, info, path)
    ref = O.render(stock_scenes[name], 640, 480, info, threads=min(16, os.cpu_count() or 4))
    assert_same_frame(rgba, hit, ref, "stock_%s_%s" % (name, path))
    assert int(hit.sum()) == GOLDEN_HITS[(name, 640, 480)]
    assert hashlib.sha256(O.ppm_bytes(rgba)).hexdigest() == GOLDEN_PPM[name]
    # q6: the explicit hit bit and the colour-derived mask agree on the stock scenes
    assert np.array_equal(hit.astype(bool), rgba[:, :, :3].any(-1))


def _golden_frames():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "frames.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("key", sorted(_golden_frames()))
def test_committed_golden_frames(key, stock_scenes, golden_dir):
    """Every frame tests/golden/make_goldens.py recorded from the oracle (configs 1-3 at full size, the
    1024-node scene at 960x540): RGBA8 sha256, hit count, and the bit-packed hit mask, bit for bit."""
    import zlib
    rec = _golden_frames()[key]
    name = key.rsplit("_", 1)[0]
    w, h = rec["width"], rec["height"]
    table = scenes.csg1024() if name == "csg1024" else stock_scenes[name]
    rgba, hit, _ = gpu_render(table, w, h, O.default_info(1, 0))
    assert int(hit.sum()) == rec["hits"]
    assert hashlib.sha256(rgba.tobytes()).hexdigest() == rec["rgba_sha256"]
    with open(os.path.join(golden_dir, "hit_%s.z" % key), "rb") as f:
        want = np.frombuffer(zlib.decompress(f.read()), dtype=np.uint8)
    assert np.array_equal(np.packbits(hit.astype(bool), axis=None), want)


@pytest.mark.parametrize("name", ["chair", "infinity"])
@pytest.mark.parametrize("path", ["jit", "program"])
def test_stock_hits_1080p(stock_scenes, name, path):
    """config 2: hit count recorded from the reference's host-side run at 1920x1080."""
    info = O.default_info(1, 0)
    rgba, hit, _ = gpu_render(stock_scenes[name], 1920, 1080, info, path)
    assert int(hit.sum()) == GOLDEN_HITS[(name, 1920, 1080)]
    # rows sampled across the frame, byte-for-byte against the oracle
    for y0 in (0, 270, 536, 540, 812, 1079):
        ref = O.render(stock_scenes[name], 1920, 1080, info, rows=(y0, y0 + 1))
        assert np.array_equal(rgba[y0], ref["rgba"][y0]), y0
        assert np.array_equal(hit[y0], ref["hit"][y0]), y0


def test_csg1024_frame():
    """config 4's 1024-record table (beyond the reference's 256-slot limit) at a
    size the oracle finishes in seconds."""
    table = scenes.csg1024()
    info = O.default_info(1, 0)
    w, h = 256, 144
    rgba, hit, _ = gpu_render(table, w, h, info, "program")
    ref = O.render(table, w, h, info, threads=min(16, os.cpu_count() or 4))
    assert ref["hit"].sum() > 0
    assert_same_frame(rgba, hit, ref, "csg1024_program")
    for path in ("walker", "jit"):              # "jit": 1023 nodes, so the compiled distance program
        rgba_w, hit_w, _ = gpu_render(table, w, h, info, path)
        assert np.array_equal(rgba_w, rgba) and np.array_equal(hit_w, hit), path


@pytest.mark.parametrize("seed,leaves", [(11, 21), (12, 22), (13, 70), (14, 200), (15, 450), (17, 682)])
def test_union_scenes_of_other_sizes(seed, leaves):
    """config 4's generator at other seeds and sizes: 63 nodes (the last size that becomes one fused
    function), 66 (the first compiled distance program), unions cut into fewer or more groups than
    csg1024's, 2046 nodes (RM_MAX_NODES is 2048); generated code and the interpreter against the oracle."""
    table = scenes.csg_union(seed, leaves)
    assert len(table) == 3 * leaves
    info = O.default_info(2, seed)
    w, h = 192, 108
    ref = O.render(table, w, h, info, threads=min(16, os.cpu_count() or 4))
    assert ref["hit"].sum() > 0
    for path in ("jit", "program"):
        rgba, hit, _ = gpu_render(table, w, h, info, path)
        assert_same_frame(rgba, hit, ref, "union_%d_%s" % (seed, path))


@pytest.mark.parametrize("name", ["mixed_kinds", "carved", "carved_spheres", "nested_frames", "glata_nested", "portanta_nested", "ripeto_frames", "deep_values"])
@pytest.mark.parametrize("path", ["jit", "jitprog", "program"])
def test_kat_scene_frames(name, path):
    """Whole frames of the nested test scenes (LDS value slots, saved frames, bounding spheres)."""
    table = kat_scenes.tables()[name]
    info = O.default_info(3, 11)
    info["StartPos"] = (-40.0, 3.0, 2.0)
    w, h = 200, 120
    rgba, hit, _ = gpu_render(table, w, h, info, path)
    ref = O.render(table, w, h, info, threads=min(16, os.cpu_count() or 4))
    assert_same_frame(rgba, hit, ref, "kat_%s_%s" % (name, path))


@pytest.mark.parametrize("path", ["jit", "program"])
def test_ray_windows_change_no_pixel(stock_scenes, path):
    """Ray windows (include/raymarching_amd.h): with them and without, frames are the oracle's (which marches
    every ray to RM_MAX_DIST as the reference does), from cameras outside, inside and far from the scene's
    volume; with them the march loop makes fewer trips."""
    nodes = stock_scenes["chair"]
    jit, evaluator = PATHS[path]

    def cam(light, drag, pos=None, wide=None):
        info = O.default_info(light, drag)
        if pos is not None:
            info["StartPos"] = pos
        if wide is not None:
            for v in ("StartWVec", "StartHVec"):
                for a in "xyz":
                    info[v][a] *= np.float32(wide)
        return info

    cams = {"default": cam(1, 0), "wide": cam(1, 0, wide=5.0), "sideways": cam(3, 120), "inside": cam(2, 30, pos=(200.0, 10.0, 5.0)),
            "along_the_axis": cam(2, 180, pos=(200.0, -300.0, -60.0)), "far_below": cam(1, 0, pos=(0.0, 0.0, -2500.0)),
            "beyond_the_limit": cam(1, 0, pos=(-20000.0, 0.0, 0.0), wide=0.05)}
    w, h = 320, 240
    trips = {}
    with api.Raymarching(w, h) as rm:
        rm.set_jit(jit)
        rm.set_nodes(nodes)
        rm.set_evaluator(evaluator)
        rm.enable_hit_mask(True)
        for label, info in cams.items():
            ref = O.render(nodes, w, h, info, threads=8)
            for windows in (True, False):
                rm.set_ray_windows(windows)
                rm.set_info(info)
                rm.process()
                assert_same_frame(rm.read_pixels(), rm.read_hit_mask(), ref, "windows_%s_%s_%d" % (path, label, windows))
                trips[(label, windows)] = int(rm.wave_stats()[:, 0].sum())
    assert trips[("wide", True)] < 0.8 * trips[("wide", False)], trips
    assert trips[("far_below", True)] == 0 < trips[("far_below", False)], trips
    assert trips[("beyond_the_limit", True)] == trips[("beyond_the_limit", False)], trips


def test_ragged_sizes(stock_scenes):
    """Image sizes that are not multiples of the 16x16 workgroup block."""
    info = O.default_info(1, 0)
    for (w, h), path in [((1, 1), "jit"), ((17, 9), "program"), ((250, 131), "jit"), ((250, 131), "program")]:
        rgba, hit, _ = gpu_render(stock_scenes["chair"], w, h, info, path)
        ref = O.render(stock_scenes["chair"], w, h, info)
        assert_same_frame(rgba, hit, ref, "ragged_%dx%d_%s" % (w, h, path))


def test_row_bands_reassemble(stock_scenes):
    """Multi-GPU partition on one GPU: every rank's bands, interleaved back, equal the whole frame."""
    from raytracing_amd import tiling
    info = O.default_info(1, 0)
    w, h = 320, 250
    whole, _, _ = gpu_render(stock_scenes["chair"], w, h, info)
    for nranks, band in [(2, 16), (3, 32), (8, 16), (5, 8), (3, 24)]:
        parts = []
        for rank in range(nranks):
            with api.Raymarching(w, h) as rm:
                rm.set_nodes(stock_scenes["chair"])
                rm.set_info(info)
                rm.set_bands(band, rank, nranks)
                assert rm.local_rows() == tiling.local_rows(h, band, rank, nranks)
                rm.process()
                parts.append(rm.read_pixels())
        frame = tiling.assemble(parts, h, band)
        assert np.array_equal(frame, whole)


@pytest.mark.parametrize("name", ["chair", "infinity"])
def test_luminance_planes_rebuild_the_frame(stock_scenes, name):
    """rm_set_luminance_output + rm_assemble_luminance: one byte per pixel from every rank, put back
    together on one device, is the RGBA8 frame byte for byte (whole frame and 3-rank bands, odd width)."""
    torch = pytest.importorskip("torch")
    info = O.default_info(1, 0)
    w, h = 322, 250
    whole, _, _ = gpu_render(stock_scenes[name], w, h, info)
    for nranks, band in [(1, 0), (3, 32), (8, 16)]:
        rows_max = -(-h // (band * nranks)) * band if band else h
        pitch = w + 5
        parts = torch.full((nranks, rows_max, pitch), 0xAB, dtype=torch.uint8, device="cuda")
        keep = []
        for rank in range(nranks):
            rm = api.Raymarching(w, h)
            rm.set_nodes(stock_scenes[name])
            rm.set_info(info)
            if band:
                rm.set_bands(band, rank, nranks)
            rm.set_luminance_output(parts[rank].data_ptr(), pitch)
            rm.process()
            # the RGBA8 output is still written, and is this rank's share of the frame
            assert np.array_equal(rm.read_pixels()[:, :, 2], parts[rank, :rm.local_rows(), :w].cpu().numpy())
            keep.append(rm)
        out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        keep[0].assemble_luminance(parts.data_ptr(), rows_max * pitch, pitch, out.data_ptr(), w * 4)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), whole), (nranks, band)
        with pytest.raises(api.RaymarchError):
            keep[0].assemble_luminance(parts.data_ptr(), rows_max * pitch, w - 1, out.data_ptr(), w * 4)
        for rm in keep:
            rm.close()


def test_uneven_band_deals_and_the_batched_rebuild(stock_scenes):
    """rm_set_band_pattern + rm_assemble_luminance_batch: bands dealt by weight (rank 0 fewer, one rank none at
    all, a rank with several bands of a run), three frames per launch from different cameras, every rank's
    luminance planes put back together in one kernel, are the frames rendered whole - RGBA8 rows read back
    per rank too.  Once with planes laid out for the vector path (4-byte loads, 16-byte stores) and once with a
    pitch that forbids it; the single-frame call must agree with the batch."""
    torch = pytest.importorskip("torch")
    from raytracing_amd import tiling
    nodes = stock_scenes["chair"]
    w, h, n = 328, 250, 3
    infos = np.concatenate([O.default_info(k + 1, 9 * k) for k in range(n)])
    whole = []
    with api.Raymarching(w, h) as rm:
        rm.set_nodes(nodes)
        for k in range(n):
            rm.set_info(infos[k:k + 1])
            rm.process()
            whole.append(rm.read_pixels())
    assert sum(int(f[:, :, :3].any(-1).sum()) for f in whole) > 3000
    for weights, band, pitch in [([1, 2, 2], 16, w), ([3, 5, 5, 5, 5, 5, 5, 5], 8, w + 8), ([0, 1, 1, 1], 32, w), ([2, 16], 8, w + 3)]:
        nranks, owner = len(weights), tiling.deal(weights)
        rows_max = -(-tiling.max_local_rows(h, band, nranks, owner) // 16) * 16
        parts = torch.full((nranks, n, rows_max, pitch), 0xAB, dtype=torch.uint8, device="cuda")
        keep = []
        for rank in range(nranks):
            rm = api.Raymarching(w, h)
            rm.set_nodes(nodes)
            rm.set_bands(band, rank, nranks, owner)
            assert rm.local_rows() == tiling.local_rows(h, band, rank, nranks, owner)
            rm.set_batch(n, 0, rows_max * pitch)
            rm.set_luminance_output(parts[rank].data_ptr(), pitch)
            rm.set_info_batch(infos)
            rm.process()
            keep.append(rm)
        rgba_parts = [rm.read_pixels() for rm in keep]                     # frame 0 of the launch, this rank's rows
        assert np.array_equal(tiling.assemble(rgba_parts, h, band, owner), whole[0]), weights
        out = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
        keep[-1].assemble_luminance_batch(parts.data_ptr(), n * rows_max * pitch, rows_max * pitch, pitch, n,
                                          out.data_ptr(), w * 4, h * w * 4)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for k in range(n):
            assert np.array_equal(got[k], whole[k]), (weights, k)
        one = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        keep[0].assemble_luminance(parts[:, 1].data_ptr(), n * rows_max * pitch, pitch, one.data_ptr(), w * 4)
        torch.cuda.synchronize()
        assert np.array_equal(one.cpu().numpy(), whole[1]), weights
        with pytest.raises(api.RaymarchError):                             # frames that would overlap
            keep[0].assemble_luminance_batch(parts.data_ptr(), n * rows_max * pitch, rows_max * pitch, pitch, n,
                                             out.data_ptr(), w * 4, h * w * 4 - 4)
        for rm in keep:
            rm.close()
    with api.Raymarching(w, h) as rm:
        for owner, rank, nranks in [([0, 1, 2], 0, 2), ([0] * 17 + [1], 0, 2), ([0, 1], 2, 2), ([0] * 65, 0, 1)]:
            with pytest.raises(api.RaymarchError):
                rm.set_bands(16, rank, nranks, owner)
        for band in (4, 12, 20):                                            # a band is whole rows of 8x8 tiles
            with pytest.raises(api.RaymarchError):
                rm.set_bands(band, 0, 2)


def test_batch_of_frames_in_one_launch(stock_scenes):
    """rm_set_batch: several frames per launch, their tiles in one pool, equal the frames rendered one
    by one: different cameras, caller-owned planes with a frame stride, row bands, three launches in a
    row (natural order, then the tile order learnt from the launch before)."""
    torch = pytest.importorskip("torch")
    w, h, n = 200, 120, 3
    infos = np.concatenate([O.default_info(k + 1, 7 * k) for k in range(n)])
    nodes = stock_scenes["chair"]
    for band, rank, nranks in [(0, 0, 1), (16, 1, 3)]:
        singles, masks = [], []
        with api.Raymarching(w, h) as rm:
            rm.set_nodes(nodes)
            rm.enable_hit_mask()
            if band:
                rm.set_bands(band, rank, nranks)
            rows = rm.local_rows()
            for k in range(n):
                rm.set_info(infos[k:k + 1])
                rm.process()
                singles.append(rm.read_pixels())
                masks.append(rm.read_hit_mask())
        assert not np.array_equal(singles[0], singles[1])
        stride, lstride = (rows + 3) * w * 4, (rows + 1) * w
        out = torch.zeros((n, rows + 3, w, 4), dtype=torch.uint8, device="cuda")
        lum = torch.zeros((n, rows + 1, w), dtype=torch.uint8, device="cuda")
        for jit_mode in (2, 0):                          # generated code, then the interpreter kernels
            with api.Raymarching(w, h) as rm:
                rm.set_jit(jit_mode)
                rm.set_nodes(nodes)
                if band:
                    rm.set_bands(band, rank, nranks)
                rm.set_batch(n, stride, lstride)
                rm.set_output(out.data_ptr(), w * 4)
                rm.set_luminance_output(lum.data_ptr(), w)
                rm.set_info_batch(infos)
                for launch in range(3):
                    out.zero_()
                    rm.process()
                    got = out.cpu().numpy()
                    for k in range(n):
                        assert np.array_equal(got[k, :rows], singles[k]), (band, launch, k)
                        assert not got[k, rows:].any()
                    assert np.array_equal(lum.cpu().numpy()[:, :rows], got[:, :rows, :, 2])
                # the context's own planes, packed, and the hit masks of frame 0
                rm.set_batch(2)
                rm.set_output(0, 0)
                rm.set_luminance_output(0, 0)
                rm.enable_hit_mask()
                rm.set_info_batch(infos[1:3])
                rm.process()
                assert np.array_equal(rm.read_pixels(), singles[1])
                assert np.array_equal(rm.read_pixels_of_frame(0), singles[1])
                assert np.array_equal(rm.read_pixels_of_frame(1), singles[2])
                with pytest.raises(api.RaymarchError):
                    rm.read_pixels_of_frame(2)
                assert np.array_equal(rm.read_hit_mask(), masks[1])
                with pytest.raises(api.RaymarchError):
                    rm.set_batch(33)
                with pytest.raises(api.RaymarchError):
                    rm.set_info_batch(infos)                 # three blocks for a batch of two


def test_scene_every_ray_leaves_after_one_step():
    """A unit sphere a million units away: every tile is one loop trip, so the launch is nothing but the
    tile queue.  (With the queue's counter bump written in C++ this was the shape of code that hung the
    device: the compiler split the wave into a lane-0 path and a rest-of-wave path.)  Several launches,
    one frame and four frames per launch, against the oracle."""
    table = scenes.Tree(("movo", (1e6, 0, 0), ("sfero", (1,)))).table()
    info = O.default_info(1, 0)
    w, h = 320, 200
    ref = O.render(table, w, h, info, threads=min(16, os.cpu_count() or 4))
    assert not ref["hit"].any()
    for batch in (1, 4):
        with api.Raymarching(w, h) as rm:
            rm.set_nodes(table)
            rm.set_batch(batch)
            rm.set_info(info)
            for launch in range(5):
                rm.process()
                assert np.array_equal(rm.read_pixels(), ref["rgba"]), (batch, launch)


def test_launches_that_are_not_full(stock_scenes):
    """rm_set_info_batch(n) with n below the batch renders n frames and touches nothing else: launches of
    5, 2, 5, 1, 3, 1, 5, 2, 4, 1, 4 frames on one context, so each runs on the tile order of a launch of another
    length: a longer one (entries of frames it does not have are skipped), a shorter one of which it is no
    multiple (2 -> 5: the tiles of the extra frames are in no list and go out behind the lists in natural
    order) or a shorter one of which it is (1 -> 3, 1 -> 5, 2 -> 4, 1 -> 4: every list entry stands for the same
    tile of the frames further on as well); every frame against the frames rendered one by one;
    the planes of the frames a launch does not render keep what they held."""
    torch = pytest.importorskip("torch")
    w, h, cap = 200, 120, 5
    nodes = stock_scenes["chair"]
    infos = np.concatenate([O.default_info(k + 1, 5 * k) for k in range(cap)])
    singles = []
    with api.Raymarching(w, h) as rm:
        rm.set_nodes(nodes)
        for k in range(cap):
            rm.set_info(infos[k:k + 1])
            rm.process()
            singles.append(rm.read_pixels())
    out = torch.zeros((cap, h, w, 4), dtype=torch.uint8, device="cuda")
    for jit_mode in (2, 0):
        with api.Raymarching(w, h) as rm:
            rm.set_jit(jit_mode)
            rm.set_nodes(nodes)
            rm.set_output(out.data_ptr(), w * 4)
            rm.set_batch(cap, h * w * 4, 0)
            for n in (5, 2, 5, 1, 3, 1, 5, 2, 4, 1, 4):
                out.fill_(0x5a)
                order = list(range(cap))[-n:] + list(range(cap))[:-n]          # another camera per slot each time
                rm.set_info_batch(infos[order[:n]])
                rm.process()
                got = out.cpu().numpy()
                for slot in range(n):
                    assert np.array_equal(got[slot], singles[order[slot]]), (jit_mode, n, slot)
                assert (got[n:] == 0x5a).all(), (jit_mode, n)
            rm.set_info(infos[0:1])                                       # all five frames again, one camera
            rm.process()
            got = out.cpu().numpy()
            for slot in range(cap):
                assert np.array_equal(got[slot], singles[0])


def test_incomplete_frame_is_reported(stock_scenes):
    """A wavefront that leaves on its watchdog must not pass for a finished frame: with the watchdog at
    one tick every tile of more than a few trips is abandoned, rm_process / rm_synchronize / rm_read_pixels
    return RM_ERR_INCOMPLETE, and the next frame (watchdog restored) is whole and in order again."""
    info = O.default_info(1, 0)
    w, h = 320, 200
    ref = O.render(stock_scenes["chair"], w, h, info, threads=min(16, os.cpu_count() or 4))
    for jit_mode in (2, 0):
        with api.Raymarching(w, h) as rm:
            rm.set_jit(jit_mode)
            rm.set_nodes(stock_scenes["chair"])
            rm.set_info(info)
            rm.enable_hit_mask(True)
            rm.process()
            assert_same_frame(rm.read_pixels(), rm.read_hit_mask(), ref, "before_watchdog")
            rm.set_boost_age(4)                     # the clock is looked at where the priority steps up: trip 5
            rm.set_watchdog(1)
            with pytest.raises(api.RaymarchError) as e:
                rm.process()
            assert e.value.status == -10
            done, launched, flag = rm.last_check()       # what the check saw: fewer tiles than launched, the watchdog's flag
            assert flag == 1 and done != launched
            rm.process_async()
            with pytest.raises(api.RaymarchError) as e:
                rm.synchronize()
            assert e.value.status == -10
            rm.process_async()
            with pytest.raises(api.RaymarchError) as e:
                rm.read_pixels()
            assert e.value.status == -10
            rm.set_watchdog(0)
            for k in range(3):                      # natural order, then the order learnt from it
                rm.process()
                assert_same_frame(rm.read_pixels(), rm.read_hit_mask(), ref, "after_watchdog_%d" % k)


def test_two_processes_share_the_gpu():
    """Two processes rendering on this GPU at once (tools/two_process_stress.py): their wavefronts are preempted for
    each other, and none may be stopped for it - with the watchdog on the per-XCD shader-cycle counter (rounds 1-2)
    a preempted wavefront could find the counter behind its starting value and give up, which showed as
    RM_ERR_INCOMPLETE in the 2-rank rehearsal (profiles/r03_two_process_stress.txt)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "two_process_stress.py"), "2", "360"],
                       capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("pid ")]
    assert len(lines) == 2 and all("360 launches, 0 checks failed" in ln for ln in lines), r.stdout


def test_external_output_and_pitch(stock_scenes):
    """rm_set_output: render into a caller-owned device buffer with a padded pitch."""
    torch = pytest.importorskip("torch")
    info = O.default_info(1, 0)
    w, h, pitch = 100, 60, 512
    buf = torch.zeros((h, pitch), dtype=torch.uint8, device="cuda:0")
    with api.Raymarching(w, h) as rm:
        rm.set_nodes(stock_scenes["chair"])
        rm.set_info(info)
        rm.set_output(buf.data_ptr(), pitch)
        rm.set_stream(torch.cuda.current_stream().cuda_stream)
        rm.process()
    torch.cuda.synchronize()
    got = buf.cpu().numpy()[:, : w * 4].reshape(h, w, 4)
    ref = O.render(stock_scenes["chair"], w, h, info)
    assert np.array_equal(got, ref["rgba"])
    assert not buf.cpu().numpy()[:, w * 4:].any(), "bytes beyond width*4 must stay untouched"


def test_read_pixels_host_pitch(stock_scenes):
    """rm_read_pixels / rm_read_hit_mask into a host frame with padded rows (the 2-D copy) and into a
    contiguous one (the linear copy): same pixels, padding untouched."""
    import ctypes as C
    info = O.default_info(1, 0)
    w, h = 100, 60
    with api.Raymarching(w, h) as rm:
        rm.set_nodes(stock_scenes["chair"])
        rm.set_info(info)
        rm.enable_hit_mask(True)
        rm.process()
        tight, tight_hit = rm.read_pixels(), rm.read_hit_mask()
        wide = np.full((h, 512), 0xab, np.uint8)
        api._check(api.lib().rm_read_pixels(rm._ctx, C.c_void_p(wide.ctypes.data), 512), "rm_read_pixels")
        wide_hit = np.full((h, 128), 0xab, np.uint8)
        api._check(api.lib().rm_read_hit_mask(rm._ctx, C.c_void_p(wide_hit.ctypes.data), 128), "rm_read_hit_mask")
        assert api.lib().rm_read_pixels(rm._ctx, C.c_void_p(wide.ctypes.data), w * 4 - 1) == -1
    ref = O.render(stock_scenes["chair"], w, h, info)
    assert np.array_equal(tight, ref["rgba"])
    assert np.array_equal(wide[:, : w * 4].reshape(h, w, 4), ref["rgba"]) and (wide[:, w * 4:] == 0xab).all()
    assert np.array_equal(tight_hit, ref["hit"])
    assert np.array_equal(wide_hit[:, :w], ref["hit"]) and (wide_hit[:, w:] == 0xab).all()


def test_errors_are_reported(stock_scenes):
    with api.Raymarching(64, 64) as rm:
        with pytest.raises(api.RaymarchError) as e:
            rm.process()                      # nothing set yet
        assert e.value.status == -6
        bad = stock_scenes["chair"].copy()
        bad[4]["type"] = 0                    # nenio reachable: the reference would call a null pointer
        with pytest.raises(api.RaymarchError) as e:
            rm.set_nodes(bad)
        assert e.value.status == -2


@pytest.mark.parametrize("feedback", [True, False])
def test_heavy_first_feedback_never_changes_pixels(stock_scenes, feedback):
    """The scheduler orders tiles by what the previous frame measured (device_render.h).
    Whatever the history holds - none, the same camera, another camera, another tile
    numbering - every frame must still be the oracle's frame."""
    from raytracing_amd import tiling
    w, h = 333, 190
    infos = [O.default_info(1, 0), O.default_info(1, 0), O.default_info(2, 3), O.default_info(2, 3), O.default_info(40, 90)]
    refs = {}
    with api.Raymarching(w, h) as rm:
        rm.set_nodes(stock_scenes["chair"])
        rm.set_heavy_first(feedback)
        rm.enable_hit_mask(True)
        for k, info in enumerate(infos):
            key = info.tobytes()
            if key not in refs:
                refs[key] = O.render(stock_scenes["chair"], w, h, info, threads=min(16, os.cpu_count() or 4))
            rm.set_info(info)
            rm.process()
            assert np.array_equal(rm.read_pixels(), refs[key]["rgba"]), "frame %d" % k
            assert np.array_equal(rm.read_hit_mask(), refs[key]["hit"]), "frame %d" % k
        want = refs[infos[-1].tobytes()]["rgba"]
        for nranks, rank in [(3, 1), (2, 0)]:    # new tile numbering: the history must be dropped
            rm.set_bands(16, rank, nranks)
            rows = [y for y in range(h) if (y // 16) % nranks == rank]
            for k in range(3):
                rm.process()
                assert np.array_equal(rm.read_pixels(), want[rows]), "banded frame %d" % k
        rm.set_bands(0, 0, 1)
        for k in range(2):
            rm.process()
            assert np.array_equal(rm.read_pixels(), want)


def test_many_frames_infinity(stock_scenes):
    """Back-to-back frames on one context (the three scheduler sets rotate): every one identical."""
    info = O.default_info(1, 0)
    ref = O.render(stock_scenes["infinity"], 200, 120, info, threads=min(16, os.cpu_count() or 4))
    with api.Raymarching(200, 120) as rm:
        rm.set_nodes(stock_scenes["infinity"])
        rm.set_info(info)
        rm.set_boost_age(4)
        for k in range(7):
            rm.process_async()
        rm.synchronize()
        assert np.array_equal(rm.read_pixels(), ref["rgba"])
        rm.set_boost_age(0xffffffff)
        rm.process()
        assert np.array_equal(rm.read_pixels(), ref["rgba"])


def test_short_sqrt_and_quotient_are_ieee_for_every_input():
    """device_math.h sqrt_exact / div_const against IEEE sqrtf and '/' on the device, all 2^32
    bit patterns (argument; numerator per divisor): zero mismatches, NaNs compared as equal."""
    divisors = [260000.0, 10000.0, 54.0, 2.5, 1.5, 0.7, 3.0, 1e-3, 123456.789, 7.0, 0.1, -2.0, 1.0, 2.0 ** -20, 2.0 ** 20,
                np.float32(np.nextafter(np.float32(2.0), np.float32(0.0)))]       # the last one: mantissa all ones
    with api.Raymarching(8, 8) as rm:
        sq, dv = rm.selftest_math(divisors)
    assert sq == 0
    assert dv == [0] * len(divisors)


def test_divisors_are_proved_per_scene(stock_scenes):
    info = O.default_info(1, 0)
    ref = O.render(stock_scenes["infinity"], 160, 100, info, threads=min(16, os.cpu_count() or 4))
    for short, path in [(True, "jit"), (False, "jit"), (True, "program"), (False, "program")]:
        with api.Raymarching(160, 100) as rm:
            rm.set_short_math(short)
            rm.set_jit(PATHS[path][0])
            rm.set_nodes(stock_scenes["infinity"])
            assert rm.scene_stats()["exact_divisors"] == (1 if short else 0)     # dot(a,a) = 10000 three times
            rm.set_info(info)
            rm.process()
            assert np.array_equal(rm.read_pixels(), ref["rgba"])


def test_camera_sweep_120_frames(stock_scenes, golden_dir):
    """config 5's camera path (tests/golden/sweep_rays_info.bin), every frame against the oracle at a
    size it renders in a blink; one context, so every frame runs on the previous frame's tile order."""
    sweep = np.fromfile(os.path.join(golden_dir, "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)
    assert len(sweep) == 120
    w, h = 192, 108
    with api.Raymarching(w, h) as rm:
        rm.set_nodes(stock_scenes["chair"])
        for k in range(120):
            rm.set_info(sweep[k:k + 1])
            rm.process()
            got = rm.read_pixels()
            ref = O.render(stock_scenes["chair"], w, h, sweep[k:k + 1], threads=min(16, os.cpu_count() or 4))
            assert np.array_equal(got, ref["rgba"]), "sweep frame %d" % k


def test_8k_frame_rows(stock_scenes, golden_dir):
    """7680x4320 (config 5's size): the last sweep frame, sampled rows against the oracle, and the
    same frame rendered in 8 ranks' bands."""
    from raytracing_amd import tiling
    sweep = np.fromfile(os.path.join(golden_dir, "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)
    info = sweep[119:120]
    w, h = 7680, 4320
    with api.Raymarching(w, h) as rm:
        rm.set_nodes(stock_scenes["chair"])
        rm.set_info(info)
        rm.process()
        rm.process()                       # second frame: tile order from the first
        whole = rm.read_pixels()
        for y0 in (0, 1733, 2160, 2161, 2987, 4319):
            ref = O.render(stock_scenes["chair"], w, h, info, rows=(y0, y0 + 1))
            assert np.array_equal(whole[y0], ref["rgba"][y0]), y0
        parts = []
        for rank in range(8):
            rm.set_bands(16, rank, 8)
            rm.process()
            parts.append(rm.read_pixels())
        assert np.array_equal(tiling.assemble(parts, h, 16), whole)


@pytest.mark.parametrize("name", ["chair", "infinity"])
def test_front_end_ppm_matches_reference_hash(name, golden_dir, tmp_path):
    """The headless front end with the reference's command line (Source/main.cpp:18-39) and default
    camera writes the PPM whose sha256 the reference's own march produced host-side."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / (name + ".ppm"))
    exe = os.path.join(root, "raytracing_amd", "rm_render")
    run = subprocess.run([exe, "--width", "640", "--height", "480", "--input",
                          os.path.join(golden_dir, "scenes", name + ".sce"), "--out", out],
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr + run.stdout
    with open(out, "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == GOLDEN_PPM[name]


@pytest.mark.parametrize("gpus,height", [(2, 480), (3, 250), (8, 1080)])
def test_front_end_several_gpu_contexts(gpus, height, stock_scenes, golden_dir, tmp_path):
    """rm_render --gpus N: the frame dealt in 8-row bands to N contexts (here all on this GPU, --devices 0,0,...:
    the box has one), three frames along config 5's camera path, put together on the host.  The PPM must be the
    one-context front end's, which is the oracle's frame; 250 rows leave a last band of 10 and ranks of unequal
    height, 1080 rows give rank 0 one band more than the others."""
    import subprocess
    exe = os.path.join(ROOT, "raytracing_amd", "rm_render")
    sweep = np.fromfile(os.path.join(golden_dir, "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)
    w, frames = 320, 3
    out = str(tmp_path / "bands.ppm")
    run = subprocess.run([exe, "--width", str(w), "--height", str(height), "--input", os.path.join(golden_dir, "scenes", "chair.sce"),
                          "--frames", str(frames), "--drag-x", "1", "--gpus", str(gpus), "--devices", ",".join(["0"] * gpus), "--out", out],
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr + run.stdout
    assert "on %d GPU context(s)" % gpus in run.stdout
    ref = O.render(stock_scenes["chair"], w, height, sweep[frames - 1:frames], threads=min(16, os.cpu_count() or 4))
    assert ref["hit"].sum() > 1000
    with open(out, "rb") as f:
        assert f.read() == O.ppm_bytes(ref["rgba"])


@pytest.mark.parametrize("frames,extra", [(70, []), (25, ["--frame-by-frame"])])
def test_front_end_scripted_sweep(frames, extra, stock_scenes, golden_dir, tmp_path):
    """--frames K --drag-x 1 is config 5's camera path: the front end's own float
    maths (Source/main.cpp:240-266) must land on the committed rays_info of frame K-1, so the PPM of
    the last frame equals the oracle's frame for that block.  The camera path is known ahead, so the
    front end hands it to hip::raymarching::ProcessBatch (70 frames: launches of 24, 23 and 23);
    --frame-by-frame is the reference's loop, a SetInfo and a Process() per frame."""
    import subprocess
    sweep = np.fromfile(os.path.join(golden_dir, "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)
    out = str(tmp_path / "sweep.ppm")
    w, h = 320, 180
    run = subprocess.run([os.path.join(ROOT, "raytracing_amd", "rm_render"), "--width", str(w), "--height", str(h), "--input",
                          os.path.join(golden_dir, "scenes", "chair.sce"), "--frames", str(frames), "--drag-x", "1", "--out", out] + extra,
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr + run.stdout
    ref = O.render(stock_scenes["chair"], w, h, sweep[frames - 1:frames], threads=min(16, os.cpu_count() or 4))
    with open(out, "rb") as f:
        assert f.read() == O.ppm_bytes(ref["rgba"])


def test_front_end_event_script(stock_scenes, golden_dir, tmp_path):
    """rm_render --script: every event of the reference's loop (Source/main.cpp:148-266: light keys, reset,
    pause, wheel, middle-button pan, right-button drag) from a file; the front end's own float maths must
    land on the oracle's camera block for the last frame, and the PPM on the oracle's frame for it."""
    import subprocess
    script = ["frame", "wheel 2", "drag 14 -9", "frame", "pan 30 -12", "left", "left", "frame", "pause", "right",
              "frame", "space", "wheel -38", "drag -25 6", "pan -8 3", "resume", "frame  # the one that is written"]
    path = tmp_path / "events.txt"
    path.write_text("\n".join(script) + "\n")
    out = str(tmp_path / "script.ppm")
    w, h = 320, 180
    run = subprocess.run([os.path.join(ROOT, "raytracing_amd", "rm_render"), "--width", str(w), "--height", str(h), "--input",
                          os.path.join(golden_dir, "scenes", "chair.sce"), "--script", str(path), "--out", out],
                         capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr + run.stdout
    assert "5 frame(s)" in run.stdout
    infos = O.camera_script(script)
    assert len(infos) == 5
    ref = O.render(stock_scenes["chair"], w, h, infos[4:5], threads=min(16, os.cpu_count() or 4))
    assert ref["hit"].sum() > 1000                                # the script ends looking at the scene
    with open(out, "rb") as f:
        assert f.read() == O.ppm_bytes(ref["rgba"])


def test_bench_two_ranks_on_one_gpu(golden_dir):
    """bench.py's N>1 path end to end, two ranks sharing this GPU (RM_BENCH_REHEARSAL: gloo gather
    through host copies instead of RCCL): launches of several frames, luminance planes gathered,
    frames rebuilt on rank 0; a last launch that is not full.  The frame must be the golden one."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RM_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "13", "--warmup", "3",
           "--frames-per-launch", "4", "--workload", "chair_1080p", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    if r.returncode != 0:                # the ranks' own tracebacks come before the launcher's summary
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_two_ranks_stderr.txt"), "w") as f:
            f.write(r.stderr)
    assert r.returncode == 0, r.stderr[-6000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["steps"] == 13 and rec["launches"] == 4
    assert rec["parity"]["frame_matches_golden"] is True
    # what the first run on a real 8-GPU node will be read by: ranks, devices, per-rank rates, gather and rebuild times
    assert rec["rccl_ranks"] == 2 and rec["backend"] == "gloo" and rec["incomplete_ranks"] == []
    assert [r["rank"] for r in rec["ranks"]] == [0, 1] and rec["distinct_devices"] == 1       # (the rehearsal shares one GPU)
    shares = rec["config"]["band_deal_measured"]["rank0_share_tried"]
    for r in rec["ranks"]:
        assert len(r["render_only_frames_per_s_by_rank0_share"]) == len(shares) and min(r["render_only_frames_per_s_by_rank0_share"]) > 0
        assert r["gather_ms_per_launch_median"] > 0 and r["device_name"]
    assert rec["ranks"][0]["rebuild_ms_per_launch_median"] > 0 and rec["ranks"][1]["rebuild_ms_per_launch_median"] is None
    assert len(rec["config"]["band_weights"]) == 2 and len(rec["block_ms"]) == rec["repeats"] == 7


def test_bench_gather_path_over_rccl_with_one_rank(golden_dir):
    """bench.py's N>1 path with ONE rank and the real collective (RM_BENCH_SOLO_GATHER): bands, luminance planes,
    three contexts taking launches in turn, an RCCL gather of the one part to itself, the frame rebuilt from it.
    What a one-GPU box can check of the RCCL leg: that the collective takes the tensors this path hands it and
    that the golden frame comes out.  Ranks exchanging bytes over xGMI remain untested until an 8-GPU run."""
    import json
    import subprocess
    import sys
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RM_BENCH_SOLO_GATHER="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "13", "--warmup", "3",
           "--frames-per-launch", "4", "--workload", "chair_1080p", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["steps"] == 13 and rec["config"]["contexts_in_flight"] == 3
    assert rec["rccl_ranks"] == 1 and rec["backend"] == "nccl" and rec["incomplete_ranks"] == []
    assert "row bands" in rec["config"]["partition"]
    assert rec["parity"]["frame_matches_golden"] is True
