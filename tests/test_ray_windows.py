"""Ray windows (include/raymarching_amd.h): the volume the host derives from a node tree must bound the
scene's distance function from below, and a march confined to the window must render the frame the
reference's march renders.  CPU suite: the bound comes from the product's host-only rm_scene_bound_of,
distances and both marches from the oracle (its windowed march restates the device's ray_window).
The GPU side of the same claim is every frame test of test_gpu_parity.py (windows are on by default there)
plus test_ray_windows_change_no_pixel."""
import os

import numpy as np
import pytest

import kat_scenes
from oracle import oracle as O
from raytracing_amd import api, scenes

THREADS = min(8, os.cpu_count() or 4)
MARGIN = 8.0          # RM_WINDOW_MARGIN (include/rm_types.h)


def bound_of(nodes):
    b = api.scene_bound(nodes)
    return b.as_dict(), O.bound_record(b.kind, list(b.c), list(b.u), b.r)


def distance_to_volume(b, pts):
    """dist(x, centre) or dist(x, axis) for every point, in double precision."""
    c, u = np.array(b["c"], dtype=np.float64), np.array(b["u"], dtype=np.float64)
    d = pts.astype(np.float64) - c
    if b["kind"] == api.BOUND_CYLINDER:
        d = d - np.outer(d @ u, u)
    return np.sqrt((d * d).sum(axis=1))


def test_bounds_of_known_scenes(stock_scenes):
    chair, _ = bound_of(stock_scenes["chair"])
    # senfina_ripeto a=(0,500,100) over movo(200,0,0) over a rotated komunajo of kubo(50,50,50): the cube's
    # corner sphere around the lattice's axis through (200,0,0)
    assert chair["kind"] == api.BOUND_CYLINDER
    assert np.allclose(chair["c"], (200, 0, 0), atol=1e-3) and np.allclose(chair["u"], np.array((0, 500, 100)) / np.hypot(500, 100), atol=1e-6)
    assert 50 * np.sqrt(3) < chair["r"] < 50 * np.sqrt(3) + 0.1
    assert bound_of(stock_scenes["infinity"])[0]["kind"] == api.BOUND_NONE          # a komplemento at the top fills space
    csg, _ = bound_of(scenes.csg1024())
    assert csg["kind"] == api.BOUND_SPHERE and abs(csg["c"][0] - 200) < 20 and 90 < csg["r"] < 125
    tables = kat_scenes.tables()
    kinds = {name: bound_of(t)[0]["kind"] for name, t in tables.items()}
    for name in ("sfero", "kubo", "cilindro", "kunigajo_2", "kunigajo_3", "kunigajo_4", "komunajo_2", "komunajo_3", "komunajo_4",
                 "glata_kunigajo_2", "glata_komunajo_2", "movo", "rotacioX", "rotacioY", "rotacioZ", "rotacioQ",
                 "mixed_kinds", "carved", "nested_frames", "deep_values", "glata_nested", "chain_same_kind"):
        assert kinds[name] == api.BOUND_SPHERE, name
    assert kinds["senfina_ripeto"] == api.BOUND_CYLINDER
    for name in ("ebeno", "komplemento", "portanta_sfero", "portanta_nested", "ripeto_frames", "neg_wrappers"):
        assert kinds[name] == api.BOUND_NONE, name
    b, _ = bound_of(tables["sfero"])
    assert np.allclose(b["c"], 0) and 5 <= b["r"] < 5.02
    b, _ = bound_of(tables["carved"])              # komunajo: the bound of its tightest operand, the cube
    assert np.allclose(b["c"], 0, atol=1e-6) and 6 * np.sqrt(3) <= b["r"] < 6 * np.sqrt(3) + 0.02


def _trees():
    out = [("kat_" + name, t) for name, t in kat_scenes.tables().items()]
    out += [("fuzz_%d" % seed, scenes.Tree(kat_scenes.random_tree(1000 + seed, depth=3 + seed % 4)).table()) for seed in range(48)]
    # random trees without the node types that never have a volume, so that every seed exercises the bound
    out += [("bounded_%d" % seed, scenes.Tree(_bounded_tree(seed)).table()) for seed in range(48)]
    return out


def _bounded_tree(seed, depth=4):
    rng = np.random.default_rng(7000 + seed)

    def f(lo, hi):
        return float(np.float32(rng.uniform(lo, hi)))

    def leaf():
        return [("sfero", (f(1, 6),)), ("kubo", (f(1, 5), f(1, 5), f(1, 5))), ("cilindro", (f(1, 4), f(1, 5)))][rng.integers(0, 3)]

    def moved(t, span=8):
        return ("movo", (f(-span, span), f(-span, span), f(-span, span)), t)

    def build(d):
        if d == 0:
            return leaf()
        pick = rng.integers(0, 12)
        if pick < 3:
            n = int(rng.integers(2, 5))
            return ("kunigajo_%d" % n, ()) + tuple(moved(build(d - 1)) for _ in range(n))
        if pick < 5:
            n = int(rng.integers(2, 5))
            return ("komunajo_%d" % n, (), ("sfero", (f(6, 10),))) + tuple(moved(build(d - 1), 2) for _ in range(n - 1))
        if pick == 5:                             # a carved solid: the komplemento itself has no volume, the solid has
            return ("komunajo_2", (), ("kubo", (f(5, 9), f(5, 9), f(5, 9))), ("komplemento", (), moved(build(d - 1))))
        if pick == 6:
            return ("glata_kunigajo_2", (f(-3, 3) or 1.0,), build(d - 1), moved(build(d - 1)))
        if pick == 7:
            return ("glata_komunajo_2", (f(-3, 3) or 1.0,), ("sfero", (f(6, 10),)), moved(build(d - 1), 2))
        if pick == 8:
            return moved(build(d - 1), 20)
        if pick in (9, 10):
            a = f(-3, 3)
            return ("rotacio" + "XYZ"[rng.integers(0, 3)], (np.float32(np.cos(a)), np.float32(np.sin(a))), build(d - 1))
        return ("rotacioQ", scenes.quaternion((f(-1, 1), f(-1, 1), f(0.1, 1)), f(-3, 3)), build(d - 1))

    t = build(depth)
    if seed % 3 == 0:                             # a lattice on top: the volume is a cylinder
        t = ("senfina_ripeto", (f(-60, 60), f(30, 80), f(-60, 60)), moved(t, 30))
    return t


@pytest.mark.parametrize("label,table", _trees(), ids=[k for k, _ in _trees()])
def test_volume_bounds_the_distance_function(label, table):
    """DIST(root, x) >= dist(x, centre or axis) - r at seeded points near and far, for every tree that has a
    volume (scene_bound.cpp's claim, node type by node type)."""
    b, _ = bound_of(table)
    if label.startswith("bounded_"):
        assert b["kind"] != api.BOUND_NONE
    if b["kind"] == api.BOUND_NONE:
        return
    rng = np.random.default_rng(abs(hash(label)) % (1 << 32))
    c = np.array(b["c"])
    pts = [c + rng.uniform(-1, 1, size=(120, 3)) * (b["r"] + 1) * s for s in (0.5, 1.5, 4, 40, 400)]
    if b["kind"] == api.BOUND_CYLINDER:           # and along the axis, where the lattice's copies are
        along = np.outer(rng.uniform(-3000, 3000, size=240), np.array(b["u"]))
        pts.append(c + along + rng.uniform(-1, 1, size=(240, 3)) * (b["r"] + 1) * 2)
    pts = np.concatenate(pts).astype(np.float32)
    dist = O.dist_many(table, 0, pts).astype(np.float64)
    lower = distance_to_volume(b, pts) - b["r"]
    worst = np.min(dist - lower)
    # rounding of DIST itself (fp32, coordinates up to ~10^4 here): far below the window's margin
    assert worst > -0.02, "%s: DIST is %.4f below the volume's bound at %s" % (label, -worst, pts[np.argmin(dist - lower)])


def _camera(light_steps=1, drag=0, pos=None, zoom=None):
    info = O.default_info(light_steps, drag)
    if pos is not None:
        info["StartPos"] = pos
    if zoom is not None:
        for v in ("StartWVec", "StartHVec"):       # the field of view, as test_gpu_parity.py's fuzz camera widens it
            for a in "xyz":
                info[v][a] *= np.float32(zoom)
    return info


def _same(nodes, w, h, info, window, label):
    a = O.render(nodes, w, h, info, threads=THREADS, want_evals=True, want_counters=True)
    b = O.render(nodes, w, h, info, threads=THREADS, want_evals=True, want_counters=True, window=window)
    assert np.array_equal(a["rgba"], b["rgba"]) and np.array_equal(a["hit"], b["hit"]), label
    assert np.all(b["evals"] <= a["evals"]), label
    return a, b


def test_windowed_march_renders_the_same_frame_chair(stock_scenes):
    nodes = stock_scenes["chair"]
    _, window = bound_of(nodes)
    saved = []
    for label, info in [("default", _camera()), ("dragged", _camera(7, 200)), ("sideways", _camera(3, 120)),
                        ("inside", _camera(pos=(200.0, 10.0, 5.0))), ("on_the_axis", _camera(2, 180, pos=(200.0, -300.0, -60.0))),
                        ("far_below", _camera(pos=(0.0, 0.0, -2500.0))), ("wide", _camera(zoom=6.0)), ("narrow", _camera(zoom=0.2))]:
        a, b = _same(nodes, 320, 240, info, window, label)
        saved.append((label, int(a["evals"].sum()), int(b["evals"].sum()), int(a["hit"].sum())))
    by = {k: v for k, *v in saved}
    assert by["dragged"][1] < by["dragged"][0] and by["wide"][1] < 0.5 * by["wide"][0], saved      # the window does cut work
    assert by["default"][2] > 0 and by["inside"][2] > 0, saved                                   # and the scene is in view


def test_windowed_march_renders_the_same_frame_other_scenes():
    info = _camera(2, 3, pos=(-45.0, 2.0, 1.0), zoom=4.0)           # test_gpu_parity.py's fuzz camera
    seen = 0
    for label, table in _trees():
        b, window = bound_of(table)
        if b["kind"] == api.BOUND_NONE or label.startswith("fuzz_") and int(label[5:]) >= 12 or label.startswith("bounded_") and int(label[8:]) >= 12:
            continue
        a, w = _same(table, 96, 64, info, window, label)
        seen += int(a["hit"].sum()) > 0
    assert seen >= 20
    csg = scenes.csg1024()
    a, w = _same(csg, 128, 72, _camera(), bound_of(csg)[1], "csg1024")
    assert a["hit"].sum() > 0


def test_rays_from_far_out_get_no_window(stock_scenes):
    """Beyond RM_WINDOW_MAX_COORD the margin's arithmetic no longer covers the march's rounding: such rays are
    marched as the reference marches them, to the last step."""
    nodes = stock_scenes["chair"]
    _, window = bound_of(nodes)
    info = _camera(pos=(-20000.0, 0.0, 0.0))
    a, b = _same(nodes, 64, 48, info, window, "far out")
    assert np.array_equal(a["evals"], b["evals"])
    info = _camera(pos=(-16000.0, 0.0, 0.0))                       # inside the limit: windows apply
    a, b = _same(nodes, 64, 48, info, window, "far, inside the limit")
    assert b["evals"].sum() < a["evals"].sum()


def test_a_volume_that_is_too_small_is_caught_by_these_tests(stock_scenes):
    """The tests above can fail: with a volume that does not bound the scene (the chair's cylinder at a third
    of its radius) the windowed march loses pixels."""
    nodes = stock_scenes["chair"]
    b, _ = bound_of(nodes)
    small = O.bound_record(b["kind"], b["c"], b["u"], b["r"] / 3 - MARGIN)
    info = _camera()
    a = O.render(nodes, 320, 240, info, threads=THREADS)
    w = O.render(nodes, 320, 240, info, threads=THREADS, window=small)
    assert not np.array_equal(a["rgba"], w["rgba"])
