"""Small scenes for node-level known-answer tests: every node type at least once,
plus nestings that exercise each path of the flattener (fused leaves, running
min/max chains, register and LDS value slots, saved frames, the bounding-sphere
skip) and the tie-breaking rules of the combinators' normals."""
import numpy as np

from raytracing_amd.scenes import Tree, quaternion

S5 = ("sfero", (5,))
K345 = ("kubo", (3, 4, 5))
C34 = ("cilindro", (3, 4))
E = ("ebeno", (0.3, 0.5, 0.8))
Q = quaternion((1.0, 30.0, 1.0), -1.8)
Q2 = quaternion((0.3, -0.2, 0.9), 0.7)
COS, SIN = np.float32(np.cos(0.6)), np.float32(np.sin(0.6))

SCENES = {
    "sfero": S5,
    "kubo": K345,
    "cilindro": C34,
    "ebeno": E,
    "portanta_sfero": ("portanta_sfero", (1, 2, 3, 6), ("kubo", (2, 2, 2))),
    "kunigajo_2": ("kunigajo_2", (), ("movo", (4, 0, 0), S5), K345),
    "kunigajo_3": ("kunigajo_3", (), ("movo", (4, 0, 0), S5), K345, ("movo", (0, -6, 1), C34)),
    "kunigajo_4": ("kunigajo_4", (), ("movo", (4, 0, 0), S5), K345, ("movo", (0, -6, 1), C34), ("movo", (0, 0, 7), S5)),
    "komunajo_2": ("komunajo_2", (), ("movo", (2, 0, 0), S5), K345),
    "komunajo_3": ("komunajo_3", (), ("movo", (2, 0, 0), S5), K345, ("movo", (0, -1, 1), C34)),
    "komunajo_4": ("komunajo_4", (), ("movo", (2, 0, 0), S5), K345, ("movo", (0, -1, 1), C34), ("movo", (0, 0, 2), S5)),
    "komplemento": ("komplemento", (), S5),
    "glata_kunigajo_2": ("glata_kunigajo_2", (2.5,), S5, ("movo", (5, 1, 0), K345)),
    "glata_komunajo_2": ("glata_komunajo_2", (1.5,), S5, ("movo", (2, 1, 0), K345)),
    "movo": ("movo", (1, -2, 3), C34),
    "rotacioX": ("rotacioX", (COS, SIN), K345),
    "rotacioY": ("rotacioY", (COS, SIN), K345),
    "rotacioZ": ("rotacioZ", (COS, SIN), K345),
    "rotacioQ": ("rotacioQ", Q, ("movo", (1, 0, 2), K345)),
    "senfina_ripeto": ("senfina_ripeto", (7, 1, 2), S5),
    # ---- flattener paths
    "tie_2": ("kunigajo_2", (), S5, S5),
    "tie_3": ("kunigajo_3", (), S5, S5, S5),
    "tie_max_2": ("komunajo_2", (), K345, K345),
    "tie_max_4": ("komunajo_4", (), K345, K345, K345, K345),
    "neg_wrappers": ("komplemento", (), ("movo", (1, 1, 1), ("komplemento", (), ("komplemento", (), K345)))),
    "two_movos": ("movo", (1, 0, 0), ("movo", (0, 2, 0), S5)),
    "chain_same_kind": ("kunigajo_2", (), ("kunigajo_2", (), ("movo", (6, 0, 0), S5), K345),
                        ("kunigajo_3", (), ("movo", (-6, 0, 0), S5), ("movo", (0, 7, 0), C34), ("movo", (0, 0, -8), S5))),
    "mixed_kinds": ("komunajo_2", (), ("kunigajo_2", (), ("movo", (2, 0, 0), S5), K345),
                    ("komplemento", (), ("kunigajo_2", (), ("movo", (0, 0, -3), ("sfero", (2,))), ("movo", (0, 0, 3), ("sfero", (1.5,)))))),
    "carved": ("komunajo_2", (), ("kubo", (6, 6, 6)),
               ("komplemento", (), ("kunigajo_3", (), ("movo", (5, 5, 0), S5), ("movo", (-6, 0, 4), ("sfero", (3,))), ("movo", (0, -6, 0), C34)))),
    # sphere_skip of the generated code: every carved-out solid a sphere (under a movo with +0 and -0 offsets, a
    # rotation's axis handed through), spheres that reach out through the box's faces, a second open member
    "carved_spheres": ("komunajo_4", (), ("kubo", (6, 5, 6)), ("movo", (0, 0, 1), ("sfero", (8.5,))),
                       ("komplemento", (), ("movo", (0, 0, -6), ("sfero", (4,)))),
                       ("komplemento", (), ("kunigajo_2", (), ("rotacioZ", (COS, SIN), ("movo", (-0.0, 5.5, 0), ("sfero", (2.25,)))),
                                            ("movo", (6, 0, 3), ("rotacioX", (COS, SIN), ("sfero", (1.5,))))))),
    "nested_frames": ("kunigajo_2", (),
                      ("movo", (3, 0, 0), ("komunajo_2", (), K345,
                                           ("rotacioX", (COS, SIN), ("kunigajo_2", (), S5, ("movo", (0, 4, 0), ("sfero", (3,))))))),
                      ("rotacioQ", Q2, ("kunigajo_2", (), ("movo", (-5, 0, 1), C34),
                                        ("rotacioZ", (COS, SIN), ("komunajo_2", (), ("movo", (-4, 0, 0), K345), ("sfero", (7,))))))),
    "deep_values": ("komunajo_2", (), ("sfero", (9,)),
                    ("kunigajo_2", (), ("movo", (1, 0, 0), K345),
                     ("komunajo_2", (), ("sfero", (8,)),
                      ("kunigajo_2", (), ("movo", (0, 1, 0), C34),
                       ("komunajo_2", (), ("sfero", (7,)), ("glata_kunigajo_2", (1.0,), ("movo", (0, 0, 2), S5), K345)))))),
    "glata_nested": ("glata_kunigajo_2", (2.0,),
                     ("glata_komunajo_2", (1.0,), ("movo", (1, 0, 0), S5), K345),
                     ("rotacioY", (COS, SIN), ("glata_kunigajo_2", (0.7,), ("movo", (0, 5, 0), C34), ("movo", (0, -5, 0), S5)))),
    "portanta_nested": ("kunigajo_2", (),
                        ("portanta_sfero", (4, 0, 0, 4), ("kunigajo_2", (), ("kubo", (2, 2, 2)), ("movo", (0, 1, 0), ("sfero", (2.5,))))),
                        ("portanta_sfero", (-4, 0, 0, 3), ("rotacioZ", (COS, SIN), ("komunajo_2", (), ("kubo", (2, 1, 1)), ("sfero", (2,)))))),
    "ripeto_frames": ("senfina_ripeto", (9, 0, 0), ("senfina_ripeto", (0, 11, 1),
                      ("kunigajo_2", (), ("movo", (1, 1, 0), ("sfero", (2,))), ("rotacioQ", Q2, ("kubo", (1, 2, 3)))))),
}


def random_tree(seed, depth=5):
    """A seeded random scene over all 20 node types: each level picks a combinator or a transform
    with sane payloads (non-zero repeat vectors, unit rotations), leaves at the bottom.  For the
    fuzz test: nestings nobody wrote down by hand."""
    rng = np.random.default_rng(seed)

    def f(lo, hi):
        return float(np.float32(rng.uniform(lo, hi)))

    def leaf():
        kind = rng.integers(0, 10)
        if kind < 3:
            return ("sfero", (f(1, 5),))
        if kind < 6:
            return ("kubo", (f(1, 5), f(1, 5), f(1, 5)))
        if kind < 9:
            return ("cilindro", (f(1, 4), f(1, 5)))
        n = np.array([f(-0.3, 0.3), f(-0.3, 0.3), 1.0], dtype=np.float32)
        n = n / np.sqrt(np.sum(n * n), dtype=np.float32)
        return ("movo", (0, 0, f(-9, -5)), ("ebeno", tuple(float(v) for v in n)))

    def moved(t):
        return ("movo", (f(-6, 6), f(-6, 6), f(-6, 6)), t)

    def nudged(t):
        return ("movo", (f(-2, 2), f(-2, 2), f(-2, 2)), t)

    def big():
        return ("sfero", (f(6, 10),)) if rng.random() < 0.5 else ("kubo", (f(5, 9), f(5, 9), f(5, 9)))

    def build(d):
        if d == 0 or (d < depth and rng.random() < 0.12):
            return leaf()
        pick = rng.integers(0, 16)
        if pick < 3:
            n = int(rng.integers(2, 5))
            return ("kunigajo_%d" % n, ()) + tuple(moved(build(d - 1)) for _ in range(n))
        if pick < 6:                              # a big solid first, so that the intersection is rarely empty
            n = int(rng.integers(2, 5))
            return ("komunajo_%d" % n, (), big()) + tuple(nudged(build(d - 1)) for _ in range(n - 1))
        if pick == 6:
            return ("komunajo_2", (), big(), ("komplemento", (), moved(build(d - 1))))
        if pick == 7:
            return ("glata_kunigajo_2", (f(0.3, 3),), build(d - 1), moved(build(d - 1)))
        if pick == 8:
            return ("glata_komunajo_2", (f(0.3, 3),), big(), nudged(build(d - 1)))
        if pick == 9:
            return moved(build(d - 1))
        if pick in (10, 11, 12):
            a = f(-3, 3)
            return ("rotacio" + "XYZ"[pick - 10], (np.float32(np.cos(a)), np.float32(np.sin(a))), build(d - 1))
        if pick == 13:
            return ("rotacioQ", quaternion((f(-1, 1), f(-1, 1), f(0.1, 1)), f(-3, 3)), build(d - 1))
        if pick == 14:
            axis = [(f(12, 30), 0, 0), (0, f(12, 30), f(-3, 3)), (f(12, 30), f(-3, 3), f(-3, 3))][rng.integers(0, 3)]
            return ("senfina_ripeto", axis, build(d - 1))
        return ("portanta_sfero", (f(-2, 2), f(-2, 2), f(-2, 2), f(6, 14)), build(d - 1))

    return ("kunigajo_3", (), moved(build(depth - 1)), moved(build(depth - 1)), moved(build(depth - 1)))


def random_carved(seed):
    """A seeded solid with spheres cut out of it, the shape of chair.sce: transforms over a komunajo of one or two
    big solids and the komplemento of one to four spheres, each under transforms of its own (offsets that are +0, -0
    or not, axis rotations), some reaching out through the solid's faces, some tangent to one another.  For the
    generated code's sphere_skip: rays that graze faces, start in hollows and cross from one into the next."""
    rng = np.random.default_rng(seed)

    def f(lo, hi):
        return float(np.float32(rng.uniform(lo, hi)))

    def off(lo, hi):
        k = rng.integers(0, 4)
        return 0.0 if k == 0 else -0.0 if k == 1 else f(lo, hi)

    def ball():
        t = ("sfero", (f(1.5, 6),))
        for _ in range(int(rng.integers(0, 3))):
            k = rng.integers(0, 5)
            if k < 3:
                t = ("movo", (off(-7, 7), off(-7, 7), off(-7, 7)), t)
            else:
                a = f(-3, 3)
                t = ("rotacio" + "XYZ"[rng.integers(0, 3)], (np.float32(np.cos(a)), np.float32(np.sin(a))), t)
        return ("movo", (off(-7, 7), off(-7, 7), off(-7, 7)), t)

    body = ("kubo", (f(4, 8), f(4, 8), f(4, 8))) if rng.random() < 0.7 else ("cilindro", (f(4, 7), f(4, 8)))
    n = int(rng.integers(1, 5))
    balls = [ball() for _ in range(n)]
    if n == 1:
        cut = ("komplemento", (), balls[0])
    elif n == 4 or rng.random() < 0.5:
        cut = ("komplemento", (), ("kunigajo_%d" % n, ()) + tuple(balls))
    else:       # komplementoj one by one: the chain merges them
        cut = None
    members = [body]
    if rng.random() < 0.4:
        members.append(("movo", (f(-1, 1), f(-1, 1), f(-1, 1)), ("sfero", (f(7, 10),))))
    if cut is not None:
        members.append(cut)
    else:
        members += [("komplemento", (), b) for b in balls]
    while len(members) > 4:
        members.pop(1)
    solid = ("komunajo_%d" % len(members), ()) + tuple(members) if len(members) > 1 else members[0]
    a = f(-3, 3)
    solid = ("rotacioQ", quaternion((f(-1, 1), f(-1, 1), f(0.1, 1)), a), solid) if rng.random() < 0.5 else solid
    solid = ("movo", (f(-3, 3), off(-3, 3), off(-3, 3)), solid)
    if rng.random() < 0.5:
        solid = ("senfina_ripeto", (0, f(30, 45), f(-5, 5)), solid)
    return solid


def random_lattice(seed):
    """A seeded scene that is mostly lattice: one to three nested senfina_ripeto (vectors along one axis, in a plane or
    general, components of either sign, periods from 20 to 120) over a small solid, plain or as the komplemento of a
    sphere larger than the cell (infinity.sce's shape).  For the generated code's cell_cache: rays that cross many cells,
    run along cell boundaries and - from the far cameras of the test - start hundreds of cells out."""
    rng = np.random.default_rng(seed)

    def f(lo, hi):
        return float(np.float32(rng.uniform(lo, hi)))

    def vec(period):
        k = rng.integers(0, 6)
        sgn = -1.0 if rng.random() < 0.3 else 1.0
        if k < 3:                                   # along one axis
            v = [0.0, 0.0, 0.0]
            v[k] = sgn * period
            return tuple(v)
        if k < 5:                                   # in a coordinate plane
            v = [sgn * period, f(-0.4, 0.4) * period, 0.0]
            rng.shuffle(v)
            return tuple(float(x) for x in v)
        return (sgn * period, f(-0.3, 0.3) * period, f(-0.3, 0.3) * period)

    period = f(20, 120)
    kind = rng.integers(0, 4)
    if kind == 0:
        solid = ("komplemento", (), ("sfero", (0.65 * period,)))
    elif kind == 1:
        solid = ("sfero", (f(0.15, 0.35) * period,))
    elif kind == 2:
        solid = ("kubo", (f(0.1, 0.3) * period, f(0.1, 0.3) * period, f(0.1, 0.3) * period))
    else:
        solid = ("kunigajo_2", (), ("sfero", (0.2 * period,)), ("movo", (0.25 * period, 0.0, 0.1 * period), ("kubo", (0.1 * period,) * 3)))
    t = solid
    for _ in range(int(rng.integers(1, 4))):
        t = ("senfina_ripeto", vec(period * f(0.8, 1.25)), t)
    if rng.random() < 0.4:
        t = ("movo", (f(-30, 30), 0.0, f(-30, 30)), t)
    return t


def tables():
    return {name: Tree(t).table() for name, t in SCENES.items()}


def points(count=192, seed=7, span=12.0):
    """Seeded query points plus hand-picked ones on symmetry planes and faces."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-span, span, size=(count, 3)).astype(np.float32)
    near = rng.uniform(-1.0, 1.0, size=(count // 4, 3)).astype(np.float32) * np.float32(6.0)
    special = np.array([
        [0, 0, 0], [5, 0, 0], [0, 5, 0], [0, 0, 5], [-5, 0, 0], [3, 4, 5], [3, 0, 0], [0, 4, 0], [0, 0, -5],
        [1, 1, 1], [-1, -1, -1], [2.5, 0, 0], [0, 0, 4], [3, 0, 4], [4.02, 0, 0], [5.02, 0, 0], [5.019, 0, 0],
        [0.5, 0.25, 0.125], [7, 1, 2], [3.5, 0.5, 1], [10.5, 1.5, 3], [1e-3, -1e-3, 1e-3], [100, -200, 300],
    ], dtype=np.float32)
    return np.concatenate([pts, near, special])


# Hand-derived known answers: values worked out on paper from the formulas in the reference's source text
# (Source/cuda_kernels.cu, the lines cited per case), with payloads and points chosen so that every
# intermediate is exact in fp32 (small dyadic numbers, Pythagorean triples).  They pin the restatement - and
# through it the device evaluators - for the node types neither stock scene uses, against something that is not
# the restatement itself.  (tree, point, DIST, NORM unnormalised); a string "sqrt:x" stands for float32(sqrt(x)).
_E124 = ("ebeno", (1, 2, 4))
_EX, _EY = ("ebeno", (1, 0, 0)), ("ebeno", (0, 1, 0))
_S1 = ("sfero", (1,))
_XP, _XM, _YP, _YM = (("movo", t, _S1) for t in ((6, 0, 0), (-6, 0, 0), (0, 6, 0), (0, -6, 0)))
ANALYTIC = [
    # sfero :72-79: |p| - r; the normal is p itself
    (S5, (3, 4, 12), 8.0, (3, 4, 12)),
    # kubo :81-105, b = (3,4,5).  Inside: the largest of q = |p| - b; axis picked by q.x>q.z ? (q.x>q.y ? X : Y) : ..., sign p>0 ? 1 : -1
    (K345, (0, 0, 0), -3.0, (-1, 0, 0)),
    (K345, (1, 3.5, -1), -0.5, (0, 1, 0)),
    (K345, (2, 0, -4.75), -0.25, (0, 0, -1)),
    # outside: |max(q,0)|; per axis q>0 ? sign : 0
    (K345, (6, 8, 5), 5.0, (1, 1, 0)),
    (K345, (-6, 0, 0), 3.0, (-1, 0, 0)),
    (K345, (-5, 10, 8), "sqrt:49", (-1, 1, 1)),            # q = (2,6,3): 4+36+9
    # cilindro :107-127, r = 3, h = 4: q = (hypot(p.x,p.y) - r, |p.z| - h)
    (C34, (3, 4, 0), 2.0, (3, 4, 0)),                       # q = (2,-4): hypot(2,0); q.x > q.y: radial normal
    (C34, (0, 0, 10), 6.0, (0, 0, 1)),                      # q = (-3,6)
    (C34, (0, 0, 0), -3.0, (0, 0, 0)),                      # inside, q = (-3,-4): the larger; q.x > q.y: (p.x,p.y,0)
    (C34, (0, 0, -1), -3.0, (0, 0, -1)),                    # q = (-3,-3): a tie goes to the cap
    (C34, (6, 8, 7), "sqrt:58", (6, 8, 0)),                 # q = (7,3)
    (C34, (0, 5, -6), "sqrt:8", (0, 0, -1)),                # q = (2,2): tie, cap; p.z > 0 ? 1 : -1
    # ebeno :129-136: dot(n,p), normal n
    (("ebeno", (0.5, 0.25, 0.75)), (2, 4, 8), 8.0, (0.5, 0.25, 0.75)),
    (_E124, (-2, 0, 0.25), -1.0, (1, 2, 4)),
    # komplemento :336-347, movo :406-423
    (("komplemento", (), S5), (3, 4, 12), -8.0, (-3, -4, -12)),
    (("movo", (1, 2, 3), _E124), (2, 4, 6), 17.0, (1, 2, 4)),
    # rotations by stored cos/sin (not required to be a rotation: c = .5, s = .25 keeps everything exact)
    # rotacioX :425-447: P.y = c y + s z, P.z = -s y + c z; normal back with (c,-s; s,c)
    (("rotacioX", (0.5, 0.25), _E124), (1, 2, 4), 11.0, (1, 0, 2.5)),
    (("rotacioX", (0, 1), ("kubo", (1, 2, 3))), (0, 0, 10), 8.0, (0, 0, 1)),
    # rotacioY :449-471: P.z = c z + s x, P.x = -s z + c x
    (("rotacioY", (0.5, 0.25), _E124), (1, 2, 4), 12.5, (1.5, 2, 1.75)),
    # rotacioZ :473-495: P.x = c x + s y, P.y = -s x + c y
    (("rotacioZ", (0.5, 0.25), _E124), (1, 2, 4), 18.5, (0, 1.25, 4)),
    # rotacioQ :497-571 with q_w = .5, q = (.5,.5,.5): Q = ((-.5,0,.5),(.5,-.5,0),(0,.5,-.5)), P = p + 2 Q p = (z,x,y)
    (("rotacioQ", (0.5, 0.5, 0.5, 0.5), ("kubo", (1, 2, 3))), (10, 0, 0), 8.0, (1, 0, 0)),
    (("rotacioQ", (0.5, 0.5, 0.5, 0.5), _E124), (1, 2, 3), 13.0, (2, 4, 1)),       # P = (3,1,2): 3+2+8; N + 2 Q^T N
    # senfina_ripeto :573-592, a = (4,0,0): N = floor(dot(a,p)/dot(a,a) + .5)
    (("senfina_ripeto", (4, 0, 0), _S1), (8, 3, 4), 4.0, (0, 3, 4)),
    (("senfina_ripeto", (4, 0, 0), _S1), (6, 0, 0), 1.0, (-2, 0, 0)),              # 1.5 + .5 = 2: the cell boundary goes up
    (("senfina_ripeto", (4, 0, 0), _S1), (-9, 0, 0), 0.0, (-1, 0, 0)),             # floor(-2.25 + .5) = -2
    (("senfina_ripeto", (4, 0, 0), _S1), (0, 0, 0), -1.0, (0, 0, 0)),              # numerators at and near zero: N = 0
    (("senfina_ripeto", (4, 0, 0), _S1), (1e-30, 0, 0), -1.0, (1e-30, 0, 0)),
    (("senfina_ripeto", (4, 0, 0), _S1), (-1e-42, 3, 4), 4.0, (-1e-42, 3, 4)),     # a denormal coordinate
    # portanta_sfero :51-70, t = (1,2,3), r = 6: outside the sphere its distance, within RAYS_MIN_DIST of it the child
    (("portanta_sfero", (1, 2, 3, 6), ("kubo", (2, 2, 2))), (1, 2, 13), 4.0, (0, 0, 1)),
    (("portanta_sfero", (1, 2, 3, 6), ("kubo", (2, 2, 2))), (1, 2, 8), 3.0, (0, 0, 1)),
    # kunigajo :138-235: min; the normal's child: _2 takes o1 unless d0 < d1, _3/_4 move on only on strict >
    (("kunigajo_2", (), _XP, _XM), (0, 0, 0), 5.0, (6, 0, 0)),
    (("kunigajo_2", (), _XP, _XM), (1, 0, 0), 4.0, (-5, 0, 0)),
    (("kunigajo_3", (), _XP, _XM, _YP), (0, 0, 0), 5.0, (-6, 0, 0)),
    (("kunigajo_4", (), _XP, _XM, _YP, _YM), (0, 0, 0), 5.0, (-6, 0, 0)),
    (("kunigajo_4", (), _XP, _XM, _YP, _YM), (0, -2, 0), 3.0, (0, 4, 0)),
    (("kunigajo_3", (), _XP, _XM, _YP), (0, 3, 0), 2.0, (0, -3, 0)),
    # komunajo :237-334: max; _2 takes o1 unless d0 > d1, _3/_4 move on only on strict <
    (("komunajo_2", (), _XP, _XM), (0, 0, 0), 5.0, (6, 0, 0)),
    (("komunajo_2", (), _XP, _XM), (-1, 0, 0), 6.0, (-7, 0, 0)),
    (("komunajo_3", (), _XP, _XM, _YP), (0, 0, 0), 5.0, (-6, 0, 0)),
    (("komunajo_4", (), _XP, _XM, _YP, _YM), (0, 0, 0), 5.0, (-6, 0, 0)),
    (("komunajo_4", (), _XP, _XM, _YP, _YM), (1, 0, 0), 6.0, (7, 0, 0)),
    (("komunajo_3", (), _XP, _XM, _YP), (0, -3, 0), 8.0, (0, -9, 0)),
    # glata_kunigajo_2 :349-375, k = 2 over the planes x and y: h = (1 - (d0-d1)/k)/2; h > 1: d0, h < 0: d1,
    # else mix(d0,d1,h) - k h (1-h) with mix(a,b,x) = a + x (b - a); normal: the same gates, else the mix of the unit normals
    (("glata_kunigajo_2", (2,), _EX, _EY), (1, 2, 0), 1.375, (0.25, 0.75, 0)),
    (("glata_kunigajo_2", (2,), _EX, _EY), (0, 8, 0), 0.0, (1, 0, 0)),
    (("glata_kunigajo_2", (2,), _EX, _EY), (8, 0, 0), 0.0, (0, 1, 0)),
    # glata_komunajo_2 :377-403: h = (1 + (d0-d1)/k)/2, + k h (1-h)
    (("glata_komunajo_2", (2,), _EX, _EY), (1, 2, 0), 1.625, (0.75, 0.25, 0)),
    (("glata_komunajo_2", (2,), _EX, _EY), (8, 0, 0), 8.0, (1, 0, 0)),
    (("glata_komunajo_2", (2,), _EX, _EY), (0, 8, 0), 8.0, (0, 1, 0)),
]


def analytic_cases():
    """[(table, point float32[3], dist float32, norm float32[3])]"""
    out = []
    for tree, p, d, n in ANALYTIC:
        if isinstance(d, str):
            d = np.sqrt(np.float32(float(d.split(":")[1])), dtype=np.float32)
        out.append((Tree(tree).table(), np.array(p, dtype=np.float32), np.float32(d), np.array(n, dtype=np.float32)))
    return out


# Hand-derived WHOLE-PIXEL known answers: the march and shading loop of Source/cuda_kernels.cu:619-717 followed
# step by step on paper, in scenes of planes (and one sphere on its axis) where every distance, every step and
# every running sum is exact in fp32 or, where it is not (the shadow ray's travelled distance, which starts at
# a = 2 * float(.02)), stays so far from every threshold it is compared with that its rounding cannot matter.
# The camera block has StartWVec = StartHVec = 0, so every pixel of the frame is the same ray (:742-760:
# pos = X*0 + Y*0, d = Depth*StartDir / |Depth*StartDir|, p = StartPos) and the whole frame must be the one
# pixel worked out here.  (table, StartPos, StartDir, LightSource, RGBA, hit, primary root evaluations, shadow
# root evaluations).  Shading constants used below: .9f - .1f rounds to 13421772 * 2^-24 (.79999995); with
# N_L = SHADOW = 1 the luminance is 255 * float(.1f + .79999995f) = 255 * .9f = 229.49999 -> rounds to 229.5
# -> LIGHT 229 -> (u8)(229*.5), (u8)(229*.7f), 229 = (114, 160, 229); with a zero product 255 * .1f = 25.5
# -> LIGHT 25 -> (12, 17, 25).
_FLOOR = ("ebeno", (0, 0, 1))                                       # DIST = z
_WALL_X4 = ("movo", (4, 0, 0), ("ebeno", (-1, 0, 0)))               # DIST = 4 - x
PIXELS = [
    # A. A miss that leaves by ray_dist >= RAYS_MAX_DIST (:712).  Unit sphere, camera on its axis at z = 8 looking away:
    #    p_k = 7*2^k + 1, cd_k = 7*2^k (sqrtf(z*z) == z), ray_dist after step k = 7*(2^(k+1) - 1) >= 1e4 first at k = 10:
    #    11 evaluations, the black pixel of :622.
    ("miss_by_distance", ("sfero", (1,)), (0, 0, 8), (0, 0, 1), (0, 0, 1), (0, 0, 0, 255), 0, 11, 0),
    # B. A floor lit from straight above.  z = 8 -> cd 8 -> z = 0: cd = 0 < .02, not < -0.f: NORM = (0,0,1), R_1 = 1,
    #    dot(n, d) = -1 < 0 (:640), N_L = 1.  Shadow ray from z = a = .04: cd = z doubles every step,
    #    128*cd/ray_dist = 128 every time (SHADOW stays 1), ray_dist = a*2^(J+1) >= 1e4 first at J = 17 (:689): 18 evaluations.
    ("lit_floor", _FLOOR, (0, 0, 8), (0, 0, -1), (0, 0, 1), (114, 160, 229, 255), 1, 2, 18),
    # C. An occluded hit.  Floor z >= 0 and a wall x >= 4, light (1,0,1) (N_L = 1).  Primary: cd = min(8, 4) = 4, then
    #    min(4, 4) = 4, then 0: kunigajo_2's normal takes child 0 because 0 < 4 (:153).  Shadow ray (a,0,a): cd = a*2^J
    #    for J = 0..5 (floor nearer), at J = 6 the wall is nearer: cd = 4 - 64a, exact, and the step lands ON the wall:
    #    x = z = ray_dist = 4 exactly.  J = 7: cd = 0, 8*0 < .02 (:662), the wall's normal (-1,0,0) against the VIEW
    #    direction (quirk q4, :673): dot = 0 < .02 -> SHADOW = 0.
    ("occluded", ("kunigajo_2", (), _FLOOR, _WALL_X4), (0, 0, 8), (0, 0, -1), (1, 0, 1), (12, 17, 25, 255), 1, 3, 8),
    # D. A start inside a solid: the slab -1 < z < 0, camera at z = -63/64 looking down, light pointing down.
    #    cd = max(z, -(z+1)) = -1/64 < -0.f (:630): the normal is -d = (0,0,1) although the nearest face's is (0,0,-1)
    #    (with NORM the sample would face away and the ray would march on).  dot = -1: hit, N_L = -1 -> 0.  The shadow ray
    #    starts below the slab (z = -(63/64 + a), rounded to 8593080 * 2^-23), cd = |z| - 1 = 204472 * 2^-23 doubles exactly,
    #    128*cd/ray_dist >= 77 throughout, ray_dist = a + cd_0*(2^(J+1) - 1) >= 1e4 first at J = 18.  LIGHT = 255*(.1f + 0) = 25.5.
    ("inside_start", ("komunajo_2", (), _FLOOR, ("komplemento", (), ("movo", (0, 0, -1), _FLOOR))),
     (0, 0, -0.984375), (0, 0, -1), (0, 0, -1), (12, 17, 25, 255), 1, 1, 19),
    # E. Near-surface samples that face away march on (quirk q1, :640 and :707-710).  Camera 1/64 above the floor moving
    #    along +x: cd = 1/64 < .02 at every step, normal (0,0,1), dot(n, d) = 0 is not < 0: 128 steps of 1/64.  The wall
    #    at x = 2 + 1/256 is farther than the floor until x = 2 (step 128): cd = 1/256, its normal (-1,0,0) faces the ray:
    #    hit at the 129th evaluation, N_L = dot((-1,0,0), (-1,0,1)) = 1.  Shadow ray up and away from the wall: the wall
    #    stays the nearer by 3145738 * 2^-28, cd doubles exactly from 368312 * 2^-23, SHADOW stays 1, ray_dist >= 1e4 at J = 17.
    ("grazing_then_wall", ("kunigajo_2", (), _FLOOR, ("movo", (2.00390625, 0, 0), ("ebeno", (-1, 0, 0)))),
     (0, 0, 0.015625), (1, 0, 0), (-1, 0, 1), (114, 160, 229, 255), 1, 129, 18),
    # F. The shadow march's cap of 300 steps (:660) inside a penumbra.  Wall x >= 0 seen head-on from x = 8 at height 1/4 over
    #    a floor: 32 primary steps of 1/4 (the floor is nearer), hit at x = 0 with the wall's normal, N_L = 1.  Shadow ray
    #    back along +x: cd = a, 2a, 4a, then the floor's 1/4 for ever: SHADOW = min(SHADOW, 32/ray_dist) falls as ray_dist
    #    = 8a + (J-3)/4 grows, never below .01 (:680), ray_dist never reaches 1e4: the loop ends at J = 300 with
    #    SHADOW = 32/(8a + 74) = .430570 (ray_dist is rounded once per binade it crosses: < 1e-5 in all),
    #    LIGHT = (u8)(255*(.1f + .79999995f*.430570)) = (u8)113.336 -> (56, 79, 113).  Without the cap: LIGHT 27.
    ("shadow_step_cap", ("kunigajo_2", (), ("ebeno", (1, 0, 0)), _FLOOR),
     (8, 0, 0.25), (-1, 0, 0), (1, 0, 0), (56, 79, 113, 255), 1, 33, 300),
    # G. The primary march's cap of 500 steps (:626): E's camera without the wall.  500 near-surface samples that face away,
    #    ray_dist = 500/64: the black pixel of a miss.
    ("primary_step_cap", _FLOOR, (0, 0, 0.015625), (1, 0, 0), (0, 0, 1), (0, 0, 0, 255), 0, 500, 0),
]


# cases whose hit surface faces away from the light (N_L <= 0): the reference marches the shadow ray all the same; the
# product's kernel shades such a pixel without that march (:694 multiplies its result by zero), so its trip count has no
# shadow evaluations in it
PIXELS_UNLIT = {"inside_start"}


def pixel_cases(info_dtype):
    """[(name, table, rays_info block, rgba uint8[4], hit, primary evaluations, shadow evaluations)]"""
    out = []
    for name, tree, pos, direction, light, rgba, hit, n_primary, n_shadow in PIXELS:
        info = np.zeros(1, dtype=info_dtype)
        info["Depth"] = 1
        for field, v in (("StartPos", pos), ("StartDir", direction), ("LightSource", light)):
            for axis, x in zip("xyz", v):
                info[field][axis] = x
        out.append((name, Tree(tree).table(), info, np.array(rgba, dtype=np.uint8), hit, n_primary, n_shadow))
    return out
