"""Scene construction on the host: the counterpart of the reference's
`primitives::TYPE::create_from({...})` builders (Include/defines.h:17-27,
Include/objects_list.h:8-30) and of the hard-coded list in Source/main.cpp:91-123,
producing canonical nodes (include/rm_types.h `rm_node`).

Also the deterministic generator of the 1024-record stress scene (SURVEY.md 8d,
config 4).
"""
import numpy as np

from .api import NODE_DTYPE

TYPE = {
    "nenio": 0x0000, "portanta_sfero": 0x0001, "sfero": 0x0002, "kubo": 0x0003, "cilindro": 0x0004,
    "ebeno": 0x0080,
    "kunigajo_2": 0x0100, "kunigajo_3": 0x0101, "kunigajo_4": 0x0102,
    "komunajo_2": 0x0103, "komunajo_3": 0x0104, "komunajo_4": 0x0105,
    "komplemento": 0x0106, "glata_kunigajo_2": 0x0107, "glata_komunajo_2": 0x0108,
    "movo": 0x0200, "rotacioX": 0x0201, "rotacioY": 0x0202, "rotacioZ": 0x0203,
    "rotacioQ": 0x0204, "senfina_ripeto": 0x0205,
}
NAME = {v: k for k, v in TYPE.items()}


def node(kind, o=(), f=()):
    """One canonical node; `o` are child links relative to the node."""
    n = np.zeros((), dtype=NODE_DTYPE)
    n["type"] = TYPE[kind]
    for i, v in enumerate(o):
        n["o"][i] = v
    for i, v in enumerate(f):
        n["f"][i] = np.float32(v)
    return n


class Tree:
    """Nested description -> node table in DFS pre-order with forward links.

    A tree is a tuple (kind, params, child, child, ...), e.g.
        ("movo", (1, 2, 3), ("sfero", (5,)))
    """

    def __init__(self, root):
        self.nodes = []
        self._emit(root)

    def _emit(self, t):
        kind, params, kids = t[0], t[1], t[2:]
        at = len(self.nodes)
        self.nodes.append(None)
        links = []
        for k in kids:
            links.append(len(self.nodes) - at)
            self._emit(k)
        self.nodes[at] = node(kind, links, params)
        return at

    def table(self, pad_to=None):
        out = np.array(self.nodes, dtype=NODE_DTYPE)
        if pad_to is not None and pad_to > len(out):
            out = np.concatenate([out, np.zeros(pad_to - len(out), dtype=NODE_DTYPE)])
        return out


def quaternion(axis, angle):
    """rotacioQ payload as Source/main.cpp:86-87,99 builds it (float32 steps)."""
    d = np.asarray(axis, dtype=np.float32)
    a = np.float32(angle)
    w = np.cos(a / np.float32(2), dtype=np.float32)
    r = np.float32(np.sin(a / np.float32(2), dtype=np.float32) /
                   np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], dtype=np.float32))
    return (w, r * d[0], r * d[1], r * d[2])


def main_cpp_scene():
    """The scene hard-coded in Source/main.cpp:91-123 (komunajo_3 + two komplemento)."""
    q = quaternion((1.0, 30.0, 1.0), -1.8)
    nodes = [
        node("senfina_ripeto", (1,), (0, 500, 100)),
        node("movo", (1,), (200, 0, 0)),
        node("rotacioQ", (1,), q),
        node("komunajo_3", (1, 2, 5)),
        node("kubo", (), (50, 50, 50)),
        node("komplemento", (1,)),
        node("movo", (1,), (0, 0, -50)),
        node("sfero", (), (60,)),
        node("komplemento", (1,)),
        node("movo", (1,), (0, 0, 50)),
        node("sfero", (), (40,)),
    ]
    return np.array(nodes, dtype=NODE_DTYPE)


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def unit(self):
        """float32 in [0,1) from the top 24 bits"""
        return np.float32((self.next() >> 40) / 16777216.0)

    def between(self, lo, hi):
        return np.float32(np.float32(lo) + self.unit() * np.float32(hi - lo))


def csg_union(seed, leaves, pad_to=None):
    """SURVEY.md 8d config 4 (csg1024 = 341 leaves padded to exactly 1024 records), any size:

    record 0 = movo(t=(200,0,0)); a balanced binary kunigajo_2 tree (340 inner
    nodes) over 341 leaves, each leaf = movo(t in [-60,60]^3) -> one of
    sfero(r in [4,12]) / kubo(b in [3,10]^3) / cilindro(r in [3,8], h in [4,12]);
    DFS pre-order, forward links; one trailing type_nenio record.
    Per leaf the generator draws: kind, t.x, t.y, t.z, then the shape parameters.
    """
    rng = SplitMix64(seed)

    def leaf():
        kind = rng.next() % 3
        t = (rng.between(-60, 60), rng.between(-60, 60), rng.between(-60, 60))
        if kind == 0:
            prim = ("sfero", (rng.between(4, 12),))
        elif kind == 1:
            prim = ("kubo", (rng.between(3, 10), rng.between(3, 10), rng.between(3, 10)))
        else:
            prim = ("cilindro", (rng.between(3, 8), rng.between(4, 12)))
        return ("movo", t, prim)

    def union(count):
        if count == 1:
            return leaf()
        left = (count + 1) // 2
        a = union(left)
        b = union(count - left)
        return ("kunigajo_2", (), a, b)

    table = Tree(("movo", (200, 0, 0), union(leaves))).table(pad_to=pad_to)
    return table


def csg1024(seed=0x5CE1024):
    table = csg_union(seed, 341, 1024)
    assert len(table) == 1024
    return table
