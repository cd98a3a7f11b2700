"""CPU suite: host logic of the product library (no GPU calls): C-ABI exports,
bazo / .sce codecs in all three layouts, the scene flattener, error codes."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

import kat_scenes
from oracle import oracle as O
from raytracing_amd import api, scenes, tiling

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Every function include/raymarching_amd.h declares is exported by the .so."""
    with open(os.path.join(ROOT, "include", "raymarching_amd.h")) as f:
        text = f.read()
    import re
    declared = sorted(set(re.findall(r"\b(rm_[a-z_0-9]+)\s*\(", text)))
    assert declared, "no declarations parsed"
    lib = api.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(api.EXPORTS) == declared


def test_product_does_not_reach_the_oracle():
    """The product tree never names the oracle, and the library does not link it."""
    import re
    uses = re.compile(r"import\s+oracle|from\s+oracle|from\s+\.+oracle|rm_oracle|rmo_|#\s*include[^\n]*oracle|-lrm_oracle")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracing_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                with open(os.path.join(dirpath, fn), errors="ignore") as f:
                    assert not uses.search(f.read()), os.path.join(dirpath, fn)
    needed = subprocess.run(["readelf", "-d", api.LIB_PATH], capture_output=True, text=True).stdout
    assert "rm_oracle" not in needed


def test_decode_stock_scenes_matches_oracle(stock_scenes, golden_dir):
    for name, want in stock_scenes.items():
        got = api.load_sce(os.path.join(golden_dir, "scenes", name + ".sce"))
        assert got.tobytes() == want.tobytes()


def _record_llp64(node):
    """a bazo record as the reference's current header lays it out with 4-byte counters"""
    na = {0x100: 2, 0x101: 3, 0x102: 4, 0x103: 2, 0x104: 3, 0x105: 4, 0x107: 2, 0x108: 2}.get(int(node["type"]), None)
    if na is None:
        na = 0 if int(node["type"]) in (2, 3, 4, 0x80) else 1
    body = b"".join(struct.pack("<i", int(v)) for v in node["o"][:na]) + node["f"].tobytes()
    body = body[:24].ljust(24, b"\xcc")
    return struct.pack("<I", int(node["type"])) + body + b"\x00" * 4 + b"\xee" * 16


def _record_lp64(node):
    na = {0x100: 2, 0x101: 3, 0x103: 2, 0x104: 3, 0x107: 2, 0x108: 2}.get(int(node["type"]), None)
    if na is None:
        na = 0 if int(node["type"]) in (2, 3, 4, 0x80) else 1
    body = b"".join(struct.pack("<q", int(v)) for v in node["o"][:na]) + node["f"].tobytes()
    body = body[:24].ljust(24, b"\xcc")
    return struct.pack("<I", int(node["type"])) + body + b"\x00" * 4 + b"\xee" * 16


def test_decode_all_layouts():
    """LP64 / LLP64 in-memory records decode to the same nodes as the .sce form; the
    garbage in unused payload bytes and in dist/norm is ignored."""
    for name, table in kat_scenes.tables().items():
        sce = O.encode_sce(table)
        ref = api.decode_bazo(sce, api.BAZO_SCE)
        llp = api.decode_bazo(b"".join(_record_llp64(n) for n in table), api.BAZO_LLP64)
        assert llp.tobytes() == ref.tobytes(), name
        if not any(int(t) in (0x102, 0x105) for t in table["type"]):
            lp = api.decode_bazo(b"".join(_record_lp64(n) for n in table), api.BAZO_LP64)
            assert lp.tobytes() == ref.tobytes(), name
        assert O.decode_bazo(sce, O.BAZO_SCE).tobytes() == ref.tobytes()


def test_lp64_rejects_four_way_combinators():
    """kunigajo_4 needs 32 payload bytes with 8-byte counters: not representable (Include/objects_list.h:17)."""
    rec = struct.pack("<I", 0x102) + b"\x00" * 44
    with pytest.raises(api.RaymarchError) as e:
        api.decode_bazo(rec, api.BAZO_LP64)
    assert e.value.status == -3


def test_sce_save_load_round_trip(tmp_path):
    table = scenes.csg1024()
    path = str(tmp_path / "csg.sce")
    api.save_sce(path, table)
    assert os.path.getsize(path) == 1024 * 48
    assert api.load_sce(path).tobytes() == table.tobytes()
    with open(path, "rb") as f:
        assert f.read() == O.encode_sce(table)
    with open(path, "ab") as f:
        f.write(b"\x00" * 7)                               # not a whole number of records
    with pytest.raises(api.RaymarchError) as e:
        api.load_sce(path)
    assert e.value.status == -4
    with pytest.raises(api.RaymarchError):
        api.load_sce(str(tmp_path / "missing.sce"))


def test_csg1024_matches_committed_scene(golden_dir):
    """config 4's generator is deterministic: it reproduces the committed .sce."""
    table = scenes.csg1024()
    assert len(table) == 1024 and int(table[1023]["type"]) == 0 and int(table[0]["type"]) == 0x200
    kinds = [int(t) for t in table["type"]]
    assert kinds.count(0x100) == 340 and kinds.count(0x200) == 342
    assert kinds.count(0x2) + kinds.count(0x3) + kinds.count(0x4) == 341
    with open(os.path.join(golden_dir, "scenes", "csg1024.sce"), "rb") as f:
        assert f.read() == O.encode_sce(table)


def test_flatten_stock_scenes(stock_scenes):
    st, listing = api.flatten_scene(stock_scenes["chair"])
    ops = [line.split()[1] for line in listing.strip().splitlines()]
    assert ops == ["RIPETO", "MOVO", "ROTQ_XY", "ROTQ_Z", "KUBO", "PUSHA_R", "SFERO", "SFERO", "NEG", "POPC_R", "END"]
    assert st["reachable"] == 11 and st["value_slots"] == 0 and st["lds_bytes"] == 11 * 48
    st, listing = api.flatten_scene(stock_scenes["infinity"])
    assert [line.split()[1] for line in listing.strip().splitlines()] == ["RIPETO", "RIPETO", "RIPETO", "SFERO", "NEG", "END"]
    # main.cpp's own list: komunajo_3 over a box and two complemented spheres -> one max chain
    st, listing = api.flatten_scene(scenes.main_cpp_scene())
    assert [line.split()[1] for line in listing.strip().splitlines()] == ["RIPETO", "MOVO", "ROTQ_XY", "ROTQ_Z", "KUBO", "SFERO", "SFERO", "END"]
    assert listing.count("neg comb=max") == 2


def test_flatten_csg1024_is_one_running_min():
    st, listing = api.flatten_scene(scenes.csg1024())
    ops = [line.split()[1] for line in listing.strip().splitlines()]
    assert ops[0] == "MOVO" and ops[-1] == "END"
    # one running min over all 341 leaves, regrouped by position: a bounding-sphere test (CULL)
    # in front of every group of more than two, no value ever pushed
    assert set(ops[1:-1]) <= {"SFERO", "KUBO", "CILINDRO", "CULL"}
    assert sum(op != "CULL" for op in ops[1:-1]) == 341
    assert st["value_slots"] == 0 and st["reachable"] == 1023 and st["lds_bytes"] == 1024 * 48
    # every CULL skips a range that lies inside the program and inside its parent's range
    rows = [line.split() for line in listing.strip().splitlines()]
    import struct
    ends = []
    for pc, f in enumerate(rows):
        while ends and pc >= ends[-1]:
            ends.pop()
        if f[1] == "CULL":
            skip = struct.unpack("<I", struct.pack("<f", float(f[-3])))[0]
            end = pc + 1 + skip
            assert skip >= 3 and end <= len(rows) - 1 and (not ends or end <= ends[-1])
            ends.append(end)


def test_flatten_culls_a_carved_out_union():
    """komunajo(A, komplemento(union of 3+)): the union sits behind a bounding-sphere test on -A
    (it is at most r - |p - c|, so far from it the running max cannot rise)."""
    st, listing = api.flatten_scene(kat_scenes.tables()["carved"])
    ops = [line.split()[1] for line in listing.strip().splitlines()]
    assert ops[:3] == ["KUBO", "CULL", "PUSHA_R"] and ops[6:] == ["NEG", "POPC_R", "END"]
    assert sorted(ops[3:6]) == ["CILINDRO", "SFERO", "SFERO"]          # the union's members, regrouped by position
    assert "CULL         src=F dst=P neg" in listing


def test_flatten_stack_discipline():
    """Every push has its pop, slots never overlap, frames restore before siblings."""
    for name, table in kat_scenes.tables().items():
        st, listing = api.flatten_scene(table)
        live, b_live = {}, False
        for line in listing.strip().splitlines():
            f = line.split()
            op = f[1]
            slot = int([x for x in f if x.startswith("slot=")][0][5:])
            if op == "PUSHA_R":
                assert not b_live, name
                b_live = True
            elif op in ("POPC_R", "GLATA_R"):
                assert b_live, name
                b_live = False
            elif op in ("PUSHA", "PORTANTA"):
                assert slot not in live, name
                live[slot] = 1
            elif op in ("POPC", "GLATA", "PORTANTA_END"):
                assert live.pop(slot) == 1, name
            elif op == "PUSHF":
                assert all(s not in live for s in (slot, slot + 1, slot + 2)), name
                live[slot] = live[slot + 1] = live[slot + 2] = 3
            elif op == "POPF":
                assert live.pop(slot) == 3 and live.pop(slot + 1) == 3 and live.pop(slot + 2) == 3, name
        assert not live and not b_live, name
        assert st["value_slots"] <= 16


def test_flatten_rejects_bad_tables(stock_scenes):
    bad = stock_scenes["chair"].copy()
    bad[8]["type"] = 0
    with pytest.raises(api.RaymarchError) as e:
        api.flatten_scene(bad)
    assert e.value.status == -2
    bad = stock_scenes["chair"].copy()
    bad[6]["o"][1] = 40
    with pytest.raises(api.RaymarchError) as e:
        api.flatten_scene(bad)
    assert e.value.status == -2
    bad = stock_scenes["chair"].copy()
    bad[0]["o"][0] = -1
    with pytest.raises(api.RaymarchError) as e:
        api.flatten_scene(bad)
    assert e.value.status == -2
    # a chain deeper than the walker's private stack
    deep = scenes.Tree(_chain(80)).table()
    with pytest.raises(api.RaymarchError) as e:
        api.flatten_scene(deep)
    assert e.value.status == -5
    too_many = np.zeros(4096, dtype=api.NODE_DTYPE)
    with pytest.raises(api.RaymarchError) as e:
        api.flatten_scene(too_many)
    assert e.value.status == -5


def _chain(depth):
    t = ("sfero", (1,))
    for k in range(depth):
        t = ("rotacioX", (1.0, 0.0), t)
    return t


def test_unreachable_nodes_are_ignored(stock_scenes):
    """Records that nothing links to (the trailing nenio pad of config 4, odd-count over-read) do not matter."""
    chair = stock_scenes["chair"]
    padded = np.concatenate([chair, np.zeros(3, dtype=api.NODE_DTYPE)])
    padded[12]["type"] = 0xDEAD
    st, listing = api.flatten_scene(padded)
    assert st["reachable"] == 11 and listing == api.flatten_scene(chair)[1]


def test_tiling_partition_covers_every_row_once():
    for (h, band, n) in [(2160, 16, 8), (4320, 16, 8), (250, 32, 3), (15, 16, 2), (1080, 16, 4), (480, 16, 1)]:
        seen = np.zeros(h, dtype=int)
        for r in range(n):
            rows = tiling.band_of_rank(h, band, r, n)
            assert sum(y1 - y0 for y0, y1 in rows) == tiling.local_rows(h, band, r, n)
            assert tiling.local_rows(h, band, r, n) <= tiling.bands_per_rank(h, band, n) * band
            for y0, y1 in rows:
                seen[y0:y1] += 1
        assert (seen == 1).all()
        assert tiling.padded_rows(h, band, n) >= h and tiling.padded_rows(h, band, n) % (band * n) == 0


def test_weighted_deals_cover_every_row_once():
    """tiling.deal / lighter_rank0: bands dealt by weight (rm_set_band_pattern's owner table) still give every
    row to exactly one rank, in the proportions asked for, each rank's bands spread over the run."""
    assert tiling.deal([1, 1, 1]) == [0, 1, 2]
    assert tiling.lighter_rank0(8, 1.0) == [1] * 8 and tiling.lighter_rank0(1, 0.3) == [1]
    assert tiling.lighter_rank0(8, 0.5) == [1] + [2] * 7 and tiling.lighter_rank0(8, 0.0)[0] == 0
    w = tiling.lighter_rank0(8, 0.58)
    assert abs(w[0] / w[1] - 0.58) < 0.03 and sum(w) <= tiling.MAX_PERIOD and max(w) <= tiling.MAX_OWN
    for weights in ([1, 2, 2], [3, 5, 5, 5, 5, 5, 5, 5], [0, 1, 1, 1], [9, 10], [2, 16]):
        owner = tiling.deal(weights)
        assert [owner.count(r) for r in range(len(weights))] == weights
        for r, wt in enumerate(weights):                       # no rank's bands bunch up: gaps within a factor of two
            at = [k for k, o in enumerate(owner) if o == r]
            gaps = [b - a for a, b in zip(at, at[1:] + [at[0] + len(owner)])] if at else [1]
            assert max(gaps) <= 2 * min(gaps) + 1, (weights, r, owner)
        for h, band in [(2160, 16), (250, 32), (4320, 16)]:
            seen = np.zeros(h, dtype=int)
            for r in range(len(weights)):
                for y0, y1 in tiling.band_of_rank(h, band, r, len(weights), owner):
                    seen[y0:y1] += 1
            assert (seen == 1).all()
    for bad in ([], [0, 0], [17, 1], [8] * 9):
        with pytest.raises(ValueError):
            tiling.deal(bad)


def test_band_lookup_magic_is_exact():
    """fill_params hands the kernel ceil(2^32 / own) so that a tile finds its run of bands with a high multiply
    (device_render.h): exact for every band number an image can have (fewer than 2^27 bands of 8 rows in 2^30
    pixels; shown for band < 2^28 since band * (own * magic - 2^32) < 2^32)."""
    for own in range(2, 17):
        magic = ((1 << 32) + own - 1) // own
        assert (own * magic - (1 << 32)) * ((1 << 28) - 1) < (1 << 32)
        edge = [own * k + d for k in (1, 2, 3, 1000, (1 << 27) // own, ((1 << 28) - 1) // own) for d in (-1, 0, 1)]
        for band in list(range(4096)) + [b for b in edge if 0 <= b < (1 << 28)]:
            assert (band * magic) >> 32 == band // own, (own, band)


def test_cpp_front_end_is_built():
    assert os.path.exists(os.path.join(ROOT, "raytracing_amd", "rm_render")), "run __graft_entry__.build()"


def test_cpp_front_end_checks_its_device_list():
    """--gpus / --devices are checked before anything touches a GPU."""
    import subprocess
    exe = os.path.join(ROOT, "raytracing_amd", "rm_render")
    for args, text in [(["--gpus", "2", "--devices", "0"], "--devices names 1"), (["--devices", "0,x"], "not a device number"),
                       (["--gpus", "0"], "at least one")]:
        run = subprocess.run([exe, "--width", "64", "--height", "48"] + args, capture_output=True, text=True, timeout=60)
        assert run.returncode == 1 and text in run.stderr, (args, run.stderr)


@pytest.mark.parametrize("name", ["chair", "infinity", "carved", "deep_values", "portanta_nested", "csg1024"])
def test_generated_code_compiles_for_gfx950_without_a_device(name, stock_scenes):
    """rm_jit_check: the per-scene HIP source every generator writes (fused distance/normal pair, the
    pair with a culled distance program, the grouped union of a large scene) goes through hiprtc
    for gfx950 here, on a machine with no GPU."""
    table = stock_scenes[name] if name in stock_scenes else (scenes.csg1024() if name == "csg1024" else kat_scenes.tables()[name])
    rc, log = api.jit_check(table, want_source=False, assume_exact_divisors=True)
    assert rc == 0, log[-2000:]
    rc, src = api.jit_check(table, want_source=True, assume_exact_divisors=True, cap=1 << 22)
    assert rc == 0
    if name == "csg1024":         # one switch of compiled groups inside a loop, for distance and for the normal's pass
        assert "scene_pick" in src and src.count("switch (g)") == 2 and "unroll(disable)" in src
    elif "if (!cull_all(" in src:     # a distance program with bounding-sphere tests, normals from n0
        assert name in ("carved", "deep_values") and "return n0(p)" in src and "struct SceneKeep" not in src
    else:                             # the fused pair over kept temporaries
        assert name != "carved" and "struct SceneKeep" in src and "scene_norm_keep" in src


class _FakeContext:
    """Records the C-ABI calls bench.py's render loop makes on a context."""
    def __init__(self, log, name):
        self.log, self.name = log, name

    def set_batch(self, *a):
        self.log.append((self.name, "set_batch", a[0]))

    def set_info_batch(self, infos):
        self.log.append((self.name, "set_info_batch", len(infos)))

    def process_async(self):
        self.log.append((self.name, "process_async", None))


@pytest.mark.parametrize("steps,batch,contexts,want", [(20, 8, 1, [8, 8, 4]), (13, 4, 1, [4, 4, 4, 1]), (16, 8, 1, [8, 8]),
                                                       (21, 8, 3, [8, 8, 5]), (3, 32, 2, [3])])
def test_a_launch_that_is_not_full_sets_no_batch(steps, batch, contexts, want):
    """The driver times `--steps 20` with 8 frames per launch: launches of 8, 8 and 4 frames.  The last
    one must take its frame count from rm_set_info_batch alone - rm_set_batch inside the timed region
    waited for the stream, reallocated the scheduler and dropped the tile order."""
    from raytracing_amd.batching import FrameBatcher
    log = []
    ctxs = [_FakeContext(log, j) for j in range(contexts)]
    seen = []
    run = FrameBatcher(ctxs, batch, before=lambda j: seen.append(("before", j)), after=lambda j, n: seen.append(("after", j, n)))
    info = np.zeros(1, dtype=api.INFO_DTYPE)
    for k in range(steps):
        run.step(info)
    run.flush()
    run.flush()                                         # nothing pending: no launch
    assert [n for _, call, n in log if call == "set_info_batch"] == want
    assert [call for _, call, _ in log].count("process_async") == len(want) == run.launches
    assert not [c for c in log if c[1] == "set_batch"]
    assert [name for name, call, _ in log if call == "process_async"] == [k % contexts for k in range(len(want))]
    assert run.last == ((len(want) - 1) % contexts, want[-1] - 1)
    assert seen == [x for k, n in enumerate(want) for x in (("before", k % contexts), ("after", k % contexts, n))]
    # the same run planned: as few launches, of nearly equal size, and still nothing but cameras and launches
    del log[:]
    run = FrameBatcher(ctxs, batch)
    run.plan(steps)
    for k in range(steps):
        run.step(info)
    run.flush()
    sizes = [n for _, call, n in log if call == "set_info_batch"]
    assert len(sizes) == len(want) and sum(sizes) == steps and max(sizes) - min(sizes) <= 1 and max(sizes) <= batch
    assert not [c for c in log if c[1] == "set_batch"]


def test_timed_blocks_follow_each_other_without_a_wait():
    """bench.py's measured part (batching.run_blocks): one untimed block and the timed ones are nothing but cameras,
    launches and event marks - the fake context would record a synchronize, a set_batch or anything else - and
    every block goes out in the same launches.  bench.py itself collects garbage and synchronises before it,
    never inside it."""
    from raytracing_amd.batching import FrameBatcher, run_blocks
    log = []

    class Ctx(_FakeContext):
        def __getattr__(self, name):                    # any other call (synchronize, set_batch, read_pixels...) is recorded
            return lambda *a: self.log.append((self.name, name, None))

    ctxs = [Ctx(log, 0)]
    run = FrameBatcher(ctxs, 16, after=lambda j, n: log.append((j, "launch_mark", n)))
    info = np.zeros(1, dtype=api.INFO_DTYPE)
    marks, nxt = run_blocks(run, lambda k: info, 20, 7, lambda: log.append(("all", "mark", None)) or len(log), first=100)
    assert nxt == 100 + 8 * 20 and len(marks) == 8 and run.launches == 16
    calls = [c for _, c, _ in log]
    assert set(calls) == {"set_info_batch", "process_async", "launch_mark", "mark"}
    per_block = " ".join(calls).split(" mark")[:-1]
    assert len(per_block) == 8 and len({b.strip() for b in per_block}) == 1
    assert [n for _, c, n in log if c == "set_info_batch"] == [10, 10] * 8
    with open(os.path.join(ROOT, "bench.py")) as f:
        text = f.read()
    a, b = text.index("    start = mark()"), text.index("    drain()", text.index("    start = mark()"))
    timed = text[a:b]
    for banned in ("synchronize", "gc.", "barrier", "drain", ".cpu()", "torch.zeros", "torch.empty"):
        assert banned not in timed, banned
    assert text.index("gc.collect()") < a and text.index("gc.disable()") < a


def test_bench_sets_the_batch_once_before_timing():
    """bench.py itself: rm_set_batch is called ahead of the warm-up only (once per layout: whole frames, or the
    bands of a deal), and the oracle is imported only for the CPU baseline (the camera block comes from the
    committed fixture)."""
    with open(os.path.join(ROOT, "bench.py")) as f:
        text = f.read()
    assert text.count(".set_batch(") == 2 and text.rindex(".set_batch(") < text.index("# Warm-up")
    assert text.count("from oracle import") == 1 and text.index("from oracle import") < text.index("def main()")
    blocks = np.fromfile(os.path.join(ROOT, "tests", "golden", "sweep_rays_info.bin"), dtype=api.INFO_DTYPE)
    assert blocks[0:1].tobytes() == O.default_info(1, 0).tobytes()
