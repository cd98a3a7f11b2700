#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/ from the CPU restatement.

What pins what:
  * BASELINE.md section 2 (outputs of the reference's own march, run host-side
    during the survey) pins the restatement: frame sha256 at 640x480, hit counts
    at 640x480 and 1920x1080, evaluation counts.  Those constants live in
    tests/test_oracle.py and are NOT produced by this script.
  * This script freezes further outputs of the pinned restatement so that later
    edits to it (or a different compiler) cannot drift unnoticed, and so that GPU
    tests at sizes the oracle cannot finish in seconds have something to compare
    with: node-level known answers, 1080p / 4K frame digests, hit masks,
    operation counters (the algorithmic FLOPs of bench.py's roofline), and the
    120 camera blocks of the 8K sweep.

Run from the repository root:  python tests/golden/make_goldens.py [--big]
"""
import argparse
import hashlib
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import kat_scenes  # noqa: E402
from oracle import oracle as O  # noqa: E402
from raytracing_amd import api, scenes  # noqa: E402

HERE = os.path.join(ROOT, "tests", "golden")


def frame_record(nodes, w, h, info, threads):
    r = O.render(nodes, w, h, info, threads=threads, want_counters=True, want_evals=True)
    packed = np.packbits(r["hit"].astype(bool), axis=None)
    rec = {
        "width": w, "height": h,
        "hits": int(r["hit"].sum()),
        "rgba_sha256": hashlib.sha256(r["rgba"].tobytes()).hexdigest(),
        "ppm_sha256": hashlib.sha256(O.ppm_bytes(r["rgba"])).hexdigest(),
        "hit_sha256": hashlib.sha256(packed.tobytes()).hexdigest(),
        "mean_root_evals": float(r["evals"].mean()),
        "max_root_evals": int(r["evals"].max()),
        "counters": r["counters"],
        "flops_per_ray": r["counters"]["flops"] / float(w * h),
        "oracle_seconds": r["seconds"], "oracle_threads": threads,
    }
    return rec, packed


def windowed_record(nodes, w, h, info, threads, want):
    """The same frame by the windowed march (the product's ray windows restated in the oracle): its counters are
    the FLOPs the product's kernel executes (bench.py's roofline), its pixels must be the plain march's."""
    b = api.scene_bound(nodes)
    window = O.bound_record(b.kind, list(b.c), list(b.u), b.r)
    r = O.render(nodes, w, h, info, threads=threads, want_counters=True, want_evals=True, window=window)
    assert hashlib.sha256(r["rgba"].tobytes()).hexdigest() == want["rgba_sha256"], "the windowed march changed a pixel"
    assert int(r["hit"].sum()) == want["hits"]
    return {"bound": b.as_dict(), "counters": r["counters"], "flops_per_ray": r["counters"]["flops"] / float(w * h),
            "mean_root_evals": float(r["evals"].mean()), "rays_not_marched": int((r["evals"] == 0).sum()),
            "oracle_seconds": r["seconds"], "oracle_threads": threads}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also the 4K csg1024 frame (minutes of CPU)")
    ap.add_argument("--windowed-only", action="store_true",
                    help="only add the windowed march's counters to the frames already in frames.json (every one of them)")
    ap.add_argument("--kat-only", action="store_true", help="only the node-level known answers (a scene added to kat_scenes.py)")
    args = ap.parse_args()
    threads = os.cpu_count() or 4
    O.build()

    if args.windowed_only:
        path = os.path.join(HERE, "frames.json")
        with open(path) as f:
            frames = json.load(f)
        info = O.default_info(1, 0)
        for key, rec in sorted(frames.items(), key=lambda kv: kv[1]["counters"]["flops"]):
            name = key.rsplit("_", 1)[0]
            nodes = scenes.csg1024() if name == "csg1024" else O.load_sce(os.path.join(HERE, "scenes", name + ".sce"))
            rec["windowed"] = windowed_record(nodes, rec["width"], rec["height"], info, threads, rec)
            print(key, "%.0f -> %.0f flop/ray" % (rec["flops_per_ray"], rec["windowed"]["flops_per_ray"]),
                  "%.1fs" % rec["windowed"]["oracle_seconds"], flush=True)
            with open(path, "w") as f:
                json.dump(frames, f, indent=1, sort_keys=True)
        return

    # ---- node-level known answers
    pts = kat_scenes.points()
    kat = {"points": pts}
    for name, table in kat_scenes.tables().items():
        kat[name + ".dist"] = O.dist_many(table, 0, pts).view(np.uint32)
        kat[name + ".norm"] = O.norm_many(table, 0, pts).view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "node_kat.npz"), **kat)
    if args.kat_only:
        return

    # ---- kernel_SetRays known answers
    rays = {}
    for (w, h, phi) in [(640, 480, 0), (1920, 1080, 37), (3840, 2160, 3), (7680, 4320, 119)]:
        info = O.default_info(1, phi)
        rng = np.random.default_rng(w + phi)
        xy = np.stack([rng.integers(0, w, 64), rng.integers(0, h, 64)], axis=1).astype(np.int32)
        xy[:5] = [[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1], [w // 2, h // 2]]
        key = "%dx%d_phi%d" % (w, h, phi)
        rays[key + ".xy"] = xy
        rays[key + ".info"] = info.view(np.uint8)
        rays[key + ".rays"] = np.stack([O.set_ray(w, h, info, int(x), int(y)) for x, y in xy]).view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "set_rays_kat.npz"), **rays)

    # ---- the 120 camera blocks of the 8K sweep
    sweep = np.concatenate([O.default_info(k + 1, k) for k in range(120)])
    sweep.tofile(os.path.join(HERE, "sweep_rays_info.bin"))

    # ---- frames
    stock = {n: O.load_sce(os.path.join(HERE, "scenes", n + ".sce")) for n in ("chair", "infinity")}
    csg = scenes.csg1024()
    with open(os.path.join(HERE, "scenes", "csg1024.sce"), "wb") as f:
        f.write(O.encode_sce(csg))
    info = O.default_info(1, 0)
    frames = {}
    jobs = [("chair", 640, 480), ("infinity", 640, 480), ("chair", 1920, 1080), ("infinity", 1920, 1080),
            ("chair", 3840, 2160), ("infinity", 3840, 2160), ("csg1024", 256, 144), ("csg1024", 960, 540)]
    if args.big:
        jobs.append(("csg1024", 3840, 2160))
    path = os.path.join(HERE, "frames.json")
    if os.path.exists(path):
        with open(path) as f:
            frames = json.load(f)
    for name, w, h in jobs:
        nodes = csg if name == "csg1024" else stock[name]
        rec, packed = frame_record(nodes, w, h, info, threads)
        rec["windowed"] = windowed_record(nodes, w, h, info, threads, rec)
        key = "%s_%dx%d" % (name, w, h)
        frames[key] = rec
        with open(os.path.join(HERE, "hit_%s.z" % key), "wb") as f:
            f.write(zlib.compress(packed.tobytes(), 9))
        print(key, rec["hits"], "%.1f evals/px" % rec["mean_root_evals"], "%.0f flop/ray" % rec["flops_per_ray"],
              "%.1fs" % rec["oracle_seconds"], flush=True)
    with open(path, "w") as f:
        json.dump(frames, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
