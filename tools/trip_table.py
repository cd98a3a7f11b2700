#!/usr/bin/env python3
"""trip_table.py SCENE [out.txt]: the march loop of a scene's generated kernel (8-wave build, compiled offline
for gfx950), instruction by instruction and by class, against the algorithmic FLOPs of one trip.  No GPU needed.
The loop is found as the innermost loop of the assembly that holds the lattice's v_floor_f32 (chair.sce,
infinity.sce) or, failing that, the first v_rsq_f32; blocks the compiler placed behind the loop (the IEEE
fall-backs of the short sqrt and quotient) are not part of it."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1]
out_dir = "/tmp/trip_table"
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "jit_isa.py"), name, out_dir], check=True, capture_output=True)
hip = os.path.join(out_dir, name + ".hip")
tight = os.path.join(out_dir, name + "_tight.hip")
with open(tight, "w") as f:
    f.write("#define RM_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))\n" + open(hip).read())
asm = os.path.join(out_dir, name + "_tight.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm",
                "-amdgpu-atomic-optimizer-strategy=None", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", asm, tight],
               check=True, capture_output=True)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("rm_render_jit:"))
end = next(i for i, l in enumerate(lines) if i > start and l.strip().startswith(".Lfunc_end"))
labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i in range(start, end):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
    if m and labels.get(m.group(1), i) < i:
        loops.append((labels[m.group(1)], i))
key = "v_floor_f32" if any("v_floor_f32" in l for l in lines[start:end]) else "v_rsq_f32"
cands = [(a, b) for a, b in loops if any(key in l for l in lines[a:b])]
a, b = min(cands, key=lambda ab: ab[1] - ab[0])
# widen to the outermost back-edge of the same header (the loop's several latches)
b = max(bb for aa, bb in loops if aa == a)


def klass(op):
    if op.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fma", "v_mac", "v_mad_f32")):
        return "fp32 mul/add/fma"
    if op.startswith(("v_rsq", "v_sqrt", "v_rcp", "v_div_")):
        return "rsq/sqrt/rcp/div steps"
    if op.startswith(("v_max", "v_min", "v_med3")):
        return "min/max/med3"
    if op.startswith("v_floor"):
        return "floor"
    if op.startswith("v_cmp"):
        return "compare"
    if op.startswith("v_cndmask"):
        return "select"
    if op.startswith("v_mov"):
        return "move"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane <-> scalar (spills, uniform values)"
    if op.startswith("v_"):
        return "integer / bit (range guards of the short sqrt and quotient)"
    if op.startswith(("s_cbranch", "s_branch")):
        return "scalar: branch"
    if op.startswith("s_"):
        return "scalar: other"
    return "memory / other"


counts, listing = {}, []
for l in lines[a:b + 1]:
    t = l.strip()
    if not t or t.startswith((";", ".")) and not re.match(r"^\.LBB", t):
        continue
    if re.match(r"^\.LBB", t):
        listing.append(t.split(";")[0].strip())
        continue
    op = t.split()[0]
    k = klass(op)
    counts[k] = counts.get(k, 0) + 1
    listing.append("    %-58s ; %s" % (t.split(";")[0].strip()[:58], k))
vec = sum(v for k, v in counts.items() if not k.startswith(("scalar", "memory")))
text = ["# %s.sce: the march loop of the generated kernel, 8-wave build (hipcc offline, gfx950), lines %d-%d of the assembly" % (name, a, b),
        "# every instruction inside the loop's outermost back-edge, ALL paths together (primary trips, shadow trips, the",
        "# general tail, the carved-out solids behind their bound): a trip executes one path through it", ""]
for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
    text.append("%-62s %5d" % (k, v))
text += ["%-62s %5d" % ("vector instructions, all paths", vec), "", "## listing"] + listing
out = sys.argv[2] if len(sys.argv) > 2 else None
if out:
    open(out, "w").write("\n".join(text) + "\n")
print("\n".join(text[:len(counts) + 6]))
