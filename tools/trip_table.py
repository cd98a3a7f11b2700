#!/usr/bin/env python3
"""trip_table.py SCENE [out.txt]: the march loops of a scene's generated kernel (8-wave build, compiled offline
for gfx950), instruction by instruction and by class.  No GPU needed.  The loops are the innermost loops of the
assembly that evaluate the scene (a v_floor_f32 or v_rsq_f32 in them): one for primary-only trips, one for
shadow-only trips, every block the compiler attributes to them (the IEEE fall-backs of the short sqrt and quotient
included: a trip executes one path through the listing)."""
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
# The compiler marks innermost loop headers and says of every other block which loop it is in ("in Loop:
# Header=BB0_n"): a loop's body is its header block plus the blocks that name it, wherever they were placed (the
# latches of the loops that run under the marching lanes' execution mask sit in front of their header; the IEEE
# fall-backs of the short sqrt and quotient behind the loop).
blocks, cur = [], None
for i in range(start, end):
    m = re.match(r"^\.L(BB\d+_\d+):", lines[i])
    if m:
        cur = {"label": m.group(1), "first": i, "lines": [], "header": None, "inner": False}
        blocks.append(cur)
    if cur is None:
        continue
    if "This Inner Loop Header" in lines[i]:
        cur["inner"], cur["header"] = True, cur["label"]
    h = re.search(r"in Loop: Header=(BB\d+_\d+)", lines[i])
    if h and cur["header"] is None:
        cur["header"] = h.group(1)
    cur["lines"].append(lines[i])
inner = []
for b in blocks:
    if not b["inner"]:
        continue
    body = [l for c in blocks if c["header"] == b["label"] for l in c["lines"]]
    code = [l for l in body if l.strip() and not l.strip().startswith((";", "."))]
    if any("v_floor_f32" in l or "v_rsq_f32" in l for l in code) and len(code) > 30:
        inner.append((b["label"], body))
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


text = ["# %s.sce: the march loops of the generated kernel, 8-wave build (hipcc offline, gfx950)" % name,
        "# every instruction inside each loop's outermost back-edge, ALL its paths together (e.g. the carved-out solids",
        "# behind their bound, the shadow quotient behind its test): a trip executes one path through it", ""]
for label, body in inner:
    counts, listing = {}, []
    for l in body:
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
    what = "shadow-only trips" if any("v_div_fixup" in l for l in body) else "primary-only trips"
    text.append("## loop of the %s, header %s" % (what, label))
    for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
        text.append("%-62s %5d" % (k, v))
    text += ["%-62s %5d" % ("vector instructions, all paths", vec), "", "### listing"] + listing + [""]
out = sys.argv[2] if len(sys.argv) > 2 else None
if out:
    open(out, "w").write("\n".join(text) + "\n")
print("\n".join(l for l in text if not l.startswith(("    ", ".LBB"))))
