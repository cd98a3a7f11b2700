#!/usr/bin/env python3
"""Offline ISA of a scene's generated kernel: jit_isa.py SCENE [out_dir]
Writes <out>/SCENE.hip (prelude + generated code) and <out>/SCENE.s (gfx950 assembly),
prints the register report.  No GPU needed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import api, scenes

name = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/jit_isa"
os.makedirs(out, exist_ok=True)
if name == "csg1024":
    nodes = scenes.csg1024()
else:
    nodes = api.load_sce(os.path.join(ROOT, "tests", "golden", "scenes", name + ".sce"))
rc, src = api.jit_check(nodes, want_source=True, assume_exact_divisors=True)
assert rc == 0, rc
bundle = open(os.path.join(ROOT, "raytracing_amd", "csrc", "jit_bundle.inc")).read()
prelude = bundle[bundle.index('R"RMJIT(') + 8: bundle.rindex(')RMJIT"')]
hip = os.path.join(out, name + ".hip")
with open(hip, "w") as f:
    f.write("#include <hip/hip_runtime.h>\n#include <stdint.h>\n" + prelude + src)
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "--offload-arch=gfx950",
       "--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(out, name + ".s"), hip]
r = subprocess.run(cmd, capture_output=True, text=True)
print(r.stderr[-3000:])
sys.exit(r.returncode)
