#!/usr/bin/env python3
"""A/B of an environment knob on bench.py workloads, interleaved: ab_env.py KNOB=a,b[,c] [workload ...]
Each (value, workload) runs bench.py in a process of its own (the knob may select generated code), `--rounds`
times in turn, and the median ms per frame is printed per value.  GPU box only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    knob, values = args[0].split("=", 1)
    values = values.split(",")
    workloads = args[1:] or ["chair_4k"]
    rounds = int(opts.get("rounds", 3))
    steps = opts.get("steps", "160")
    extra = opts.get("extra", "").split()
    res = {}
    for r in range(rounds):
        for wl in workloads:
            for v in values:
                env = dict(os.environ)
                env[knob] = v
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", steps, "--warmup", "16",
                       "--no-cpu-baseline"] + extra
                out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
                if out.returncode != 0:
                    print(wl, v, "FAILED", out.stderr[-500:], flush=True)
                    continue
                rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
                one = rec.get("single_frame_launch", {}).get("kernel_ms")
                res.setdefault((wl, v), []).append((rec["ms_per_step"], one, rec["parity"]["frame_matches_golden"]))
                print("round %d %s %s=%s: %.4f ms/frame, single %s, golden %s" % (r, wl, knob, v, rec["ms_per_step"], one,
                                                                                rec["parity"]["frame_matches_golden"]), flush=True)
    for (wl, v), rows in sorted(res.items()):
        ms = sorted(x[0] for x in rows)
        ones = sorted(x[1] for x in rows if x[1] is not None)
        print("%-16s %s=%-6s median %.4f ms/frame (min %.4f)  single-frame median %s" %
              (wl, knob, v, ms[len(ms) // 2], ms[0], "%.4f" % ones[len(ones) // 2] if ones else "-"))


if __name__ == "__main__":
    main()
