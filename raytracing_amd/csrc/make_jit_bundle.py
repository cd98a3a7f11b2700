#!/usr/bin/env python3
"""Flattens the device headers into one raw-string literal (jit_bundle.inc) that
jit.cpp hands to hiprtc at run time: local includes are expanded once, system
includes and `#pragma once` are dropped (hiprtc has no system headers)."""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
seen = set()


def expand(path, out):
    path = os.path.normpath(path)
    if path in seen:
        return
    seen.add(path)
    out.append("// ---- %s\n" % os.path.relpath(path, HERE))
    for line in open(path):
        m = re.match(r'\s*#\s*include\s+"([^"]+)"', line)
        if m:
            expand(os.path.join(os.path.dirname(path), m.group(1)), out)
        elif re.match(r"\s*#\s*include\s+<", line) or re.match(r"\s*#\s*pragma\s+once", line):
            continue
        else:
            out.append(line)


def main():
    out = []
    for name in sys.argv[2:]:
        expand(os.path.join(HERE, name), out)
    text = "".join(out)
    assert ')RMJIT"' not in text
    with open(sys.argv[1], "w") as f:
        f.write('R"RMJIT(\n' + text + ')RMJIT"\n')


if __name__ == "__main__":
    main()
