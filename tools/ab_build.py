#!/usr/bin/env python3
"""Builds an A/B variant of the library into tools/bin/lib_<tag>.so (objects under kzg_amd/build/<tag>/):
   python tools/ab_build.py <tag> [--keep-nops] [-DNAME[=VALUE] ...]
Compare same-box with tools/ab_libs.py tools/bin/lib_A.so tools/bin/lib_B.so."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kzg_amd import build as kb  # noqa: E402

tag = sys.argv[1]
keep = "--keep-nops" in sys.argv
defs = [a[2:] for a in sys.argv[2:] if a.startswith("-D")]
os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
out = os.path.join(ROOT, "tools", "bin", f"lib_{tag}.so")
print(kb.build(out=out, defines=defs, strip_nops=not keep, tag=tag, verbose="-v" in sys.argv))
