#!/usr/bin/env python3
"""Same-box A/B helper: run bench.py against another build of libkzg_mi355x.so (same ABI).
   python tools/bench_with_lib.py tools/bin/lib_X.so [bench.py arguments...]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
lib_path = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
from kzg_amd import _lib as L  # noqa: E402

L.load(lib_path)
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
