#!/usr/bin/env python3
"""Run a script of this repo against another build of libkzg_mi355x.so (same ABI):
   python tools/run_with_lib.py tools/bin/lib_X.so tools/skew_probe.py [arguments...]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_path = os.path.abspath(sys.argv[1])
script = os.path.abspath(sys.argv[2])
sys.argv = [script] + sys.argv[3:]
import torch  # noqa: E402,F401
from kzg_amd import _lib as L  # noqa: E402

L.load(lib_path)
runpy.run_path(script, run_name="__main__")
