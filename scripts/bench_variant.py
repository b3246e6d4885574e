#!/usr/bin/env python
"""Run bench.py's measurement against an experimental build of the library (timing experiments only):
KPN_EXPERIMENT_LIB=path/to/lib.so python scripts/bench_variant.py [bench.py flags]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (before the library)
from keypointnerf_amd import lib as kl  # noqa: E402

kl._default = kl.KpnLibrary(os.environ["KPN_EXPERIMENT_LIB"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
