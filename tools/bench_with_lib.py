#!/usr/bin/env python3
"""bench.py against another build of the library (A/B on the same box): bench_with_lib.py <libgb25hip.so> [bench.py arguments]"""
import os, runpy, sys
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, here)
os.environ["GB25_LIB"] = "1"           # (no rebuild check: the file is what it is)
import gb25_amd.binding as _bind
_bind.LIB_PATHS["Float32"] = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(here, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
