#!/usr/bin/env python3
"""bench.py with developer knobs from the environment (A/B runs on one device): PBE_TUNE="key=value,key=value" -> pbe_tune(key, value)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pbe_amd import ops  # noqa: E402

for kv in filter(None, os.environ.get("PBE_TUNE", "").split(",")):
    k, v = kv.split("=")
    ops.tune(int(k), int(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
