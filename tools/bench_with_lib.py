#!/usr/bin/env python3
"""Tuning aid: run bench.py against another build of the library (A/B runs on one box).
usage: tools/bench_with_lib.py <lib.so> [bench args...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrhash_amd import capi
capi.HIP_LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
