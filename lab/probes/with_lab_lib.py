#!/usr/bin/env python3
"""Run a probe script against the lab / profile build of the library (csrc/libpiccolo_hip_lab.so, built with `_lib.build_library(profile=True)`): the options of
experiments that give WRONG results (`v4_variant`, `profile_flags`) exist only there.  usage: with_lab_lib.py <script.py> [args ...]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
assert os.path.exists(pa._lib.SO_PATH_LAB), "build it first: python -c 'import piccolo_jl_amd as pa; pa._lib.build_library(profile=True)'"
pa._lib._variant_path = pa._lib.SO_PATH_LAB
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
