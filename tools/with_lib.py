"""Run a script against a variant build of the library (tools/build_variant.sh): A/B timing inside one gpurun call.

    python tools/with_lib.py tools/bin/libddsp_amd_<name>.so tools/bench_reverb.py 128 64000 48000 1
"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddsp_amd import _lib, build
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
build.build = lambda *a, **k: _lib.LIB_PATH            # (the scripts call build.build(): the variant is what it is)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
