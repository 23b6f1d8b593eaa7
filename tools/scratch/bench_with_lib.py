"""bench.py against another build of the library (same-box A/B): python tools/scratch/bench_with_lib.py <path.so> [bench args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd")); sys.path.insert(0, ROOT)
import cellscreen._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
