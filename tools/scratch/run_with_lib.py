"""Run a repo script against another build of the library (same-box A/B): python tools/scratch/run_with_lib.py <path.so> <script.py> [args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd")); sys.path.insert(0, ROOT)
import cellscreen._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
script = os.path.join(ROOT, sys.argv[2])
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
