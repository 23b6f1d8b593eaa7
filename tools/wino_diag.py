"""Runs the diagnostic build of the Winograd conv2 kernel (CS_WINO_DIAG=1: s_memtime stamps summed per wave)
and prints the phase shares of the last launch.  Diagnostic only; run on an MI355X from the repo root."""
import os, sys, ctypes as C
os.environ["CS_WINO_DIAG"] = "1"
sys.path.insert(0, "cell-image-analysis_amd")
import numpy as np, torch
from cellscreen import synth, _lib
from cellscreen.engine import Engine
e = Engine.from_weights(synth.random_cae(42))
x = torch.empty((32768, 64, 64), dtype=torch.float32, device="cuda")
e.synth_crops(42, 0, x)
e.set_chunk(32768)
f = e.encode(x, which=0)
lib = _lib.load_library()
out = (C.c_double * 4)()
rc = lib.cs_debug_wino_diag(out)
v = list(out)
tot = sum(v)
print("rc", rc, "per-wave cycles (s_memtime ticks): prep %.0f  mfma %.0f  epilogue %.0f  barrier %.0f" % tuple(v))
print("shares: prep %.1f%%  mfma %.1f%%  epi %.1f%%  barrier %.1f%%" % tuple(100 * a / tot for a in v))
items = 32768 * 8 / 512
print("items per WG %.1f; ticks per item %.0f; per tile-row %.0f (MFMA issue alone = 128 x 32 = 4096 cycles)" % (items, tot / items, tot / items / 2))
