"""A/B helper (development only): run a fixed set of calls through the library at argv[1] and save every output to argv[2].npz,
so that two builds can be compared bit for bit by tools/scratch/ab/compare.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cellscreen import _lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
from cellscreen import synth
from cellscreen.engine import Engine
import helpers as H
g = np.load(os.path.join(ROOT, "tests", "golden", "golden_detector.npz"))
det = H.det_from_golden(g)
w = synth.random_cae(seed=42)
x = synth.synth_crops(11, 7000, 700)
x[0] = 0.0; x[1] = 1.0; x[2, ::2] = 0.0; x[3, :, ::2] = 0.0; x[4:40] = synth.blob_crops(5, 36); x[41] *= 255.0; x[42] *= 1e-6
out = {}
for prec in ("split16", "fp32_exact"):
    e = Engine.from_weights(w, None, det, precision=prec)
    for l in (1, 2, 4):
        out[f"{prec}_layer{l}"] = e.layer_output(x, l)
    r = e.screen(x)
    for k, v in r.items():
        out[f"{prec}_{k}"] = v
    e.close()
np.savez(sys.argv[2], **out)
print("dumped", len(out), "arrays with", sys.argv[1])
