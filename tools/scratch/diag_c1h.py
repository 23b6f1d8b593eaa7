import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd")); sys.path.insert(0, ROOT)
from cellscreen import synth
from cellscreen.engine import Engine
from oracle import oracle
w = synth.random_cae(seed=42)
x = oracle.synth_crops(7, 100, 3)
x[1] = 0.5
x[2] = 0.0; x[2, 10, 20] = 1.0          # a single pixel
ref = oracle.cae_forward(w, x, acc64=True, want=("features",), layers=True)["layers"][1]
e = Engine.from_weights(w)
got = e.layer_output(x, 1)
e.close()
for c in range(3):
    d = np.abs(got[c] - ref[c])
    print("cell", c, "max err", d.max(), "of", np.abs(ref[c]).max())
    bad = np.argwhere(d > 1e-4 * np.abs(ref[c]).max())
    print("  bad count", len(bad), "of", d.size, "first", bad[:6].tolist())
    if len(bad):
        ys = sorted(set(bad[:, 0].tolist())); xs = sorted(set(bad[:, 1].tolist())); cs = sorted(set(bad[:, 2].tolist()))
        print("  rows", ys[:20], "cols", xs[:20], "channels", cs[:20], len(cs))
np.set_printoptions(precision=5, suppress=True, linewidth=200)
for (y, xx) in ((0, 0), (0, 5), (5, 0), (1, 5), (15, 5)):
    print("cell1 p2[%d,%d,:8] got" % (y, xx), got[1][y, xx, :8], "ref", ref[1][y, xx, :8])
# conv1 alone through the fused kernel cannot be tapped; compare the two-kernel p1 (reference path) for sanity
os.environ["CS_NO_FP16X2_CONV1"] = "1"
e = Engine.from_weights(w); g2 = e.layer_output(x, 1); e.close()
print("with CS_NO_FP16X2_CONV1: max err", [float(np.abs(g2[c] - ref[c]).max()) for c in range(3)])
