"""Per-kernel device time of the fp32_exact precision on 262,144 resident crops (development: where does the exact mode's step go?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from cellscreen import synth
from cellscreen.engine import Engine
n = 262144
det = H.det_from_golden(np.load(os.path.join(ROOT, "tests", "golden", "golden_detector.npz")))
for prec in ("fp32_exact", "split16"):
    e = Engine.from_weights(synth.random_cae(seed=42), None, det, precision=prec)
    x = torch.empty((n, 64, 64), dtype=torch.float32, device="cuda")
    e.synth_crops(42, 0, x)
    e.screen(x)
    e.profile_enable(True); e.profile_reset()
    e.screen(x)
    pr = {k: v for k, v in e.profile().items() if v["launches"]}
    tot = sum(v["ms"] for v in pr.values())
    print(prec, "total", round(tot * 1e6 / n, 1), "ms per 1 M cells")
    for k, v in sorted(pr.items(), key=lambda kv: -kv[1]["ms"]):
        mf = v.get("mfma_per_cell", 0); b16 = v.get("bf16_mfma_per_cell", 0)
        ms1m = v["ms"] * 1e6 / n
        tf32 = mf * 2048 * 1e6 / (ms1m * 1e-3) / 1e12 if mf else 0
        print(f"   {k:26s} {ms1m:7.1f} ms/1M  fp32 MFMA/cell {mf:7.0f} ({tf32:6.1f} TF = {tf32 / 157.3:5.2f} of peak)  16-bit MFMA/cell {b16:6.0f}")
    e.close(); del x
