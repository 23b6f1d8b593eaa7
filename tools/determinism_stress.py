"""Run-to-run determinism of the screening path on one GPU: every launch of the same library on the same crops has to give the
same bits.  A kernel whose result depends on how its waves happen to interleave (a missing barrier, an instruction hazard the
compiler does not see inside inline asm) shows up here and nowhere else: such faults hit one value in tens of thousands of cells,
far below any parity tolerance that is checked on a few hundred cells.  Round 4 found one this way.

    python tools/determinism_stress.py [--reps 40] [--cells 8192] [--precision split16|fp32_exact] [--lib path/to/libcellscreen.so]

Compares, against the first run: the output of every encoder stage (layers 0..3) and the six per-cell results of cs_screen.
Prints one line per array and exits 1 if any run differed.  Not a test of correctness (tests/ does that) and not a benchmark."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--cells", type=int, default=8192)
ap.add_argument("--precision", default="split16")
ap.add_argument("--lib", default=None)
args = ap.parse_args()
from cellscreen import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from cellscreen import synth
from cellscreen.engine import Engine

import helpers as H

w = synth.random_cae(seed=42)
det = H.det_from_golden(np.load(os.path.join(ROOT, "tests", "golden", "golden_detector.npz")))      # the fitted detector of the golden vectors
e = Engine.from_weights(w, None, det, precision=args.precision)
x = torch.empty((args.cells, 64, 64), dtype=torch.float32, device="cuda")
e.synth_crops(42, 0, x)


def run():
    out = {f"layer{l}": e.layer_output(x, l).cpu().numpy() for l in range(4)}
    for k, v in e.screen(x).items():
        out["screen_" + k] = v.cpu().numpy()
    return out


ref = run()
bad = {k: 0 for k in ref}
for r in range(args.reps):
    cur = run()
    for k in ref:
        d = cur[k].view(f"u{cur[k].dtype.itemsize}") != ref[k].view(f"u{ref[k].dtype.itemsize}")          # bits, so that NaN == NaN
        if d.any():
            bad[k] += 1
            if bad[k] <= 3:
                idx = np.argwhere(d)
                print(f"rep {r}: {k} differs in {int(d.sum())} elements, first at {idx[0].tolist()}: {cur[k][tuple(idx[0])]} vs {ref[k][tuple(idx[0])]}", flush=True)
for k, v in bad.items():
    print(f"{k:24s} {args.cells} cells x {args.reps} runs: {'identical' if v == 0 else f'{v} runs DIFFER from the first'}")
sys.exit(1 if any(bad.values()) else 0)
