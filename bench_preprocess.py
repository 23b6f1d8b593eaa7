"""Side bench of the crop preprocess (cs_preprocess; improved_detection.py:98-99): raw ragged
uint16 bounding-box crops resident in HBM -> float32 [n,64,64] resident in HBM.
Not the headline metric (bench.py is); prints one JSON line of the same shape.

    python bench_preprocess.py [--crops N] [--steps K] [--warmup W] [--min-side A] [--max-side B]

cpu_baseline: the real library the reference calls (scikit-image 0.18.3 under /opt/conda/bin/python3.9,
one core, as the reference's per-crop Python loop runs it) when that interpreter exists on the box,
else the numpy oracle ("port")."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd"))
sys.path.insert(0, ROOT)

SKIMAGE_TIMER = r"""
import sys, time, warnings, numpy as np
warnings.filterwarnings("ignore")
from skimage import exposure
from skimage.transform import resize
d = np.load(sys.argv[1])
crops = [d[k] for k in d.files]
t0 = time.perf_counter()
for c in crops:
    resize(exposure.equalize_adapthist(c, clip_limit=0.02), (64, 64), anti_aliasing=True)
print(len(crops) / (time.perf_counter() - t0))
"""


def cpu_baseline(crops):
    sample = crops[:64]
    conda = "/opt/conda/bin/python3.9"
    if os.path.exists(conda):
        tmp = os.path.join(ROOT, "gpurun_out", "pp_bench_sample.npz")
        os.makedirs(os.path.dirname(tmp), exist_ok=True)
        np.savez(tmp, **{f"c{i}": c for i, c in enumerate(sample)})
        r = subprocess.run([conda, "-c", SKIMAGE_TIMER, tmp], capture_output=True, text=True)
        if r.returncode == 0:
            return {"value": float(r.stdout.strip().splitlines()[-1]), "unit": "crops/s", "cores": 1, "kind": "reference-library",
                    "sample": f"{len(sample)} crops through scikit-image 0.18.3 equalize_adapthist + resize"}
    from oracle import preprocess_oracle as po
    t0 = time.perf_counter()
    po.preprocess_crops(sample[:16])
    return {"value": 16 / (time.perf_counter() - t0), "unit": "crops/s", "cores": 1, "kind": "port",
            "sample": "16 crops through oracle/preprocess_oracle.py"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crops", type=int, default=200_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--min-side", type=int, default=32)
    ap.add_argument("--max-side", type=int, default=100)
    a = ap.parse_args()
    import torch
    from cellscreen import preprocess as pp
    from cellscreen import synth

    base = synth.raw_crops(7, 512, np.uint16, a.min_side, a.max_side)
    crops = [base[i % len(base)] for i in range(a.crops)]
    pix, off, hs, ws = pp.pack_crops(crops)
    d_pix = torch.from_numpy(pix.view(np.int16)).cuda()
    d_out = torch.empty((a.crops, 64, 64), dtype=torch.float32, device="cuda")
    proc = pp.Preprocessor(0)
    for _ in range(a.warmup):
        proc.run_packed(d_pix, off, hs, ws, out=d_out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = 0.0
    for _ in range(a.steps):
        proc.run_packed(d_pix, off, hs, ws, out=d_out)
        kms += proc.last_timing()[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    kms /= a.steps
    bytes_alg = pix.nbytes + d_out.numel() * 4
    line = {"metric": "crops_preprocessed_per_second", "value": a.crops / dt, "unit": "crops/s", "n_gpus": 1,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
            "dtype": "u16 -> f64 -> f32", "data": "synthetic",
            "config": {"workload": f"{a.crops} uint16 crops, sides U[{a.min_side},{a.max_side}], CLAHE(0.02)+resize 64x64",
                       "mean_pixels": float(pix.size / a.crops)},
            "roofline": {"bound": "hbm", "achieved": bytes_alg / (kms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": bytes_alg / (kms * 1e-3) / 1e9 / 8000.0, "traffic": None, "kernel_ms": kms,
                         "note": "algorithmic bytes = raw pixels in + 16 KB fp32 out per crop; the kernel is "
                                 "latency/fp64-ALU bound, far from the HBM roof"},
            "cpu_baseline": cpu_baseline(crops)}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
