"""Throughput of the generic-shape path on BASELINE.json configs[4] (128x128 crops, filters 32-64-128 |
128-64-32-1; 349.18 M MAC/cell): autoencoder forward + reconstruction error,
crops resident in HBM.  Not a bench line (bench.py is); prints one JSON object."""
import json, sys, time
sys.path.insert(0, "cell-image-analysis_amd")
import numpy as np, torch
from cellscreen import synth
from cellscreen.engine import Engine

HW, CH = (128, 128), (32, 64, 128, 128, 64, 32, 1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = synth.random_cae(seed=5, hw=HW, channels=CH, n_enc=3)
e = Engine.from_weights(w)
x = torch.rand((n, *HW), dtype=torch.float32, device="cuda")
e.set_chunk(4096)
e.reconstruct(x, want_recon=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 3
for _ in range(steps):
    e.reconstruct(x, want_recon=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
e.profile_enable(True); e.profile_reset()
e.reconstruct(x, want_recon=False)
pr = {k: v for k, v in e.profile().items() if v["launches"]}
prof = {k: round(v["ms"], 3) for k, v in pr.items()}
e.profile_enable(False)
# kernels on the split-bf16 path: v_mfma_f32_16x16x32_bf16 count x 16,384 FLOP against the 2.5 PFLOP/s dense bf16 peak
bf16 = {k: {"bf16_mfma_per_cell": v["bf16_mfma_per_cell"], "tflops_bf16": round(v["bf16_mfma_per_cell"] * 16384 * v["cells"] / (v["ms"] * 1e-3) / 1e12, 1),
            "frac_bf16_mfma_peak": round(v["bf16_mfma_per_cell"] * 16384 * v["cells"] / (v["ms"] * 1e-3) / 2.5e15, 4)}
        for k, v in pr.items() if v.get("bf16_mfma_per_cell", 0) > 0}
x3 = bool(bf16)
macs = 349.18e6
# multiply-adds the matrix pipe executes: conv1 pads K = 9 to 12, the upsample-fed conv5 / conv6 run folded (4/9), the 1-filter
# conv7 runs on the vector ALU (not counted)
c = [128 * 128 * 9 * 1 * 32, 64 * 64 * 9 * 32 * 64, 32 * 32 * 9 * 64 * 128, 16 * 16 * 9 * 128 * 128, 32 * 32 * 9 * 128 * 64, 64 * 64 * 9 * 64 * 32]
folded = True
exec_macs = c[0] * 12 / 9 + c[1] + c[2] + c[3] + (c[4] + c[5]) * (4 / 9 if folded else 1.0)
print(json.dumps({"workload": f"{n} crops 128x128, filters {CH}, CAE forward + reconstruction MSE/MAE", "cells_per_s": n / dt,
                  "ms_per_step": dt * 1e3, "tflops_algorithmic": 2 * macs * n / dt / 1e12, "frac_fp32_mfma_peak_algorithmic": 2 * macs * n / dt / 157.3e12,
                  # with the split-bf16 kernels the fp32-MFMA pricing does not apply (see split_bf16_kernels instead)
                  "tflops_executed_mfma": None if x3 else 2 * exec_macs * n / dt / 1e12, "frac_fp32_mfma_peak": None if x3 else 2 * exec_macs * n / dt / 157.3e12,
                  "split_bf16_kernels": bf16, "kernel_ms_one_step": prof}))
