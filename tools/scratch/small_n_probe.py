"""Where does a 128-cell cs_screen call spend its 0.67 ms?"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cell-image-analysis_amd"), ROOT):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import numpy as np
import torch
from cellscreen import synth
from cellscreen.engine import Engine
from cellscreen.detector_fit import fit_detector
dev = torch.device("cuda", 0)
weights = synth.random_cae(seed=42)
enc = Engine.from_weights(weights, device_id=0)
xt = torch.empty((5000, 64, 64), dtype=torch.float32, device=dev)
enc.synth_crops(42, 10_000_000_000, xt); torch.cuda.synchronize()
feats = enc.encode(xt, which=0).cpu().numpy(); enc.close()
det, _ = fit_detector(feats, pca_random_state=0)
eng = Engine.from_weights(weights, None, det, device_id=0)
def med(f, reps=20):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 4)
for n in (128, 1024):
    x = synth.synth_crops(42, 5_000_000, n)
    xd = torch.from_numpy(x).to(dev)
    xp = torch.from_numpy(x).pin_memory()
    res = {}
    res["host_pageable_ms"] = med(lambda: eng.screen(x))
    res["host_pinned_ms"] = med(lambda: eng.screen(xp.numpy()))
    out = None
    def dev_call():
        eng.screen(xd, out_device=True); torch.cuda.synchronize()
    res["device_in_device_out_ms"] = med(dev_call)
    def dev_host():
        eng.screen(xd, out_device=False)
    res["device_in_host_out_ms"] = med(dev_host)
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(10): eng.screen(xd, out_device=True)
    torch.cuda.synchronize(); eng.profile_enable(False)
    pr = eng.profile()
    res["kernel_ms_sum"] = round(sum(v["ms"] for v in pr.values()) / 10, 4)
    res["kernels_us"] = {k: round(v["ms"] / 10 * 1e3, 1) for k, v in pr.items() if v["launches"]}
    print(json.dumps({"n": n, **res}), flush=True)
eng.close()
