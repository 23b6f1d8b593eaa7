import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
from cellscreen import synth, preprocess as pp
from cellscreen.detector_fit import fit_detector
from cellscreen.engine import Engine
w = synth.random_cae(seed=42)
enc = Engine.from_weights(w)
xt = torch.empty((2000, 64, 64), dtype=torch.float32, device="cuda"); enc.synth_crops(42, 10**10, xt); torch.cuda.synchronize()
det, _ = fit_detector(enc.encode(xt, which=0).cpu().numpy(), pca_random_state=0); enc.close()
eng = Engine.from_weights(w, None, det); eng.set_chunk(65536)
n = 1_000_000
print("e2e", json.dumps(bench.e2e_raw_leg(eng, n, 42, 0))[:300], flush=True)
# components
base = synth.raw_crops(42, 4096, np.uint16, 32, 100)
bpix, boff, bhs, bws = pp.pack_crops(base)
reps = (n + 4095) // 4096
hs, ws = np.tile(bhs, reps)[:n], np.tile(bws, reps)[:n]
sizes = hs.astype(np.int64) * ws.astype(np.int64); off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64); total = int(sizes.sum())
host = torch.empty(total, dtype=torch.int16, pin_memory=True)
hv = host.numpy().view(np.uint16)
for r in range(reps):
    lo = r * len(bpix); m = min(len(bpix), total - lo)
    if m > 0: hv[lo:lo + m] = bpix[:m]
dev = torch.empty(total, dtype=torch.int16, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dev.copy_(host, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("H2D whole buffer: %.1f GB/s (%.3f s)" % (2 * total / dt / 1e9, dt), flush=True)
chunk = 65536
t0 = time.perf_counter()
for a in range(0, n, chunk):
    b = min(a + chunk, n); lo, hi = int(off[a]), int(off[b - 1] + sizes[b - 1])
    dev[lo:hi].copy_(host[lo:hi], non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("H2D in 16 chunks: %.1f GB/s" % (2 * total / dt / 1e9), flush=True)
proc = pp.Preprocessor(0)
d_crops = torch.empty((chunk, 64, 64), dtype=torch.float32, device="cuda")
for rep in range(2):
    t0 = time.perf_counter(); tp = 0.0
    for a in range(0, n, chunk):
        b = min(a + chunk, n); lo, hi = int(off[a]), int(off[b - 1] + sizes[b - 1])
        t1 = time.perf_counter()
        proc.run_packed(dev[lo:hi], off[a:b] - off[a], hs[a:b], ws[a:b], out=d_crops[:b - a])
        tp += time.perf_counter() - t1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("preprocess only (resident pixels): %.3f s = %.2f M crops/s" % (dt, n / dt / 1e6), flush=True)
out = dict(mse=torch.empty(chunk, dtype=torch.float32, device="cuda"), mae=torch.empty(chunk, dtype=torch.float32, device="cuda"),
           cons_score=torch.empty(chunk, dtype=torch.float64, device="cuda"), mod_score=torch.empty(chunk, dtype=torch.float64, device="cuda"),
           cons_pred=torch.empty(chunk, dtype=torch.int8, device="cuda"), mod_pred=torch.empty(chunk, dtype=torch.int8, device="cuda"))
t0 = time.perf_counter()
for a in range(0, n, chunk):
    eng.screen(d_crops[:min(chunk, n - a)], out={k: v[:min(chunk, n - a)] for k, v in out.items()}, out_device=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("screen only in 16 chunk calls: %.3f s = %.2f M/s" % (dt, n / dt / 1e6), flush=True)
