import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from cellscreen import synth, preprocess as pp
from cellscreen.detector_fit import fit_detector
from cellscreen.engine import Engine
w = synth.random_cae(seed=42)
enc = Engine.from_weights(w)
xt = torch.empty((2000, 64, 64), dtype=torch.float32, device="cuda"); enc.synth_crops(42, 10**10, xt); torch.cuda.synchronize()
det, _ = fit_detector(enc.encode(xt, which=0).cpu().numpy(), pca_random_state=0); enc.close()
eng = Engine.from_weights(w, None, det); eng.set_chunk(65536)
n = 1_000_000; chunk = 65536
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
bounds = [(i, min(i + chunk, n)) for i in range(0, n, chunk)]
span = max(int(off[b - 1] + sizes[b - 1] - off[a]) for a, b in bounds)
d_pix = [torch.empty(span, dtype=torch.int16, device="cuda") for _ in range(2)]
d_crops = torch.empty((chunk, 64, 64), dtype=torch.float32, device="cuda")
out = dict(mse=torch.empty(n, dtype=torch.float32, device="cuda"), mae=torch.empty(n, dtype=torch.float32, device="cuda"),
           cons_score=torch.empty(n, dtype=torch.float64, device="cuda"), mod_score=torch.empty(n, dtype=torch.float64, device="cuda"),
           cons_pred=torch.empty(n, dtype=torch.int8, device="cuda"), mod_pred=torch.empty(n, dtype=torch.int8, device="cuda"))
proc = pp.Preprocessor(0)
cs, ms = torch.cuda.Stream(), torch.cuda.Stream()
evs = [torch.cuda.Event() for _ in range(2)]
def copy_in(ci):
    a, b = bounds[ci]; lo, hi = int(off[a]), int(off[b - 1] + sizes[b - 1])
    with torch.cuda.stream(cs):
        d_pix[ci & 1][:hi - lo].copy_(host[lo:hi], non_blocking=True); evs[ci & 1].record(cs)
kms = []
def run(do_copy, do_pre, do_screen):
    tl = []
    kms.clear()
    with torch.cuda.stream(ms):
        if do_copy: copy_in(0)
        for ci, (a, b) in enumerate(bounds):
            t0 = time.perf_counter()
            if do_copy and ci + 1 < len(bounds): copy_in(ci + 1)
            t1 = time.perf_counter()
            if do_copy: ms.wait_event(evs[ci & 1])
            if do_pre:
                proc.run_packed(d_pix[ci & 1], off[a:b] - off[a], hs[a:b], ws[a:b], out=d_crops[:b - a])
                kms.append(proc.last_timing()[0])
            t2 = time.perf_counter()
            if do_screen: eng.screen(d_crops[:b - a], out={k: v[a:b] for k, v in out.items()}, out_device=True)
            t3 = time.perf_counter()
            tl.append((t1 - t0, t2 - t1, t3 - t2))
    torch.cuda.synchronize()
    return tl
for name, flags in (("all", (1, 1, 1)), ("no copy", (0, 1, 1)), ("copy+pre", (1, 1, 0))):
    run(*flags)
    t0 = time.perf_counter(); tl = run(*flags); dt = time.perf_counter() - t0
    a = np.array(tl)
    print("kernel ms per chunk (events):", np.round(kms[2:6], 2) if kms else None)
    print("%-12s wall %.4f s; per chunk (ms): enqueue copy %.2f, preprocess call %.2f, screen call %.2f" % (name, dt, *(a[2:-1].mean(axis=0) * 1e3)), flush=True)
