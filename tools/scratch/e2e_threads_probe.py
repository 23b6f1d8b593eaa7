"""Does preprocess(chunk i + 1) overlap screen(chunk i) when two host threads drive the two handles?  (bench.py's e2e_raw leg runs them
one after the other on one thread: both calls end in a stream synchronise.)"""
import os, sys, time, json, threading, queue
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cell-image-analysis_amd"), ROOT):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import numpy as np
import torch
import bench
from cellscreen import synth, preprocess as pp
from cellscreen.engine import Engine
from cellscreen.detector_fit import fit_detector
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
weights = synth.random_cae(seed=42)
enc = Engine.from_weights(weights, device_id=0)
xt = torch.empty((5000, 64, 64), dtype=torch.float32, device=dev)
enc.synth_crops(42, 10_000_000_000, xt); torch.cuda.synchronize()
feats = enc.encode(xt, which=0).cpu().numpy(); enc.close()
det, _ = fit_detector(feats, pca_random_state=0)
eng = Engine.from_weights(weights, None, det, device_id=0); eng.set_chunk(65536)
n, chunk = 1_000_000, 65536
print(json.dumps({"one_thread": bench.e2e_raw_leg(eng, n, 42, 0)["value"]}), flush=True)

base = synth.raw_crops(42, 4096, np.uint16, 32, 100)
bpix, boff, bhs, bws = pp.pack_crops(base)
reps = (n + len(base) - 1) // len(base)
hs, ws = np.tile(bhs, reps)[:n], np.tile(bws, reps)[:n]
sizes = hs.astype(np.int64) * ws.astype(np.int64)
off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
total = int(sizes.sum())
host = torch.empty(total, dtype=torch.int16, pin_memory=True)
hv = host.numpy().view(np.uint16)
for r in range(reps):
    lo = r * len(bpix); m = min(len(bpix), total - lo)
    if m > 0: hv[lo:lo + m] = bpix[:m]
proc = pp.Preprocessor(0)
copy_stream = torch.cuda.Stream(device=dev)
bounds = [(i, min(i + chunk, n)) for i in range(0, n, chunk)]
span = max(int(off[b - 1] + sizes[b - 1] - off[a]) for a, b in bounds)
d_pix = [torch.empty(span, dtype=torch.int16, device=dev) for _ in range(3)]
d_crops = [torch.empty((chunk, 64, 64), dtype=torch.float32, device=dev) for _ in range(2)]
out = dict(mse=torch.empty(n, dtype=torch.float32, device=dev), mae=torch.empty(n, dtype=torch.float32, device=dev),
           cons_score=torch.empty(n, dtype=torch.float64, device=dev), mod_score=torch.empty(n, dtype=torch.float64, device=dev),
           cons_pred=torch.empty(n, dtype=torch.int8, device=dev), mod_pred=torch.empty(n, dtype=torch.int8, device=dev))
res = {k: torch.empty(n, dtype=v.dtype, pin_memory=True) for k, v in out.items()}

def run():
    up_done = [torch.cuda.Event() for _ in bounds]
    pix_free = [threading.Semaphore(0) for _ in bounds]       # d_pix[ci % 3] free again (its preprocess has returned)
    crops_ready, crops_free = queue.Queue(), threading.Semaphore(2)
    def uploader():
        torch.cuda.set_device(0)
        for ci, (a, b) in enumerate(bounds):
            if ci >= 3: pix_free[ci - 3].acquire()
            lo, hi = int(off[a]), int(off[b - 1] + sizes[b - 1])
            with torch.cuda.stream(copy_stream):
                d_pix[ci % 3][:hi - lo].copy_(host[lo:hi], non_blocking=True)
                up_done[ci].record(copy_stream)
            up_done[ci].synchronize()
            up_q.put(ci)
    def preprocessor():
        torch.cuda.set_device(0)
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            for ci, (a, b) in enumerate(bounds):
                assert up_q.get() == ci
                crops_free.acquire()
                proc.run_packed(d_pix[ci % 3], off[a:b] - off[a], hs[a:b], ws[a:b], out=d_crops[ci & 1][:b - a])
                pix_free[ci].release()
                crops_ready.put(ci)
    def screener():
        torch.cuda.set_device(0)
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            for ci, (a, b) in enumerate(bounds):
                assert crops_ready.get() == ci
                eng.screen(d_crops[ci & 1][:b - a], out={k: v[a:b] for k, v in out.items()}, out_device=True)
                crops_free.release()
            for k in out: res[k].copy_(out[k], non_blocking=True)
            s.synchronize()
    up_q = queue.Queue()
    ts = [threading.Thread(target=f) for f in (uploader, preprocessor, screener)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    return time.perf_counter() - t0
run()
dt = min(run(), run())
print(json.dumps({"three_threads": round(n / dt, 1), "wall_s": round(dt, 4), "rate_neg": float((res["cons_pred"] == -1).float().mean())}), flush=True)
proc.close(); eng.close()
