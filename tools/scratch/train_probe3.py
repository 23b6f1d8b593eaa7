"""Which of bench.py's earlier legs slows the training leg that follows?  argv: comma list of {fit,engine,screen,small,exact,e2e}"""
import os, sys, time, json, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cell-image-analysis_amd"), ROOT):
    sys.path.insert(0, p)
import bench
what = set((sys.argv[1] if len(sys.argv) > 1 else "").split(","))
import numpy as np
import torch
from cellscreen import synth
from cellscreen.engine import Engine
from cellscreen.detector_fit import fit_detector
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
weights = synth.random_cae(seed=42)
eng = None
if what & {"fit", "engine", "screen", "small", "exact", "e2e"}:
    enc = Engine.from_weights(weights, device_id=0)
    xt = torch.empty((5000, 64, 64), dtype=torch.float32, device=dev)
    enc.synth_crops(42, 10_000_000_000, xt)
    torch.cuda.current_stream().synchronize()
    feats = enc.encode(xt, which=0).cpu().numpy()
    enc.close(); del xt
    det, sk = fit_detector(feats, pca_random_state=0)
if what & {"engine", "screen", "small", "exact", "e2e"}:
    eng = Engine.from_weights(weights, None, det, device_id=0)
    eng.set_chunk(65536)
args = types.SimpleNamespace(chunk=65536)
if what & {"screen", "exact"}:
    x = torch.empty((200000, 64, 64), dtype=torch.float32, device=dev)
    eng.synth_crops(42, 0, x)
    out = dict(mse=torch.empty(len(x), dtype=torch.float32, device=dev), mae=torch.empty(len(x), dtype=torch.float32, device=dev),
               cons_score=torch.empty(len(x), dtype=torch.float64, device=dev), mod_score=torch.empty(len(x), dtype=torch.float64, device=dev),
               cons_pred=torch.empty(len(x), dtype=torch.int8, device=dev), mod_pred=torch.empty(len(x), dtype=torch.int8, device=dev))
    eng.profile_enable(True); eng.screen(x, out=out, out_device=True); torch.cuda.synchronize(); eng.profile_enable(False)
if "small" in what:
    bench.small_n_leg(eng, 42)
if "exact" in what:
    bench.exact_fp32_leg(weights, det, x, out, args, 0)
if "e2e" in what:
    e2e = bench.e2e_raw_leg(eng, 400000, 42, 0)["value"]
r = bench.train_leg(400, 0, 42)
e2e = globals().get("e2e")
print(json.dumps(dict(pre=sorted(what), ms_per_step=r["ms_per_step"], enq=r["host_enqueue_ms_per_step"], e2e=e2e, hwq=os.environ.get("GPU_MAX_HW_QUEUES"))), flush=True)
