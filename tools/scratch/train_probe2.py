"""Does what ran earlier in the process (other handles, streams, pinned memory) change the asynchronous training epoch?"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cell-image-analysis_amd"), ROOT):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
pre = sys.argv[1] if len(sys.argv) > 1 else "none"
import numpy as np
import torch
from cellscreen import synth
from cellscreen.augment import ImageDataGenerator
from cellscreen.trainer import Trainer
dev = torch.device("cuda", 0)
keep = []
if "streams" in pre:
    for _ in range(int(pre.split(":")[1])):
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            keep.append(torch.zeros(1024, device=dev) + 1)
        keep.append(s)
    torch.cuda.synchronize()
if "pinned" in pre:
    keep.append(torch.empty(4 << 30, dtype=torch.uint8, pin_memory=True))
if "engine" in pre:
    from cellscreen.engine import Engine
    from cellscreen import preprocess as pp
    w = synth.random_cae(seed=1)
    det = synth.random_detector(seed=1) if hasattr(synth, "random_detector") else None
    e = Engine.from_weights(w, None, det, device_id=0) if det is not None else None
    keep.append(e); keep.append(pp.Preprocessor(0))
X = torch.from_numpy(synth.blob_crops(42, 40000)).to(dev)
tr = Trainer(synth.random_cae(seed=42, trivial_bn=True), device_id=0)
gen = ImageDataGenerator(rotation_range=2, width_shift_range=0.02, height_shift_range=0.02, zoom_range=0.02, horizontal_flip=True, vertical_flip=True, fill_mode="nearest")
rng = np.random.default_rng(1)
tg = torch.Generator(device=dev); tg.manual_seed(1)
for rep in range(2):
    steps = 1250
    perm = torch.randperm(40000, device=dev, generator=tg)[:steps * 32].view(steps, 32)
    tr.reset_metrics(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        yb = X[perm[i]].contiguous()
        tr.step_async(tr.augment(yb, gen.random_transforms(32, (64, 64), rng)), yb, 1e-3)
    t_enq = time.perf_counter() - t0
    tr.read_metrics(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(json.dumps(dict(pre=pre, ms_per_step=round(t_all / steps * 1e3, 4), enqueue_ms_per_step=round(t_enq / steps * 1e3, 4))), flush=True)
tr.close()
