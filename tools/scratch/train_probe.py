"""Where does an asynchronous training epoch spend the HOST's time?  (perf_counter around each call of the loop)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cell-image-analysis_amd"), ROOT):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "8")
import numpy as np
import torch
from cellscreen import synth
from cellscreen.augment import ImageDataGenerator
from cellscreen.trainer import Trainer
dev = torch.device("cuda", 0)
X = torch.from_numpy(synth.blob_crops(42, 40000)).to(dev)
tr = Trainer(synth.random_cae(seed=42, trivial_bn=True), device_id=0)
gen = ImageDataGenerator(rotation_range=2, width_shift_range=0.02, height_shift_range=0.02, zoom_range=0.02, horizontal_flip=True, vertical_flip=True, fill_mode="nearest")
rng = np.random.default_rng(1)
tg = torch.Generator(device=dev); tg.manual_seed(1)
for mode in ("full", "full", "no_aug", "no_gather"):
    steps = 1250
    perm = torch.randperm(40000, device=dev, generator=tg)[:steps * 32].view(steps, 32)
    yb0 = X[perm[0]].contiguous()
    acc = dict(gather=0.0, draw=0.0, augment=0.0, step=0.0)
    tr.reset_metrics(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        a = time.perf_counter()
        yb = yb0 if mode == "no_gather" else X[perm[i]].contiguous()
        b = time.perf_counter()
        tf = gen.random_transforms(32, (64, 64), rng)
        c = time.perf_counter()
        xb = yb if mode == "no_aug" else tr.augment(yb, tf)
        d = time.perf_counter()
        tr.step_async(xb, yb, 1e-3)
        e = time.perf_counter()
        acc["gather"] += b - a; acc["draw"] += c - b; acc["augment"] += d - c; acc["step"] += e - d
    t_enq = time.perf_counter() - t0
    tr.read_metrics(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(json.dumps(dict(mode=mode, hwq=os.environ["GPU_MAX_HW_QUEUES"], ms_per_step=round(t_all / steps * 1e3, 4), enqueue_ms_per_step=round(t_enq / steps * 1e3, 4),
                          host_us={k: round(v / steps * 1e6, 1) for k, v in acc.items()})), flush=True)
tr.close()
