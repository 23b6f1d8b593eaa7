"""Development: fit_step steps of the 128x128 / 32-64-128 variant for a rocprofv3 --kernel-trace run."""
import sys, time
sys.path.insert(0, "cell-image-analysis_amd")
import numpy as np, torch
from cellscreen import synth
from cellscreen.trainer import Trainer
hw, ch = (128, 128), (32, 64, 128, 128, 64, 32, 1)
X = torch.from_numpy(synth.blob_crops(1, 512, hw=hw)).cuda()
tr = Trainer(synth.random_cae(seed=1, hw=hw, channels=ch, trivial_bn=True))
rng = np.random.default_rng(0)
idx = rng.integers(0, 512, (200, 32)).astype(np.int32)
for i in range(20): tr.fit_step(X, idx[i], None, seed=1, step=i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20, 120): tr.fit_step(X, idx[i], None, seed=1, step=i)
tr.read_metrics()
print("ms per step", (time.perf_counter() - t0) / 100 * 1e3)
tr.close()
