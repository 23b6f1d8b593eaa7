"""Development: 300 fit_step steps for a rocprofv3 --kernel-trace run (gap / critical-path analysis of the training step)."""
import sys, time
sys.path.insert(0, "cell-image-analysis_amd")
import numpy as np, torch
from cellscreen import synth
from cellscreen.augment import ImageDataGenerator
from cellscreen.trainer import Trainer
X = torch.from_numpy(synth.blob_crops(1, 4096)).cuda()
tr = Trainer(synth.random_cae(seed=1, trivial_bn=True))
cfg = ImageDataGenerator.reference().config()
rng = np.random.default_rng(0)
idx = rng.integers(0, 4096, (400, 32)).astype(np.int32)
for i in range(50): tr.fit_step(X, idx[i], cfg, seed=1, step=i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50, 350): tr.fit_step(X, idx[i], cfg, seed=1, step=i)
tr.read_metrics()
print("ms per step", (time.perf_counter() - t0) / 300 * 1e3)
tr.close()
