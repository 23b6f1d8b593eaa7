import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cell-image-analysis_amd")); sys.path.insert(0, ROOT)
import torch
from cellscreen import synth
from cellscreen.engine import Engine
eng = Engine.from_weights(synth.random_cae(seed=42)); eng.set_chunk(65536)
n = 262144
x = torch.empty((n, 64, 64), dtype=torch.float32, device="cuda"); eng.synth_crops(1, 0, x)
host = torch.empty(2_000_000_000 // 2, dtype=torch.int16, pin_memory=True); host.zero_()
dst = torch.empty_like(host, device="cuda")
cs = torch.cuda.Stream(); ms = torch.cuda.Stream()
def copy():
    with torch.cuda.stream(cs): dst.copy_(host, non_blocking=True)
def comp():
    with torch.cuda.stream(ms): eng.encode(x)        # conv12 + conv3, blocks the host until done
for name, fn in (("copy 2 GB", lambda: copy()), ("compute", lambda: comp()), ("copy then compute (enqueued together)", lambda: (copy(), comp()))):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-45s %.4f s" % (name, dt), flush=True)
print("HSA_ENABLE_SDMA", os.environ.get("HSA_ENABLE_SDMA"), "HIP env", {k: v for k, v in os.environ.items() if k.startswith(("HIP_", "HSA_", "GPU_", "AMD_"))})
