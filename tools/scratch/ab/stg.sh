# experiment: stagger of waves 4-7 per phase (P1,P2,P3,P4 start), conv12 ms per 1M
for s in 0,0,0,0 2,0,0,0 4,0,0,0 8,0,0,0 0,2,0,0 0,4,0,0 0,8,0,0 0,0,2,0 0,0,4,0 0,0,8,0 0,0,0,2 0,0,0,4 0,0,0,8 4,4,4,4 8,8,8,8; do
  CS_X_STG=$s python - <<PY
import sys,os
sys.path.insert(0,"cell-image-analysis_amd")
import torch
from cellscreen import synth
from cellscreen.engine import Engine
N=65536
e=Engine.from_weights(synth.random_cae(42))
x=torch.empty((N,64,64),dtype=torch.float32,device="cuda"); e.synth_crops(42,0,x); e.set_chunk(N)
f=e.layer_output(x,1)
e.profile_enable(True); e.profile_reset()
for _ in range(10): f=e.layer_output(x,1)
p=e.profile()["conv1_conv2_fused"]
print(os.environ["CS_X_STG"], "conv12 ms per 1M cells: %.2f" % (p["ms"]/p["launches"]/N*1e6))
PY
done
