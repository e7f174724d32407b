import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gandiva_amd as gandiva
from gandiva_amd import workloads as W
n = 100_000_000
proj = gandiva.make_projector(W.c5_schema(), W.c5_expressions(), None)
db = W.c5_device_batch(n, non_ascii_fraction=0.01)
outs = proj.evaluate_device(db)
for _ in range(5):
    outs = proj.evaluate_device(db, outputs=outs)
torch.cuda.synchronize()
