"""Head-training throughput: optimisation steps per second on the GPU (cbas_head_train_step, batch 512 as
in train_lstm_model's default).  The comparison with the CPU restatement of the same step lives in the test-suite
(tests/test_gpu_train.py::test_step_rate_next_to_the_cpu_oracle): nothing outside tests/, smoke() and bench.py's CPU leg
touches oracle/."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, synth  # noqa: E402
from cbas_amd.train import HeadTrainer, initial_head_weights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
hcfg = C.HeadConfig(in_features=768, out_features=9)
w0 = initial_head_weights(hcfg, 0)
x, y = synth.train_windows(1, B, 768, 9, 31)
xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
tr = HeadTrainer(hcfg, w0, "cuda", lr=1e-4, max_batch=B, seed=1)
for _ in range(3):
    tr.step(xt, yt, want_loss=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(xt, yt, want_loss=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
loss = tr.step(xt, yt)[0]
print(f"GPU: batch {B}: {dt * 1e3:.3f} ms/step, {B / dt:.0f} windows/s (loss after {steps + 4} steps {loss:.4f})", flush=True)
tr.close()
