import os, sys
sys.path.insert(0, "/root/repo/tools")
import torch
from bench_sim import run
run(64, 1_000_000, 768, 10, torch.float16, iters=20)
run(64, 1_000_000, 768, 10, torch.float32, iters=20)
