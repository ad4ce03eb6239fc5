"""N>1 rehearsal on ONE GPU: two bench.py ranks share cuda:0 with the gloo backend (the KB-sized
collectives hop through the host).  Exercises exactly the control flow the driver's RCCL runs use:
gallery sharding with global index bases, query all-gather, per-shard exact search, top-k all-gather
and merge.  A dedicated check compares the 2-rank result with a single scan of the whole gallery."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHECK = r'''
import os, sys
sys.path.insert(0, "{root}"); sys.path.insert(0, os.path.join("{root}", "hair-centric-image-retrieval_amd"))
import numpy as np, torch, torch.distributed as dist
import torch.nn.functional as F
from hcir.dist import ShardedGallery, shard_bounds
from hcir.gallery import ResidentGallery
from oracle import knn as oknn
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
g = F.normalize(torch.randn(40001, 128, generator=torch.Generator().manual_seed(0)), dim=1)
g[30000] = g[5]                                    # tie across shards
q = F.normalize(torch.randn(24, 128, generator=torch.Generator().manual_seed(1)), dim=1)
q[0] = g[5]
lo, hi = shard_bounds(40001, world, rank)
shard = g[lo:hi].cuda()
gal = ShardedGallery(shard, lo, resident=ResidentGallery(shard, lo))
per = 24 // world
q_all = gal.gather_queries(q[rank * per:(rank + 1) * per].cuda())
val, idx = gal.search(q_all, 10)
rv, ri = oknn.cosine_topk(q.numpy(), g.numpy(), 10)
assert np.array_equal(idx.cpu().numpy(), ri), "indices differ from the single-scan oracle"
assert np.array_equal(val.cpu().numpy(), rv), "values differ"
assert list(ri[0, :2]) == [5, 30000]
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _run(args, env_extra, timeout=600, nproc=2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                           "--master-addr", "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300)] + args,
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("nproc", [2, 4])   # 4 ranks + this process stay inside the box's 6-process GPU guard
def test_multi_rank_sharded_search_equals_single_scan(tmp_path, nproc):
    script = tmp_path / "check.py"
    script.write_text(CHECK.format(root=ROOT))
    r = _run([str(script)], {}, nproc=nproc)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == nproc


def test_bench_two_ranks_rehearsal():
    r = _run(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "32", "--gallery", "100000"],
             {"HCIR_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_query_batch"] == 64
    assert d["value"] > 0 and d["roofline"]["bound"] == "mfma" and "cpu_baseline" not in d
