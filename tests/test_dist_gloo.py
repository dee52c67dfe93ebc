"""CPU, world_size 2 over gloo: the flat gradient bucket all-reduce == gradients of the global batch."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import mvs_amd  # noqa: F401
    from mvs_amd import dist as mdist
    r, w, _ = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # deliberately different init per rank -> broadcast must fix it
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                                torch.nn.Conv2d(4, 1, 3, padding=1))
    mdist.broadcast_parameters(model)
    bucket = mdist.FlatGradBucket(model.parameters())
    g = torch.Generator().manual_seed(7)
    data = torch.randn(world, 2, 3, 8, 8, generator=g)  # one "sample" (mini batch) per rank
    bucket.zero()
    model(data[rank]).mean().backward()
    bucket.gather()
    bucket.all_reduce()
    if rank == 0:
        ret["flat"] = bucket.flat.clone()
        ret["nbytes"] = bucket.nbytes
        ret["sd"] = {k: v.clone() for k, v in model.state_dict().items()}
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_matches_global_batch():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    # single-process reference: mean over ranks of per-rank (per-replica BN) losses
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                                torch.nn.Conv2d(4, 1, 3, padding=1))
    sd = dict(ret["sd"])
    for k in list(sd):
        if "running" in k or "num_batches" in k:
            sd.pop(k)
    model.load_state_dict(sd, strict=False)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_running_stats()
    g = torch.Generator().manual_seed(7)
    data = torch.randn(world, 2, 3, 8, 8, generator=g)
    loss = sum(model(data[r]).mean() for r in range(world)) / world
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert ret["nbytes"] == flat.numel() * 4
    assert torch.allclose(ret["flat"], flat, atol=1e-6, rtol=1e-5)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without torchrun must start N ranks by itself (VERDICT r1 missing #4)."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["rank_sum"] == 3.0     # ranks 0 and 1 both took part in the collective
    # a launcher that starts a different number of ranks than --gpus asks for is an error, not a warning
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
