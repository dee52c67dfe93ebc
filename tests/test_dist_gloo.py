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
    # the 8-rank launch the driver uses on an 8-GPU node (VERDICT r3 next #8f): same self-launcher, gloo, one all-reduce + barrier
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 8 and res["rank_sum"] == 36.0
    # a launcher that starts a different number of ranks than --gpus asks for is an error, not a warning
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


# ---- SURVEY 8(e) with the REAL model: 2 ranks x 1 sample == the mean of the per-sample gradients of one process -------------
def _mvsnet_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import mvs_amd  # noqa: F401
    from mvs_amd import dist as mdist
    from mvs_amd.jdacs.models.mvsnet import MVSNet, mvsnet_loss
    from mvs_amd.synthetic import synthetic_mvsnet_inputs
    mdist.init_from_env("gloo")            # both ranks share the box's ONE GPU: the collective goes through host memory
    dev = torch.device("cuda:0")
    torch.manual_seed(100 + rank)          # deliberately different init per rank -> the broadcast must fix it
    net = MVSNet(refine=False)
    with torch.no_grad():
        net.cost_regularization.prob.weight.mul_(50.0)
    net = net.to(dev).train()
    mdist.broadcast_parameters(net)
    bucket = mdist.FlatGradBucket(net.parameters(), flatten_params=True)
    imgs, proj, dv = synthetic_mvsnet_inputs(1, 3, 64, 96, 16, seed=1 + rank)     # one sample per rank
    gt = torch.full((1, 16, 24), 450.0, device=dev)
    bucket.zero()
    out = net(imgs.to(dev), proj.to(dev), dv.to(dev))
    mvsnet_loss(out["depth"], gt, torch.ones_like(gt)).backward()
    bucket.gather()
    bucket.all_reduce()
    torch.cuda.synchronize()
    if rank == 0:
        ret["flat"] = bucket.flat.cpu()
        ret["sd"] = {k: v.cpu() for k, v in net.state_dict().items() if "running" not in k and "num_batches" not in k}
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs cuda:0 (the workers run the real MVSNet on the HIP path)")
def test_two_ranks_real_mvsnet_equals_mean_of_per_sample_gradients():
    """One rank per sample with per-replica BatchNorm statistics and ONE flat all-reduce (the data-parallel scheme of dist.py,
    here 2 ranks sharing the box's single GPU over gloo) == one process running the two samples one after the other and
    averaging the gradients -- with the real MVSNet on the HIP path."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_mvsnet_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import mvs_amd  # noqa: F401
    from mvs_amd.jdacs.models.mvsnet import MVSNet, mvsnet_loss
    from mvs_amd.synthetic import synthetic_mvsnet_inputs
    dev = torch.device("cuda:0")
    net = MVSNet(refine=False)
    net.load_state_dict(dict(ret["sd"]), strict=False)
    net = net.to(dev).train()
    gt = torch.full((1, 16, 24), 450.0, device=dev)
    grads = None
    for r in range(world):
        imgs, proj, dv = synthetic_mvsnet_inputs(1, 3, 64, 96, 16, seed=1 + r)
        net.zero_grad(set_to_none=True)
        for m in net.modules():                       # per-replica BatchNorm: every sample starts from fresh running stats
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.reset_running_stats()
        out = net(imgs.to(dev), proj.to(dev), dv.to(dev))
        mvsnet_loss(out["depth"], gt, torch.ones_like(gt)).backward()
        # the bucket holds every gradient in its parameter's storage order (channels-last conv weights of the feature extractor)
        g = torch.cat([torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device).copy_(p.grad).as_strided((p.numel(),), (1,))
                       for p in net.parameters() if p.requires_grad])
        grads = g if grads is None else grads + g
    ref = (grads / world).cpu()
    got = ret["flat"]
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


def _forced_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    import mvs_amd  # noqa: F401
    from mvs_amd import dist as mdist
    r, w, _ = mdist.init_from_env("gloo", force=True)        # picks its own loopback port
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1
    p = torch.nn.Parameter(torch.arange(6.0))
    bucket = mdist.FlatGradBucket([p])
    (p * torch.arange(6.0)).sum().backward()
    bucket.gather()
    before = bucket.flat.clone()
    bucket.all_reduce()                                      # world size 1, not forced: no collective
    bucket.force_collective = True
    bucket.all_reduce()                                      # one all_reduce(sum) over the one rank: the values stay
    ret["same"] = bool(torch.equal(before, bucket.flat))
    dist.destroy_process_group()


def test_forced_collective_at_world_size_one():
    """bench.py --force-collective: the process group and the bucket's all-reduce run at world size 1 (on the GPU box: RCCL)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_forced_worker, args=(1, 0, ret), nprocs=1, join=True)
    assert ret["same"] is True
