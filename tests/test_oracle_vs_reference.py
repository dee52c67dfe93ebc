"""Build container only: the oracle against the reference's own Python imported from /root/reference
(skipped where the reference is absent, e.g. on the GPU box).  Wider cases than the committed goldens:
more views, align_corners=True (the authors' torch-1.1 behaviour, App. A Q1), 3-level CVP."""
import os
import subprocess
import sys
import textwrap

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "jdacs")), reason="reference not mounted")


def _run(code):
    """Each case in a fresh interpreter: jdacs and jdacs-ms both own a top-level `models` package."""
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


PRE = """
import sys, types, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, %r)
import torch, torch.nn.functional as F
torch.set_num_threads(4)
from oracle import ref_torch as R
""" % ROOT


def test_mvsnet_five_views_train_and_eval():
    _run(PRE + """
sys.path.insert(0, "/root/reference/jdacs")
from models.mvsnet import MVSNet
torch.manual_seed(0)
ref = MVSNet(refine=True)
with torch.no_grad():
    ref.cost_regularization.prob.weight.mul_(50.0)
ora = R.OracleMVSNet(refine=True)
ora.load_state_dict(ref.state_dict())
imgs, proj, dv = R.synthetic_mvsnet_inputs(2, 5, 64, 96, 16, seed=3)
for mode in ("train", "eval"):
    getattr(ref, mode)(); getattr(ora, mode)()
    a = ref(imgs, proj, dv); b = ora(imgs, proj, dv)
    assert torch.allclose(a["depth"], b["depth"], rtol=1e-5, atol=1e-3), mode
    assert torch.allclose(a["photometric_confidence"], b["photometric_confidence"], atol=1e-4), mode
    if mode == "train":
        a["depth"].mean().backward(); b["depth"].mean().backward()
        for (k, p), (_, q) in zip(ref.named_parameters(), ora.named_parameters()):
            if p.grad is None or k.endswith("prob.bias"): continue
            assert float((p.grad - q.grad).abs().mean()) <= 2e-3 * float(p.grad.abs().mean()) + 1e-9, k
""")


def test_homo_warping_align_corners_true():
    _run(PRE + """
sys.path.insert(0, "/root/reference/jdacs")
from models import module as refmod
_gs = F.grid_sample
refmod.F.grid_sample = lambda inp, grid, **kw: _gs(inp, grid, align_corners=True, **kw)  # torch-1.1 semantics
imgs, proj, dv = R.synthetic_mvsnet_inputs(2, 3, 48, 64, 8, seed=5)
src = torch.randn(2, 8, 12, 16)
sp, rp = proj[:, 1].contiguous(), proj[:, 0].contiguous()
a = refmod.homo_warping(src, sp, rp, dv)
b = R.homo_warping(src, sp, rp, dv, align_corners=True)
assert float((a - b).abs().max()) < 2e-5
""")


def test_cvpmvsnet_three_levels():
    _run(PRE + """
sys.path.insert(0, "/root/reference/jdacs-ms")
torch.Tensor.cuda = lambda self, *a, **k: self
from models.network import CVPMVSNet
torch.manual_seed(0)
args = types.SimpleNamespace(nsrc=3, nscale=3, mode="train")
ref = CVPMVSNet(args); ora = R.OracleCVPMVSNet(args)
ora.load_state_dict(ref.state_dict())
ref.train(); ora.train()
g = torch.Generator().manual_seed(2)
ih, iw = 96, 128
K, E = R.synthetic_cameras(4, ih, iw, iw)
ins = [torch.randn(1, 3, ih, iw, generator=g), torch.randn(1, 3, 3, ih, iw, generator=g), K.unsqueeze(0),
       K.view(1, 1, 3, 3).repeat(1, 3, 1, 1), E[0].unsqueeze(0), E[1:].unsqueeze(0), torch.tensor([425.0]),
       torch.tensor([425.0 + 47 * 13.5])]
a = ref(*ins); b = ora(*ins)
for x, y in zip(a["depth_est_list"], b["depth_est_list"]):
    assert float((x - y).abs().mean() / y.abs().mean()) < 1e-5   # fp32 roundoff chained through 3 levels of batch-stat BN
    assert torch.allclose(x, y, rtol=1e-4, atol=5e-2)
assert torch.allclose(a["prob_confidence"], b["prob_confidence"], atol=1e-4)
sum(d.mean() for d in a["depth_est_list"]).backward(); sum(d.mean() for d in b["depth_est_list"]).backward()
for (k, p), (_, q) in zip(ref.named_parameters(), ora.named_parameters()):
    if p.grad is None or k.endswith("prob0.bias"): continue
    assert float((p.grad - q.grad).abs().mean()) <= 3e-2 * float(p.grad.abs().mean()) + 1e-9, k  # roundoff grows 4e-7 -> 9e-3 from last to first layer (3 levels of batch-stat BN)
""")


def test_unsup_loss_seven_views_other_lambda_and_pfm():
    """SURVEY 8(f)-1 / A12 beyond the committed fixtures: N = 7, odd quarter-resolution size, batch 3; the reference's
    save_pfm / read_pfm against the drop-in's on a random map."""
    _run(PRE + """
sys.argv = ["x", "--smooth_lambda", "0.5"]
sys.path.insert(0, "/root/reference/jdacs")
from losses.unsup_loss import UnSupLoss
g = torch.Generator().manual_seed(4)
b, n, h, w = 3, 7, 52, 76
imgs = F.avg_pool2d(torch.randn(b * n, 3, h, w, generator=g), 5, 1, 2).view(b, n, 3, h, w) * 3
K, E = R.synthetic_cameras(n, h // 4, w // 4, w)
cams = torch.zeros(b, n, 2, 4, 4); cams[:, :, 0] = E; cams[:, :, 1, :3, :3] = K
depth = 600.0 + 40.0 * torch.rand(b, h // 4, w // 4, generator=g)
da, db = depth.clone().requires_grad_(True), depth.clone().requires_grad_(True)
la = UnSupLoss()(imgs, cams, da)
lb = R.unsup_loss(imgs, cams, db, smooth_lambda=0.5)
la.backward(); lb.backward()
assert abs(float(la) - float(lb)) < 2e-5 * abs(float(la)), (float(la), float(lb))
assert float((da.grad - db.grad).abs().max()) < 2e-6 + 1e-4 * float(da.grad.abs().max())
import numpy as np, tempfile, os
import mvs_amd
from mvs_amd.jdacs.datasets import data_io as mine
import importlib.util
spec = importlib.util.spec_from_file_location("ref_data_io", "/root/reference/jdacs/datasets/data_io.py")
ref_io = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_io)
arr = np.random.RandomState(1).rand(17, 23).astype(np.float32) * 900
with tempfile.TemporaryDirectory() as d:
    ref_io.save_pfm(os.path.join(d, "a.pfm"), arr); mine.save_pfm(os.path.join(d, "b.pfm"), arr)
    assert open(os.path.join(d, "a.pfm"), "rb").read() == open(os.path.join(d, "b.pfm"), "rb").read()
    assert np.array_equal(mine.read_pfm(os.path.join(d, "a.pfm"))[0], ref_io.read_pfm(os.path.join(d, "b.pfm"))[0])
""")
