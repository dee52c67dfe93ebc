#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (build container only).

    python tests/golden/make_goldens.py

Imports /root/reference/jdacs and /root/reference/jdacs-ms exactly as SURVEY.md Appendix D
describes (``.cuda()`` shim for jdacs-ms, clean argv), runs the reference's own functions on seeded
inputs and stores inputs + outputs (+ gradients) as small fixtures.  Nothing of the reference's
source is stored -- only tensors.  Weights are rounded to fp16-representable values before the
reference runs so they can be stored as float16 without loss.

The fixtures pin torch-2.10 semantics of F.grid_sample (align_corners=False, App. A Q1).
"""
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.ref_torch import synthetic_cameras  # shared synthetic camera definition (inputs only)

REF = "/root/reference"
torch.set_num_threads(4)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def fp16_exact_(module):
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(p.half().float())


def sd_to_np16(sd):
    out = {}
    for k, v in sd.items():
        if v.dtype == torch.float32:
            assert torch.equal(v.half().float(), v) or "running" in k, k
            out["sd." + k] = v.numpy() if "running" in k else v.half().numpy()
        else:
            out["sd." + k] = v.numpy()
    return out


def projections(nviews, fh, fw, img_w, batch=1, rot_scale=1.0):
    K, E = synthetic_cameras(nviews, fh, fw, img_w)
    P = E.clone()
    P[:, :3, :4] = torch.matmul(K, E[:, :3, :4])
    return K, E, P.unsqueeze(0).repeat(batch, 1, 1, 1)


# =============================================================================================
# jdacs (MVSNet backbone)
# =============================================================================================
sys.path.insert(0, os.path.join(REF, "jdacs"))
from models.mvsnet import MVSNet, CostRegNet  # noqa: E402
from models.module import homo_warping, depth_regression  # noqa: E402

# ---- G1: homo_warping fwd + grad wrt src_fea --------------------------------------------------
g = torch.Generator().manual_seed(11)
B, C, D, H, W = 2, 8, 6, 16, 20
_, _, P = projections(3, H, W, 4 * W, batch=B)
P[1, 1, :3, 3] *= 1.7  # make batch item 1 differ
depth = (425.0 + 37.5 * torch.arange(D, dtype=torch.float32)).unsqueeze(0).repeat(B, 1)
depth[1] += 11.0
for tag, sv in (("a", 1), ("b", 2)):
    src = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    # contiguous copies: torch.inverse of a strided batch view rounds differently (5.96e-8 in P_ref^-1),
    # which the ill-conditioned homography amplifies to ~7e-5 in the samples of white-noise features.
    sp, rp = P[:, sv].contiguous(), P[:, 0].contiguous()
    out = homo_warping(src, sp, rp, depth)
    gup = torch.randn(out.shape, generator=g)
    out.backward(gup)
    save("g1_homo_warping_" + tag, src_fea=src, src_proj=sp, ref_proj=rp, depth_values=depth,
         out=out, grad_out=gup, grad_src=src.grad)

# ---- G4a: CostRegNet (MVSNet) train / calibrated-eval ------------------------------------------
torch.manual_seed(0)
reg = CostRegNet()
fp16_exact_(reg)
with torch.no_grad():
    reg.prob.weight.mul_(40.0)  # peaky softmax downstream; still fp16-exact? re-round:
fp16_exact_(reg)
x = (torch.randn(1, 32, 8, 16, 16, generator=g).abs() * 0.3).requires_grad_(True)
sd0 = {k: v.clone() for k, v in reg.state_dict().items()}
reg.train()
y = reg(x)
gup = torch.randn(y.shape, generator=g)
y.backward(gup)
grads = {"grad." + k: p.grad for k, p in reg.named_parameters()}
sd1 = {k: v.clone() for k, v in reg.state_dict().items()}
# second calibration pass (different input), then eval
x2 = torch.randn(1, 32, 8, 16, 16, generator=g).abs() * 0.3
with torch.no_grad():
    reg(x2)
sd2 = {k: v.clone() for k, v in reg.state_dict().items()}
reg.eval()
with torch.no_grad():
    y_eval = reg(x.detach())
bufs1 = {"after1." + k: v for k, v in sd1.items() if "running" in k or "num_batches" in k}
bufs2 = {"after2." + k: v for k, v in sd2.items() if "running" in k or "num_batches" in k}
save("g4_costregnet_mvs", x=x, x2=x2, y_train=y, grad_out=gup, grad_x=x.grad, y_eval=y_eval,
     **sd_to_np16(sd0), **grads, **bufs1, **bufs2)

# ---- G6: MVSNet end to end (small config-1-like), with intermediates via hooks -----------------
torch.manual_seed(0)
net = MVSNet(refine=False)
fp16_exact_(net)
with torch.no_grad():
    net.cost_regularization.prob.weight.mul_(60.0)
fp16_exact_(net)
Bm, N, IH, IW, Dm = 1, 3, 64, 96, 16
imgs = torch.randn(Bm, N, 3, IH, IW, generator=g)
_, _, Pm = projections(N, IH // 4, IW // 4, IW, batch=Bm)
dv = (425.0 + 10.6 * torch.arange(Dm, dtype=torch.float32)).unsqueeze(0)
cap = {}
hk = net.cost_regularization.register_forward_hook(
    lambda m, i, o: cap.update(variance=i[0].detach().clone(), logits=o.detach().clone()))
sd0 = {k: v.clone() for k, v in net.state_dict().items()}
net.train()
out_t = net(imgs, Pm, dv)
cap_t = dict(cap)
loss = (out_t["depth"] * torch.linspace(0.5, 1.5, out_t["depth"].numel()).view_as(out_t["depth"])).mean()
loss.backward()
grads = {"grad." + k: p.grad for k, p in net.named_parameters()}
# one more calibration pass, then eval (non-trivial running stats -> non-degenerate eval output)
with torch.no_grad():
    net(torch.randn(Bm, N, 3, IH, IW, generator=g), Pm, dv)
sd_cal = {k: v.clone() for k, v in net.state_dict().items()}
net.eval()
with torch.no_grad():
    out_e = net(imgs, Pm, dv)
cap_e = dict(cap)
hk.remove()
cal = {"cal." + k: v for k, v in sd_cal.items() if "running" in k}
save("g6_mvsnet_e2e", imgs=imgs, proj=Pm, depth_values=dv,
     train_depth=out_t["depth"], train_conf=out_t["photometric_confidence"],
     train_variance=cap_t["variance"], train_logits=cap_t["logits"],
     eval_depth=out_e["depth"], eval_conf=out_e["photometric_confidence"],
     eval_variance=cap_e["variance"], eval_logits=cap_e["logits"],
     **sd_to_np16(sd0), **grads, **cal)

# ---- G5: softmax / regression / confidence on peaky logits (through the reference forward) -----
# The reference has no standalone function for this stage; run its forward with the regulariser
# replaced by a module that returns prescribed logits, and record depth + confidence.
class _FixedLogits(torch.nn.Module):
    def __init__(self, logits):
        super().__init__()
        self.logits = logits

    def forward(self, x):
        return self.logits.unsqueeze(1)


Ds, Hs, Ws = 24, 8, 12
logits = torch.randn(2, Ds, Hs, Ws, generator=g) * 4.0
logits[0, :, 0, 0] = 0.0          # flat column -> idx = trunc((D-1)/2)
logits[0, 0, 0, 1] = 30.0         # peak at d=0 (window clipped on the left)
logits[0, Ds - 1, 0, 2] = 30.0    # peak at d=D-1 (window clipped on the right)
logits[1, 5, 1, 1] = 25.0
logits[1, 6, 1, 1] = 25.0         # expected index exactly between -> truncation edge
logits.requires_grad_(True)
net5 = MVSNet(refine=False)
net5.cost_regularization = _FixedLogits(logits)
net5.eval()
imgs5 = torch.randn(2, 2, 3, Hs * 4, Ws * 4, generator=g)
_, _, P5 = projections(2, Hs, Ws, Ws * 4, batch=2)
dv5 = (425.0 + 2.65 * torch.arange(Ds, dtype=torch.float32)).unsqueeze(0).repeat(2, 1)
dv5[1] = 600.0 + 7.5 * torch.arange(Ds, dtype=torch.float32)
o5 = net5(imgs5, P5, dv5)
gd = torch.randn(o5["depth"].shape, generator=g)
o5["depth"].backward(gd)
save("g5_softargmin", logits=logits, depth_values=dv5, depth=o5["depth"], conf=o5["photometric_confidence"],
     grad_depth=gd, grad_logits=logits.grad)

# =============================================================================================
# jdacs-ms (CVP-MVSNet backbone)
# =============================================================================================
for m in [k for k in list(sys.modules) if k.split(".")[0] in ("models", "losses", "utils", "config", "datasets")]:
    del sys.modules[m]
sys.path[0] = os.path.join(REF, "jdacs-ms")
torch.Tensor.cuda = lambda self, *a, **k: self  # modules.py:59,73,130,223 hard-code .cuda()
from models.network import CVPMVSNet, CostRegNet as CostRegNetMS  # noqa: E402
from models import modules as msmod  # noqa: E402

# ---- G3: proj_cost (per-pixel hypotheses, alias quirk) + ms homo_warping -----------------------
B, C, D, H, W = 1, 16, 8, 12, 16
nsrc = 2
K, E = synthetic_cameras(nsrc + 1, H, W, 4 * W)
ref_in = K.unsqueeze(0)
src_in = K.unsqueeze(0).unsqueeze(0).repeat(1, nsrc, 1, 1).clone()
src_in[:, 1, 0, 0] *= 1.03
ref_ex = E[0].unsqueeze(0)
src_ex = E[1:].unsqueeze(0)
ref_f = torch.randn(B, C, H, W, generator=g, requires_grad=True)
src_f = [torch.randn(B, C, H, W, generator=g, requires_grad=True) for _ in range(nsrc)]
hyp = 500.0 + 40.0 * torch.rand(B, 1, H, W, generator=g) + 6.0 * torch.arange(D).view(1, D, 1, 1).float()
settings = types.SimpleNamespace(nsrc=nsrc, mode="train")
cost = msmod.proj_cost(settings, ref_f, [[f] for f in src_f], 0, ref_in, src_in, ref_ex, src_ex, hyp)
gup = torch.randn(cost.shape, generator=g)
cost.backward(gup)
planes = (450.0 + 25.0 * torch.arange(D, dtype=torch.float32)).unsqueeze(0)
with torch.no_grad():
    warped_ms = msmod.homo_warping(src_f[0].detach(), ref_in, src_in[:, 0], ref_ex, src_ex[:, 0], planes)
save("g3_proj_cost", ref_fea=ref_f, src_fea0=src_f[0], src_fea1=src_f[1], ref_in=ref_in, src_in=src_in,
     ref_ex=ref_ex, src_ex=src_ex, hypos=hyp, cost=cost, grad_out=gup, grad_ref=ref_f.grad,
     grad_src0=src_f[0].grad, grad_src1=src_f[1].grad, planes=planes, warped_ms=warped_ms)

# ---- G4b: CostRegNet (CVP) ----------------------------------------------------------------------
torch.manual_seed(0)
regms = CostRegNetMS()
fp16_exact_(regms)
x = (torch.randn(1, 16, 8, 12, 16, generator=g).abs() * 0.3).requires_grad_(True)
sd0 = {k: v.clone() for k, v in regms.state_dict().items()}
regms.train()
y = regms(x)
gup = torch.randn(y.shape, generator=g)
y.backward(gup)
grads = {"grad." + k: p.grad for k, p in regms.named_parameters()}
sd1 = {k: v.clone() for k, v in regms.state_dict().items()}
regms.eval()
with torch.no_grad():
    y_eval = regms(x.detach())
bufs1 = {"after1." + k: v for k, v in sd1.items() if "running" in k or "num_batches" in k}
save("g4_costregnet_cvp", x=x, y_train=y, grad_out=gup, grad_x=x.grad, y_eval=y_eval,
     **sd_to_np16(sd0), **grads, **bufs1)

# ---- G7: CVPMVSNet 2-level end to end + calDepthHypo --------------------------------------------
torch.manual_seed(0)
args = types.SimpleNamespace(nsrc=2, nscale=2, mode="test")
cvp = CVPMVSNet(args)
fp16_exact_(cvp)
IH, IW = 64, 96
ref_img = torch.randn(1, 3, IH, IW, generator=g)
src_imgs = torch.randn(1, 2, 3, IH, IW, generator=g)
K, E = synthetic_cameras(3, IH, IW, IW)  # image-resolution intrinsics
ref_in = K.unsqueeze(0)
src_in = K.unsqueeze(0).unsqueeze(0).repeat(1, 2, 1, 1).clone()
ref_ex = E[0].unsqueeze(0)
src_ex = E[1:].unsqueeze(0)
dmin = torch.tensor([425.0])
dmax = torch.tensor([425.0 + 47 * 13.5])  # exactly representable step -> torch.range gives 48 planes (Q3)
cvp.train()  # batch-stat BN => non-degenerate outputs; args.mode stays "test" (only affects del/empty_cache)
cvp.args.mode = "train"
with torch.no_grad():
    o7 = cvp(ref_img, src_imgs, ref_in, src_in, ref_ex, src_ex, dmin, dmax)
    up = F.interpolate(o7["depth_est_list"][1][None, :], scale_factor=2, mode="bilinear").squeeze(0)
    hyp7 = msmod.calDepthHypo(args, up, ref_in, src_in, ref_ex, src_ex, dmin, dmax, 0)
sd7 = {k: v.clone() for k, v in cvp.state_dict().items() if "running" not in k and "num_batches" not in k}
save("g7_cvpmvsnet_e2e", ref_img=ref_img, src_imgs=src_imgs, ref_in=ref_in, src_in=src_in, ref_ex=ref_ex,
     src_ex=src_ex, depth_min=dmin, depth_max=dmax, depth0=o7["depth_est_list"][0],
     depth1=o7["depth_est_list"][1], conf=o7["prob_confidence"], depth_up=up, hypos0=hyp7, **sd_to_np16(sd7))
print("done")
