"""Kernel LOGIC checks without a GPU: the product's kernel sources compiled for the host
(tests/cpu_emul) and driven through the product's Python wrappers, compared with the oracle.
(The real parity tests are the -m gpu ones; these catch index-map / fragment-layout / reduction
bugs in the build container.)"""
import os

import numpy as np

import pytest
import torch
import torch.nn.functional as F

from conftest import (assert_as_accurate_as_fp32_reference, assert_grads_as_accurate_as_fp32_reference, load_golden, rel_l1,
                      state_dict_from)
from emul_util import emul_lib  # noqa: F401
from oracle import ref_torch as R

torch.set_num_threads(4)


def _cams(b, ns, h, w, gen):
    K, E = R.synthetic_cameras(ns + 1, h, w, 4 * w)
    P = E.clone()
    P[:, :3, :4] = K @ E[:, :3, :4]
    rots, transs = [], []
    for s in range(1, ns + 1):
        r, t = R.relative_projection(P[s:s + 1].repeat(b, 1, 1), P[0:1].repeat(b, 1, 1))
        rots.append(r)
        transs.append(t)
    return torch.stack(rots, 1), torch.stack(transs, 1)


@pytest.mark.parametrize("c,ns,per_pixel,alias,ac", [(8, 1, False, False, False), (16, 2, True, True, False),
                                                     (32, 2, False, False, True), (32, 4, False, False, False),
                                                     (32, 3, True, True, False), (16, 3, False, False, False),
                                                     (16, 6, False, False, False)])
def test_plane_sweep_variance(emul_lib, c, ns, per_pixel, alias, ac):
    from mvs_amd import ops
    g = torch.Generator().manual_seed(3)
    b, d, h, w = 2, 5, 12, 20
    rot, trans = _cams(b, ns, h, w, g)
    ref = torch.randn(b, c, h, w, generator=g, requires_grad=True)
    srcs = [torch.randn(b, c, h, w, generator=g, requires_grad=True) for _ in range(ns)]
    if per_pixel:
        depth = 450 + 30 * torch.rand(b, 1, h, w, generator=g) + 20.0 * torch.arange(d).view(1, d, 1, 1)
    else:
        depth = (430 + 35.0 * torch.arange(d)).unsqueeze(0).repeat(b, 1)
    var = ops.plane_sweep_variance(ref, srcs, rot, trans, depth, align_corners=ac, ms_alias=alias)
    assert var.is_contiguous(memory_format=torch.channels_last_3d)
    gup = torch.randn(var.shape, generator=g)
    var.backward(gup)
    got = [ref.grad.clone()] + [s.grad.clone() for s in srcs]
    for t in [ref] + srcs:
        t.grad = None
    exp = R.plane_sweep_variance(ref, srcs, [rot[:, i] for i in range(ns)], [trans[:, i] for i in range(ns)], depth,
                                 ms_alias=alias, align_corners=ac)
    exp.backward(gup)
    ref64 = ref.detach().double().requires_grad_(True)
    src64 = [s.detach().double().requires_grad_(True) for s in srcs]
    t64 = R.plane_sweep_variance(ref64, src64, [rot[:, i].double() for i in range(ns)],
                                 [trans[:, i].double() for i in range(ns)], depth.double(), ms_alias=alias, align_corners=ac)
    t64.backward(gup.double())
    assert_as_accurate_as_fp32_reference(var.detach(), exp.detach(), t64.detach(), what="variance volume")
    names = ["ref"] + ["src%d" % i for i in range(ns)]
    assert_grads_as_accurate_as_fp32_reference(
        dict(zip(names, got)), {n: t.grad for n, t in zip(names, [ref] + srcs)},
        {n: t.grad for n, t in zip(names, [ref64] + src64)}, what="plane-sweep feature gradients N=%d" % (ns + 1))


@pytest.mark.parametrize("c,ns,d,gd", [(8, 2, 67, 2), (16, 1, 66, 2), (8, 2, 65, 0), (8, 2, 66, -1)])
def test_plane_sweep_backward_long_segment(emul_lib, c, ns, d, gd):
    """One depth segment longer than 64 planes with a narrow depth range (the backward stages the per-plane hypotheses 64 planes
    at a time and takes the upstream gradient over in groups of 1 or 2 planes: odd / even tails, refill of the staging row)."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(5)
    b, h, w = 1, 6, 9
    rot, trans = _cams(b, ns, h, w, g)
    ref = torch.randn(b, c, h, w, generator=g, requires_grad=True)
    srcs = [torch.randn(b, c, h, w, generator=g, requires_grad=True) for _ in range(ns)]
    depth = (430 + 1.5 * torch.arange(d)).unsqueeze(0).repeat(b, 1)
    emul_lib.call("mvs_set_tuning", b"sweep_bwd", 0)                   # the round-2 per-wave-window kernel (kept behind the knob)
    emul_lib.call("mvs_set_tuning", b"bwd_gd", max(gd, 0))
    emul_lib.call("mvs_set_tuning", b"bwd_pf", 1 if gd < 0 else 0)   # gd = -1: the block-lookahead form
    emul_lib.call("mvs_set_tuning", b"bwd_dslab", d)
    try:
        var = ops.plane_sweep_variance(ref, srcs, rot, trans, depth)
        gup = torch.randn(var.shape, generator=g)
        var.backward(gup)
    finally:
        emul_lib.call("mvs_set_tuning", b"sweep_bwd", 0)
        emul_lib.call("mvs_set_tuning", b"bwd_gd", 2)
        emul_lib.call("mvs_set_tuning", b"bwd_pf", 0)
        emul_lib.call("mvs_set_tuning", b"bwd_dslab", 0)
    got = [ref.grad.clone()] + [s.grad.clone() for s in srcs]
    for t in [ref] + srcs:
        t.grad = None
    exp = R.plane_sweep_variance(ref, srcs, [rot[:, i] for i in range(ns)], [trans[:, i] for i in range(ns)], depth)
    exp.backward(gup)
    for a, t in zip(got, [ref] + srcs):
        assert float((a - t.grad).abs().max()) < 1e-3 * max(1.0, float(t.grad.abs().max()))



@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("c,ns,step,hw", [(32, 2, 400.0, (13, 21)), (16, 3, 150.0, (10, 18))])
def test_plane_sweep_backward_wide_depth_range(emul_lib, c, ns, step, hw, variant):
    """Footprints larger than an accumulation window: depth segmentation + global-atomic path (both backward kernels)."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(11)
    b, d = 1, 12
    h, w = hw
    rot, trans = _cams(b, ns, h, w, g)
    ref = torch.randn(b, c, h, w, generator=g, requires_grad=True)
    srcs = [torch.randn(b, c, h, w, generator=g, requires_grad=True) for _ in range(ns)]
    depth = (300 + step * torch.arange(d)).unsqueeze(0).repeat(b, 1)
    # variant 2 = the per-wave-window kernel with its windows switched off (every flush takes the global-atomic path),
    # variant 3 = ... in its 3-waves/SIMD form (one rotating register set for the upstream gradient), variant 4 = ... at ONE
    # wave/SIMD for 3-4 source views, variant 5 = ... with the block lookahead (1-2 source views)
    emul_lib.call("mvs_set_tuning", b"sweep_bwd", 1 if variant == 1 else 0)
    emul_lib.call("mvs_set_tuning", b"bwd_nowin", 1 if variant == 2 else 0)
    emul_lib.call("mvs_set_tuning", b"bwd_gd", 0 if variant == 3 else 2)
    emul_lib.call("mvs_set_tuning", b"bwd_pf", 2 if variant == 4 else (1 if variant == 5 else 0))
    try:
        var = ops.plane_sweep_variance(ref, srcs, rot, trans, depth)
        gup = torch.randn(var.shape, generator=g)
        var.backward(gup)
    finally:
        emul_lib.call("mvs_set_tuning", b"sweep_bwd", 0)
        emul_lib.call("mvs_set_tuning", b"bwd_nowin", 0)
        emul_lib.call("mvs_set_tuning", b"bwd_gd", 2)
        emul_lib.call("mvs_set_tuning", b"bwd_pf", 0)
    got = [ref.grad.clone()] + [s.grad.clone() for s in srcs]
    for t in [ref] + srcs:
        t.grad = None
    exp = R.plane_sweep_variance(ref, srcs, [rot[:, i] for i in range(ns)], [trans[:, i] for i in range(ns)], depth)
    exp.backward(gup)
    for a, t in zip(got, [ref] + srcs):
        assert float((a - t.grad).abs().max()) < 1e-3 * max(1.0, float(t.grad.abs().max()))


def test_homo_warping_golden(emul_lib):
    from mvs_amd.jdacs.models.module import homo_warping
    g = load_golden("g1_homo_warping_a")
    src = g["src_fea"].clone().requires_grad_(True)
    out = homo_warping(src, g["src_proj"], g["ref_proj"], g["depth_values"])
    with torch.no_grad():
        t64 = R.homo_warping(g["src_fea"].double(), g["src_proj"].double(), g["ref_proj"].double(),
                             g["depth_values"].double())
    # fp64 truth uses an fp64 inverse, so the golden's own error includes the fp32 inverse: compare loosely
    assert float((out - g["out"]).abs().max()) < 3e-4 and float((out.double() - t64).abs().max()) < 1e-3
    out.backward(g["grad_out"])
    assert float((src.grad - g["grad_src"]).abs().max()) < 1e-4


def test_softargmin_golden(emul_lib):
    from mvs_amd import ops
    g = load_golden("g5_softargmin")
    lg = g["logits"].clone().requires_grad_(True)
    depth, conf = ops.softargmin_conf(lg, g["depth_values"])
    assert float((depth - g["depth"]).abs().max()) < 1e-3
    assert float((conf - g["conf"]).abs().max()) < 1e-5
    depth.backward(g["grad_depth"])
    assert float((lg.grad - g["grad_logits"]).abs().max()) < 1e-5 * float(g["grad_logits"].abs().max()) + 1e-5
    # per-pixel hypotheses, small D (CVP refine levels) -> DS=1 kernel
    gen = torch.Generator().manual_seed(5)
    lg2 = torch.randn(1, 8, 6, 70, generator=gen) * 3
    hyp = 500 + torch.rand(1, 8, 6, 70, generator=gen) * 50
    d2, c2 = ops.softargmin_conf(lg2, hyp)
    e2, ec2, _ = R.softargmin_conf(lg2, hyp)
    assert float((d2 - e2).abs().max()) < 1e-3 and float((c2 - ec2).abs().max()) < 1e-5


CONV_CASES = [
    # cin, cout, stride, transposed, (d,h,w)
    (8, 16, 1, False, (4, 6, 20)),
    (16, 8, 1, False, (5, 4, 16)),
    (32, 8, 1, False, (4, 4, 18)),
    (8, 16, 2, False, (4, 8, 20)),
    (16, 32, 2, False, (6, 4, 8)),
    (16, 8, 2, True, (2, 4, 10)),
    (64, 32, 2, True, (2, 2, 4)),
    (64, 32, 1, True, (2, 3, 5)),
    (8, 1, 1, False, (4, 5, 18)),
    (16, 1, 1, False, (4, 5, 18)),      # CVP-MVSNet's probability layer: one output per thread
    (8, 1, 1, False, (5, 11, 37)),      # the four-outputs-per-thread form: ragged 4 x 8 x 32 tiles in all three directions
]


@pytest.fixture(params=[0, 2], ids=["full_tiles", "quarter_tiles"])
def conv_tiles(request, emul_lib):
    """The generic implicit-GEMM kernel's two workgroup tilings (knob "conv_small": the library picks by launch size)."""
    emul_lib.call("mvs_set_tuning", b"conv_small", request.param)
    yield request.param
    emul_lib.call("mvs_set_tuning", b"conv_small", 1)


@pytest.mark.parametrize("cin,cout,stride,transposed,dims", CONV_CASES)
def test_conv3d_family(emul_lib, conv_tiles, cin, cout, stride, transposed, dims):
    from mvs_amd import ops
    g = torch.Generator().manual_seed(cin * 7 + cout)
    b = 2 if max(dims) <= 16 and cin < 64 else 1      # (the emulated MFMA is a 64-thread barrier: keep the 64-channel cases small)
    x = torch.randn(b, cin, *dims, generator=g)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = torch.randn(wshape, generator=g) * 0.2
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    if transposed:
        yr = F.conv_transpose3d(xr, wr, stride=stride, padding=1, output_padding=stride - 1)
    else:
        yr = F.conv3d(xr, wr, stride=stride, padding=1)
    y, parts = ops.conv3d_forward(x, w, stride, transposed, want_stats=True)
    assert y.shape == yr.shape
    assert float((y - yr).abs().max()) < 2e-4
    s = parts.sum(0).float()
    assert torch.allclose(s[0], yr.detach().sum(dim=(0, 2, 3, 4)), atol=1e-2, rtol=1e-4)
    assert torch.allclose(s[1], (yr.detach() ** 2).sum(dim=(0, 2, 3, 4)), atol=1e-2, rtol=1e-4)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    gx = ops.conv3d_dgrad(gy, w, tuple(x.shape), stride, transposed)
    assert float((gx - xr.grad).abs().max()) < 3e-4
    gw = ops.conv3d_wgrad(x, gy, wshape, stride, transposed)
    assert float((gw - wr.grad).abs().max()) < 1e-3 * max(1.0, float(wr.grad.abs().max()))


def test_conv_epilogue_and_bn(emul_lib):
    from mvs_amd import ops
    from mvs_amd.nn3d import ConvBnReLU3D, DeconvBnReLU3D
    g = torch.Generator().manual_seed(9)
    for mod_t, ref_mod, xs in ((ConvBnReLU3D(8, 16, stride=2), None, (1, 8, 4, 8, 16)),
                               (DeconvBnReLU3D(16, 8, stride=2), None, (1, 16, 2, 4, 8))):
        transposed = isinstance(mod_t, DeconvBnReLU3D)
        conv = mod_t[0] if transposed else mod_t.conv
        bn = mod_t[1] if transposed else mod_t.bn
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5, generator=g)
            bn.bias.uniform_(-0.3, 0.3, generator=g)
        import copy
        conv_r, bn_r = copy.deepcopy(conv), copy.deepcopy(bn)
        x = torch.randn(xs, generator=g)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        mod_t.train()
        bn_r.train()
        yr0 = F.relu(bn_r(conv_r(xb)))
        skip = torch.randn(yr0.shape, generator=g)
        sa, sb = skip.clone().requires_grad_(True), skip.clone().requires_grad_(True)
        y = mod_t(xa, skip=sa)
        yr = sb + yr0
        assert float((y - yr).abs().max()) < 2e-4
        gy = torch.randn(yr.shape, generator=g)
        y.backward(gy)
        yr.backward(gy)
        assert float((xa.grad - xb.grad).abs().max()) < 1e-3
        assert float((sa.grad - sb.grad).abs().max()) == 0
        assert rel_l1(conv.weight.grad, conv_r.weight.grad) < 1e-3
        assert rel_l1(bn.weight.grad, bn_r.weight.grad) < 1e-3 and rel_l1(bn.bias.grad, bn_r.bias.grad) < 1e-3
        assert torch.allclose(bn.running_mean, bn_r.running_mean, atol=1e-5)
        assert torch.allclose(bn.running_var, bn_r.running_var, atol=1e-5, rtol=1e-4)
        assert int(bn.num_batches_tracked) == 1
        mod_t.eval()
        bn_r.eval()
        with torch.no_grad():
            ye = mod_t(x, skip=skip)
            yre = skip + F.relu(bn_r(conv_r(x)))
        assert float((ye - yre).abs().max()) < 2e-4


DGRAD_BN_CASES = [   # (Cin, Cout, stride, transposed, input dims): the geometries that write a block's complete output gradient
    (8, 16, 2, False, (4, 8, 20)),      # conv1's input gradient: transposed geometry, W-parity merged (Cout' = 8)
    (16, 16, 1, False, (3, 5, 18)),     # stride 1, one Cout tile: side inputs requested before the k-loop
    (16, 8, 2, True, (2, 4, 10)),       # input gradient of a transposed layer: stride-2 geometry
    (32, 32, 1, False, (2, 3, 17)),     # two Cout tiles: side inputs in the epilogue's own phase
    (8, 1, 1, False, (4, 5, 18)),       # the probability layer: direct Cin == 1 kernel
]


@pytest.mark.parametrize("side_pre", [1, 0])
@pytest.mark.parametrize("cin,cout,stride,transposed,dims", DGRAD_BN_CASES)
def test_conv3d_dgrad_with_summand_and_batchnorm_backward_statistics(emul_lib, cin, cout, stride, transposed, dims, side_pre):
    """mvs_conv3d_dgrad / mvs_convT3d_dgrad with `add` and `bn_raw`: gx = d conv/dx + add, and the slots receive
    (sum dyh, sum dyh*xhat) of the BatchNorm+ReLU block whose raw output is bn_raw -- against ATen's input gradient and the sums
    written out in torch; then mvs_bn_relu_bwd_slots on those slots against autograd through batch_norm + relu."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(cin * 3 + cout + stride)
    b = 2
    x_shape = (b, cin) + tuple(dims)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = torch.randn(wshape, generator=g) * 0.2
    raw = torch.randn(x_shape, generator=g).requires_grad_(True)            # the block's pre-BatchNorm output
    gamma, beta = 0.5 + torch.rand(cin, generator=g), torch.randn(cin, generator=g) * 0.3
    xin = F.relu(F.batch_norm(raw, None, None, gamma, beta, True, 0.1, 1e-5))
    y = F.conv_transpose3d(xin, w, stride=stride, padding=1, output_padding=stride - 1) if transposed else F.conv3d(xin, w, stride=stride, padding=1)
    gy = torch.randn(y.shape, generator=g)
    add = torch.randn(x_shape, generator=g) if cout != 1 else None
    (gxin_ref,) = torch.autograd.grad(y, xin, gy, retain_graph=True)
    gtot = gxin_ref + (add if add is not None else 0)                        # the block's complete output gradient
    (graw_ref,) = torch.autograd.grad(xin, raw, gtot)
    mean = raw.detach().mean(dim=(0, 2, 3, 4))
    var = raw.detach().var(dim=(0, 2, 3, 4), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    stats = torch.stack([mean, invstd, gamma * invstd, beta - mean * gamma * invstd]).contiguous()
    slots = torch.zeros((8, 2, cin), dtype=torch.float64)
    emul_lib.call("mvs_set_tuning", b"side_pre", side_pre)
    if cout == 1:      # the direct Cin == 1 kernel: one voxel per thread (side_pre = 1) and four (knob cin1_vpt = 5: always; measured slower on the GPU, off by default)
        emul_lib.call("mvs_set_tuning", b"cin1_vpt", 1 if side_pre else 5)
    try:
        gx = ops.conv3d_dgrad(gy, w, x_shape, stride, transposed, add=add, bn=(raw.detach(), stats, slots))
    finally:
        emul_lib.call("mvs_set_tuning", b"side_pre", 1)
        emul_lib.call("mvs_set_tuning", b"cin1_vpt", 1)
    assert float((gx - gtot).abs().max()) < 5e-4
    view = lambda v: v.view(1, cin, 1, 1, 1)
    dyh = gtot * (raw.detach() * view(stats[2]) + view(stats[3]) > 0)
    xhat = (raw.detach() - view(mean)) * view(invstd)
    s = slots.sum(0).float()
    assert torch.allclose(s[0], dyh.sum(dim=(0, 2, 3, 4)), atol=2e-3, rtol=1e-4)
    assert torch.allclose(s[1], (dyh * xhat).sum(dim=(0, 2, 3, 4)), atol=2e-3, rtol=1e-4)
    draw, dgamma, dbeta = ops.bn_relu_bwd_slots(gx, raw.detach(), stats, slots, True)
    assert float((draw - graw_ref).abs().max()) < 1e-3 * max(1.0, float(graw_ref.abs().max()))
    assert torch.allclose(dbeta, dyh.sum(dim=(0, 2, 3, 4)), atol=2e-3, rtol=1e-4)
    assert torch.allclose(dgamma, (dyh * xhat).sum(dim=(0, 2, 3, 4)), atol=2e-3, rtol=1e-4)


@pytest.mark.skipif(os.environ.get("MVS_EMUL_FULL") != "1", reason="2 minutes of emulation; set MVS_EMUL_FULL=1 (the same golden runs on the GPU in test_gpu_parity.py::test_golden_costregnet_mvs; the per-layer conv family runs by default)")
def test_costregnet_golden(emul_lib):
    from mvs_amd.jdacs.models.mvsnet import CostRegNet
    g = load_golden("g4_costregnet_mvs")
    net = CostRegNet()
    net.load_state_dict(state_dict_from(g))
    net.train()
    x = g["x"].clone().requires_grad_(True)
    y = net(x)
    assert float((y - g["y_train"]).abs().max()) < 1e-3 * max(1.0, float(g["y_train"].abs().max()))
    y.backward(g["grad_out"])
    assert rel_l1(x.grad, g["grad_x"]) < 5e-3
    for k, p in net.named_parameters():
        assert rel_l1(p.grad, g["grad." + k]) < 5e-3, k
    sd = net.state_dict()
    for k, v in g.items():
        if k.startswith("after1.") and "num_batches" not in k:
            assert torch.allclose(sd[k[7:]], v, atol=1e-5, rtol=1e-3), k


def test_plane_sweep_bwd_segmented_windows(emul_lib):
    """Wide baseline + zoom: the tile footprint exceeds the LDS window, so the backward has to split
    the depth range into segments and, for single planes that still do not fit, fall back to global
    atomics.  Gradients must not depend on that."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(13)
    b, c, d, h, w, ns = 1, 32, 12, 24, 48, 2
    rot, trans = _cams(b, ns, h, w, g)
    trans = trans * torch.tensor([6.0, 6.0, 1.0])      # disparity sweep of tens of pixels
    rot = rot.clone()
    rot[:, 1, :2, :2] *= 3.0                            # view 1 magnifies x3: a 8x4 tile covers > 240 texels
    ref = torch.randn(b, c, h, w, generator=g, requires_grad=True)
    srcs = [torch.randn(b, c, h, w, generator=g, requires_grad=True) for _ in range(ns)]
    depth = (430 + 45.0 * torch.arange(d)).unsqueeze(0)
    var = ops.plane_sweep_variance(ref, srcs, rot, trans, depth)
    gup = torch.randn(var.shape, generator=g)
    var.backward(gup)
    got = [ref.grad.clone()] + [s.grad.clone() for s in srcs]
    for t in [ref] + srcs:
        t.grad = None
    exp = R.plane_sweep_variance(ref, srcs, [rot[:, i] for i in range(ns)], [trans[:, i] for i in range(ns)], depth)
    exp.backward(gup)
    assert float((exp != exp[:, :, :1]).float().mean()) > 0.5  # the planes really differ
    for a, t in zip(got, [ref] + srcs):
        assert float(t.grad.abs().max()) > 0
        assert float((a - t.grad).abs().max()) < 1e-3 * max(1.0, float(t.grad.abs().max()))

@pytest.mark.parametrize("c,ns", [(32, 2), (16, 3)])
def test_plane_sweep_fwd_depth_staging_forms_agree(emul_lib, c, ns):
    """Knob fwd_dl: 0 = depth loaded per plane, 1 = the slab's per-plane depths staged in LDS (default), 2 = + gathers waited
    for inside the re-gather block.  Same arithmetic: the three volumes are bit-identical."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(9)
    b, d, h, w = 1, 11, 9, 14
    rot, trans = _cams(b, ns, h, w, g)
    ref = torch.randn(b, c, h, w, generator=g)
    srcs = [torch.randn(b, c, h, w, generator=g) for _ in range(ns)]
    depth = (430 + 35.0 * torch.arange(d)).unsqueeze(0).repeat(b, 1)
    vols = []
    try:
        for dl, pt in ((0, 1), (1, 1), (2, 1), (1, 0)):     # fwd_pt = 1: the projection-table kernel serves dl != 0 (off by default: measured slower)
            emul_lib.call("mvs_set_tuning", b"fwd_dl", dl)
            emul_lib.call("mvs_set_tuning", b"fwd_pt", pt)
            with torch.no_grad():
                vols.append(ops.plane_sweep_variance(ref, srcs, rot, trans, depth).clone())
    finally:
        emul_lib.call("mvs_set_tuning", b"fwd_dl", 2)
        emul_lib.call("mvs_set_tuning", b"fwd_pt", 0)
    exp = R.plane_sweep_variance(ref, srcs, [rot[:, i] for i in range(ns)], [trans[:, i] for i in range(ns)], depth)
    assert float((vols[1] - exp).abs().max()) < 2e-4
    assert torch.equal(vols[0], vols[1]) and torch.equal(vols[1], vols[2]) and torch.equal(vols[1], vols[3])


@pytest.mark.parametrize("c,ns,d,hw,per_pixel", [(8, 1, 1, (2, 3), False), (8, 1, 1, (2, 2), True), (16, 2, 2, (3, 2), False),
                                                  (32, 1, 3, (2, 5), False)])
def test_plane_sweep_smallest_shapes(emul_lib, c, ns, d, hw, per_pixel):
    """The smallest shapes the C ABI accepts (one depth plane, one source view, images of 2 x 2 ... pixels: tiles, slabs and
    depth segments are all partial), forward and backward."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(21)
    b = 1
    h, w = hw
    rot, trans = _cams(b, ns, 8, 12, g)     # cameras of a larger image: sample points fall in and out of the tiny maps
    ref = torch.randn(b, c, h, w, generator=g, requires_grad=True)
    srcs = [torch.randn(b, c, h, w, generator=g, requires_grad=True) for _ in range(ns)]
    if per_pixel:
        depth = 450 + 30 * torch.rand(b, d, h, w, generator=g)
    else:
        depth = (430 + 35.0 * torch.arange(d)).unsqueeze(0).repeat(b, 1)
    var = ops.plane_sweep_variance(ref, srcs, rot, trans, depth)
    gup = torch.randn(var.shape, generator=g)
    var.backward(gup)
    got = [ref.grad.clone()] + [s.grad.clone() for s in srcs]
    for t in [ref] + srcs:
        t.grad = None
    exp = R.plane_sweep_variance(ref, srcs, [rot[:, i] for i in range(ns)], [trans[:, i] for i in range(ns)], depth)
    exp.backward(gup)
    assert float((var - exp).abs().max()) < 2e-4
    for a, t in zip(got, [ref] + srcs):
        assert float((a - t.grad).abs().max()) < 1e-3 * max(1.0, float(t.grad.abs().max()))


def test_c_abi_rejects_bad_arguments_with_a_message(emul_lib):
    """Error behaviour of the boundary (include/mvs_hip.h): unsupported channel counts, too few views, degenerate images and null
    pointers come back as error codes with a message in mvs_last_error(), and the Python mirror raises ValueError like the
    reference's shape errors do -- nothing is launched."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(1)
    rot, trans = _cams(1, 1, 8, 12, g)
    depth = torch.full((1, 2), 500.0)
    with pytest.raises(ValueError, match="C must be 8, 16 or 32"):
        ops.plane_sweep_variance(torch.randn(1, 12, 8, 12), [torch.randn(1, 12, 8, 12)], rot, trans, depth)
    with pytest.raises(ValueError):
        ops.plane_sweep_variance(torch.randn(1, 8, 1, 12), [torch.randn(1, 8, 1, 12)], rot, trans, depth)   # H must be > 1
    with pytest.raises(ValueError):
        ops.plane_sweep_variance(torch.randn(1, 8, 8, 12), [], rot[:, :0], trans[:, :0], depth)              # no source view
    with pytest.raises(ValueError, match="unknown|not a tuning key|mvs_set_tuning"):
        emul_lib.call("mvs_set_tuning", b"no_such_knob", 1)
    assert emul_lib.raw("mvs_set_tuning", b"bwd", 1) != 0        # full-string keys: a prefix of a real key is rejected


@pytest.mark.parametrize("d,hw,per_pixel", [(1, (1, 1), False), (2, (3, 5), False), (3, (1, 17), True), (5, (2, 9), False), (9, (7, 3), True)])
def test_softargmin_smallest_shapes(emul_lib, d, hw, per_pixel):
    """Soft-argmin + confidence at shapes smaller than one wave's 16 pixels x 4 depth slices (D = 1: the confidence window
    [idx - 1, idx + 2] is clipped on both sides), forward and backward vs the oracle (jdacs/models/module.py:145-151)."""
    from mvs_amd import ops
    gen = torch.Generator().manual_seed(d * 10 + hw[1])
    b = 2
    h, w = hw
    lg = (torch.randn(b, d, h, w, generator=gen) * 3).requires_grad_(True)
    hyp = 500 + torch.rand(b, d, h, w, generator=gen) * 50 if per_pixel else (425 + 7.0 * torch.arange(d)).unsqueeze(0).repeat(b, 1)
    dep, conf = ops.softargmin_conf(lg, hyp)
    gd = torch.randn(dep.shape, generator=gen)
    dep.backward(gd)
    got = lg.grad.clone()
    lg.grad = None
    e, ec, _ = R.softargmin_conf(lg, hyp)
    e.backward(gd)
    assert float((dep - e).abs().max()) < 1e-3 and float((conf - ec).abs().max()) < 1e-5
    assert float((got - lg.grad).abs().max()) < 1e-4 * max(1.0, float(lg.grad.abs().max()))


@pytest.mark.parametrize("cin,cout,stride,transposed,dims", [(8, 8, 1, False, (1, 1, 1)), (32, 8, 1, False, (1, 2, 3)),
                                                             (8, 16, 2, False, (2, 2, 2)), (16, 8, 2, True, (1, 1, 1)),
                                                             (16, 16, 1, False, (1, 1, 17)), (8, 1, 1, False, (1, 1, 2))])
def test_conv3d_smallest_volumes(emul_lib, cin, cout, stride, transposed, dims):
    """Volumes smaller than one workgroup tile in every dimension (every tile is partial, every tap of some voxels is padding):
    forward, input gradient and weight gradient vs ATen (mvsnet.py:40-74 layer shapes)."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(cin + cout + dims[2])
    x = torch.randn(1, cin, *dims, generator=g, requires_grad=True)
    if transposed:
        w = (torch.randn(cin, cout, 3, 3, 3, generator=g) * 0.2).requires_grad_(True)
        yr = F.conv_transpose3d(x, w, stride=stride, padding=1, output_padding=stride - 1)
    else:
        w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.2).requires_grad_(True)
        yr = F.conv3d(x, w, stride=stride, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    with torch.no_grad():
        y, _ = ops.conv3d_forward(x, w, stride, transposed)
        gx = ops.conv3d_dgrad(gy, w, tuple(x.shape), stride, transposed)
        gw = ops.conv3d_wgrad(x, gy, tuple(w.shape), stride, transposed)
    assert float((y - yr).abs().max()) < 2e-4
    assert float((gx - x.grad).abs().max()) < 3e-4
    assert float((gw - w.grad).abs().max()) < 3e-4 * max(1.0, float(w.grad.abs().max()))


@pytest.mark.parametrize("b,dims,layout", [(1, (5, 6, 19), "conv"), (2, (4, 4, 16), "conv"), (1, (2, 9, 33), "transposed_forward")],
                         ids=["ragged", "two_whole_tiles", "oik_layout_through_the_transposed_forward"])
def test_conv0_input_gradient_split_bf16_form(emul_lib, b, dims, layout):
    """Opt-in knob conv0_x3 (csrc/conv3d_x3.hip): the 8 -> 32 channel stride-1 convolution behind conv0's input gradient
    (mvsnet.py:40 backward) as six bf16 MFMA products of three-term splits of the fp32 operands.  Against fp64 it must be as close
    as the fp32-MFMA kernel (it is closer), with partial tiles, a batch, and BOTH weight layouts / tap orders the dispatcher hands
    it (conv dgrad: [in'][out'] flipped; a transposed stride-1 forward: the same; checked against ATen in fp64)."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(sum(dims))
    w = torch.randn(8, 32, 3, 3, 3, generator=g) * 0.2
    gy = torch.randn(b, 8, *dims, generator=g) * torch.rand(b, 8, *dims, generator=g).pow(4) * 10     # a wide dynamic range
    if layout == "conv":
        ref = torch.nn.grad.conv3d_input((b, 32, *dims), w.double(), gy.double(), padding=1)
        run = lambda: ops.conv3d_dgrad(gy, w, (b, 32, *dims), 1, False)
    else:     # ConvTranspose3d(8 -> 32, stride 1) forward: weight [Cin = 8][Cout = 32]
        ref = F.conv_transpose3d(gy.double(), w.double(), stride=1, padding=1)
        run = lambda: ops.conv3d_forward(gy, w, 1, True)[0]
    try:
        emul_lib.call("mvs_set_tuning", b"conv0_x3", 0)
        base = run()
        emul_lib.call("mvs_set_tuning", b"conv0_x3", 1)
        got = run()
    finally:
        emul_lib.call("mvs_set_tuning", b"conv0_x3", 0)
    assert not torch.equal(base, got)                          # (the knob did select another kernel)
    e0 = (base.double() - ref).abs().sum() / ref.abs().sum()
    e1 = (got.double() - ref).abs().sum() / ref.abs().sum()
    assert float(e1) < 1.2 * float(e0) + 1e-8 and float(e1) < 5e-7, (float(e0), float(e1))
    assert float((got.double() - ref).abs().max()) < 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize("b,dims", [(1, (5, 6, 19)), (2, (4, 9, 16)), (1, (3, 17, 5))], ids=["ragged", "batch_2_two_columns", "odd_depth_three_columns"])
def test_conv0_forward_split_bf16_form(emul_lib, b, dims):
    """Opt-in knob conv0_x3 bit 1 (csrc/conv3d_x3.hip: conv_x3_fwd_march_kernel): conv0's forward (mvsnet.py:40, 32 -> 8 channels) as
    split-bf16 products, marching along D with two output slices per MFMA.  Output and fused BatchNorm statistics against ATen in fp64:
    as close as the fp32-MFMA kernel; odd depth (a half-filled last pair), partial columns, a batch."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(sum(dims))
    w = torch.randn(8, 32, 3, 3, 3, generator=g) * 0.2
    x = torch.randn(b, 32, *dims, generator=g) * torch.rand(b, 32, *dims, generator=g).pow(4) * 10
    ref = F.conv3d(x.double(), w.double(), padding=1)
    try:
        emul_lib.call("mvs_set_tuning", b"conv0_x3", 0)
        base, _ = ops.conv3d_forward(x, w, 1, False, want_stats=True)
        emul_lib.call("mvs_set_tuning", b"conv0_x3", 2)
        got, slots = ops.conv3d_forward(x, w, 1, False, want_stats=True)
    finally:
        emul_lib.call("mvs_set_tuning", b"conv0_x3", 0)
    assert not torch.equal(base, got)
    e0 = (base.double() - ref).abs().sum() / ref.abs().sum()
    e1 = (got.double() - ref).abs().sum() / ref.abs().sum()
    assert float(e1) < 1.2 * float(e0) + 1e-8 and float(e1) < 6e-7, (float(e0), float(e1))
    assert float((got.double() - ref).abs().max()) < 3e-6 * float(ref.abs().max())
    st = slots.sum(0)
    assert float(((st[0] - ref.sum((0, 2, 3, 4))).abs() / ref.abs().sum((0, 2, 3, 4))).max()) < 1e-6
    assert float(((st[1] - ref.pow(2).sum((0, 2, 3, 4))).abs() / ref.pow(2).sum((0, 2, 3, 4))).max()) < 1e-6


def test_relative_projections_one_launch(emul_lib):
    """mvs_relative_projection (all source views, fp64 Gauss-Jordan + product) vs the reference's lines
    torch.matmul(src_proj, torch.inverse(ref_proj)) per view (jdacs/models/module.py:116-118), on DTU-like cameras
    (fp64 truth: the kernel must be at least as accurate as the fp32 torch path) and on a matrix that needs pivoting."""
    from mvs_amd import ops
    K, E = R.synthetic_cameras(5, 128, 160, 640)
    P = E.clone()
    P[:, :3, :4] = K @ E[:, :3, :4]
    ref = torch.stack([P[0], P[2]], 0)                      # B = 2
    srcs = [torch.stack([P[1], P[3]], 0), torch.stack([P[4], P[1]], 0), torch.stack([P[3], P[0]], 0)]
    rot, trans = ops.relative_projections(srcs, ref)
    assert rot.shape == (2, 3, 3, 3) and trans.shape == (2, 3, 3)
    for s, sp in enumerate(srcs):
        t64 = sp.double() @ torch.linalg.inv(ref.double())
        e32 = torch.matmul(sp, torch.inverse(ref))
        for got, exp, tru in ((rot[:, s], e32[:, :3, :3], t64[:, :3, :3]), (trans[:, s], e32[:, :3, 3], t64[:, :3, 3])):
            scale = float(tru.abs().max())
            assert float((got.double() - tru).abs().max()) <= max(float((exp.double() - tru).abs().max()), 2e-7 * scale)
            assert float((got - exp).abs().max()) < 1e-5 * scale
    # zero on the diagonal: needs the row exchange
    refp = torch.tensor([[[0.0, 2.0, 0.0, 1.0], [1.0, 0.0, 0.0, 2.0], [0.0, 0.0, 0.0, 3.0], [0.0, 0.0, 4.0, 1.0]]])
    srcp = [torch.eye(4).unsqueeze(0)]
    rot, trans = ops.relative_projections(srcp, refp)
    inv = torch.linalg.inv(refp.double())[0]
    assert float((rot[0, 0].double() - inv[:3, :3]).abs().max()) < 1e-6 and float((trans[0, 0].double() - inv[:3, 3]).abs().max()) < 1e-6
    with pytest.raises(ValueError):
        ops.relative_projections([torch.eye(3).unsqueeze(0)], torch.eye(4).unsqueeze(0))



@pytest.mark.parametrize("variant", [0, 2, 3, 4])
def test_plane_sweep_fwd_variants_agree(emul_lib, variant):
    """The forward variants (0: taps through L1 -- the direct kernel, what plain warps and 5 / 7+ source views run; 2-4: register-cached
    blocks with 4 / 8 / 16 channels per thread) against the oracle."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(17)
    b, c, d, h, w, ns = 1, 32, 20, 16, 24, 2
    rot, trans = _cams(b, ns, h, w, g)
    trans = trans * torch.tensor([3.0, 3.0, 1.0])   # several texel crossings along the sweep
    ref = torch.randn(b, c, h, w, generator=g)
    srcs = [torch.randn(b, c, h, w, generator=g) for _ in range(ns)]
    depth = (430 + 25.0 * torch.arange(d)).unsqueeze(0)
    emul_lib.call("mvs_set_tuning", b"sweep_fwd", variant)
    try:
        var = ops.plane_sweep_variance(ref, srcs, rot, trans, depth)
    finally:
        emul_lib.call("mvs_set_tuning", b"sweep_fwd", 3)
    exp = R.plane_sweep_variance(ref, srcs, [rot[:, i] for i in range(ns)], [trans[:, i] for i in range(ns)], depth)
    assert float((var - exp).abs().max()) < 2e-4


def test_conv_multi_cout_tiles_per_workgroup(emul_lib):
    """Large volumes keep all 16-wide Cout tiles in one workgroup (NB = 2 / 4); tiny test volumes would
    otherwise always take the split path."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(23)
    emul_lib.call("mvs_set_tuning", b"conv_split", 0)
    try:
        for cin, cout, stride, transposed, dims in ((8, 32, 1, False, (4, 4, 16)), (16, 64, 1, False, (4, 4, 16)),
                                                    (32, 32, 2, True, (2, 4, 8))):
            x = torch.randn(1, cin, *dims, generator=g)
            wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
            w = torch.randn(wshape, generator=g) * 0.2
            yr = (F.conv_transpose3d(x, w, stride=stride, padding=1, output_padding=stride - 1) if transposed
                  else F.conv3d(x, w, stride=stride, padding=1))
            y, parts = ops.conv3d_forward(x, w, stride, transposed, want_stats=True)
            assert float((y - yr).abs().max()) < 2e-4
            assert torch.allclose(parts.sum(0)[0].float(), yr.sum(dim=(0, 2, 3, 4)), atol=1e-2, rtol=1e-4)
    finally:
        emul_lib.call("mvs_set_tuning", b"conv_split", 1)


def test_bn_relu_2d(emul_lib):
    """BatchNorm2d + ReLU of the feature extractor's ConvBnReLU through the library's BN kernels."""
    import copy
    from mvs_amd import ops
    g = torch.Generator().manual_seed(31)
    bn = torch.nn.BatchNorm2d(16)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g)
        bn.bias.uniform_(-0.3, 0.3, generator=g)
    bn_r = copy.deepcopy(bn)
    x = (torch.randn(2, 16, 12, 20, generator=g) * 2 + 0.5).contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y = ops.BnReLUFn.apply(xa, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, bn.eps, 0.1, 1)
    yr = F.relu(bn_r(xb))
    assert float((y - yr).abs().max()) < 1e-5
    gy = torch.randn(yr.shape, generator=g)
    y.backward(gy)
    yr.backward(gy)
    assert float((xa.grad - xb.grad).abs().max()) < 1e-5
    assert rel_l1(bn.weight.grad, bn_r.weight.grad) < 1e-5 and rel_l1(bn.bias.grad, bn_r.bias.grad) < 1e-5
    assert torch.allclose(bn.running_mean, bn_r.running_mean, atol=1e-6)
    assert torch.allclose(bn.running_var, bn_r.running_var, atol=1e-6, rtol=1e-5)
    bn_r.eval()
    with torch.no_grad():
        ye = ops.BnReLUFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, False, bn.eps, 0.1, 1)
    assert float((ye - F.relu(bn_r(x))).abs().max()) < 1e-5


def test_bn_relu_2d_grouped_equals_successive_calls(emul_lib):
    """3 views stacked along the batch dim with groups=3 == three successive BatchNorm2d calls (statistics,
    running stats, gradients of the shared affine parameters)."""
    import copy
    from mvs_amd import ops
    g = torch.Generator().manual_seed(37)
    bn = torch.nn.BatchNorm2d(8)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g)
        bn.bias.uniform_(-0.3, 0.3, generator=g)
    bn_r = copy.deepcopy(bn)
    views = [(torch.randn(2, 8, 10, 12, generator=g) * (1 + i) + 0.3 * i) for i in range(3)]
    x = torch.cat(views, 0).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    vr = [v.clone().requires_grad_(True) for v in views]
    y = ops.BnReLUFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, bn.eps, 0.1, 3)
    yr = torch.cat([F.relu(bn_r(v)) for v in vr], 0)
    assert float((y - yr).abs().max()) < 1e-5
    gy = torch.randn(yr.shape, generator=g)
    y.backward(gy)
    yr.backward(gy)
    assert float((x.grad - torch.cat([v.grad for v in vr], 0)).abs().max()) < 1e-5
    assert rel_l1(bn.weight.grad, bn_r.weight.grad) < 1e-5 and rel_l1(bn.bias.grad, bn_r.bias.grad) < 1e-5
    assert torch.allclose(bn.running_mean, bn_r.running_mean, atol=1e-6)
    assert torch.allclose(bn.running_var, bn_r.running_var, atol=1e-6, rtol=1e-5)


@pytest.mark.skipif(os.environ.get("MVS_EMUL_FULL") != "1", reason="4 minutes of emulation; set MVS_EMUL_FULL=1 (the same golden runs on the GPU in test_gpu_parity.py::test_golden_costregnet_cvp)")
def test_costregnet_cvp_golden(emul_lib):
    """jdacs-ms regulariser (stride-1 transposed conv, 16->1 prob layer) vs the fixture from the imported reference."""
    from mvs_amd.jdacs_ms.models.network import CostRegNet
    g = load_golden("g4_costregnet_cvp")
    net = CostRegNet()
    net.load_state_dict(state_dict_from(g))
    net.train()
    x = g["x"].clone().requires_grad_(True)
    y = net(x)
    assert float((y - g["y_train"]).abs().max()) < 1e-3 * max(1.0, float(g["y_train"].abs().max()))
    y.backward(g["grad_out"])
    assert rel_l1(x.grad, g["grad_x"]) < 5e-3
    for k, p in net.named_parameters():
        assert rel_l1(p.grad, g["grad." + k]) < 5e-3, k


@pytest.mark.parametrize("cin,dims,xcd", [(32, (4, 4, 18), 1), (16, (5, 6, 16), 1), (8, (3, 4, 33), 0), (8, (2, 18, 16), 1)])
def test_conv_c8_broadcast_operand_forward(emul_lib, cin, dims, xcd):
    """Cout == 8 stride-1 forward with the weights as the MFMA broadcast operand (tuning k8 = 7): conv, epilogue
    (scale/shift/relu/skip), BN stat partials and both tile orders vs F.conv3d (mvsnet.py:40 conv0, network.py:47)."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(cin + dims[1])
    x = torch.randn(2 if (cin == 8 and dims[2] == 33) else 1, cin, *dims, generator=g)   # (the emulated MFMA is a 64-thread barrier: keep the tile count small)
    w = torch.randn(8, cin, 3, 3, 3, generator=g) * 0.2
    yr = F.conv3d(x, w, padding=1)
    emul_lib.call("mvs_set_tuning", b"k8", 7)
    emul_lib.call("mvs_set_tuning", b"xcd", xcd)
    try:
        y, parts = ops.conv3d_forward(x, w, 1, False, want_stats=True)
        assert float((y - yr).abs().max()) < 2e-4
        s = parts.sum(0).float()
        assert torch.allclose(s[0], yr.sum(dim=(0, 2, 3, 4)), atol=1e-2, rtol=1e-4)
        assert torch.allclose(s[1], (yr ** 2).sum(dim=(0, 2, 3, 4)), atol=1e-2, rtol=1e-4)
        scale = torch.rand(8, generator=g) + 0.5
        shift = torch.randn(8, generator=g)
        skip = torch.randn(yr.shape, generator=g)
        y2, _ = ops.conv3d_forward(x, w, 1, False, scale=scale, shift=shift, relu=True, skip=skip)
        ref2 = F.relu(yr * scale.view(1, 8, 1, 1, 1) + shift.view(1, 8, 1, 1, 1)) + skip
        assert float((y2 - ref2).abs().max()) < 3e-4
        gy = torch.randn(yr.shape, generator=g)
        wr = w.clone().requires_grad_(True)
        F.conv3d(x, wr, padding=1).backward(gy)
        gw = ops.conv3d_wgrad(x, gy, tuple(w.shape), 1, False)   # cin 16/32: output gradient as the broadcast operand
        assert float((gw - wr.grad).abs().max()) < 1e-3 * max(1.0, float(wr.grad.abs().max()))
        # transposed stride-1 weights (dgrad of a 8 -> cin conv maps onto this path only for cout == 8, i.e. cin == 8)
        if cin == 8:
            gx = ops.conv3d_dgrad(yr, w, tuple(x.shape), 1, False)
            xr = x.clone().requires_grad_(True)
            F.conv3d(xr, w, padding=1).backward(yr)
            assert float((gx - xr.grad).abs().max()) < 2e-3
    finally:
        emul_lib.call("mvs_set_tuning", b"k8", 1)
        emul_lib.call("mvs_set_tuning", b"xcd", 1)


@pytest.mark.parametrize("name", ["g8_unsup_loss", "g8_unsup_loss_n4"])
def test_unsup_loss_golden(emul_lib, name):
    """UnSupLoss through the kernels (csrc/unsup_loss.hip) vs the fixture from the imported reference:
    total, the three terms and d total / d depth."""
    from mvs_amd.jdacs.losses.unsup_loss import UnSupLoss
    g = load_golden(name)
    depth = g["depth"].clone().requires_grad_(True)
    crit = UnSupLoss()
    total = crit(g["imgs"].float(), g["cams"], depth)
    (2.0 * total).backward()
    assert abs(float(total) - float(g["loss"])) < 3e-5 * abs(float(g["loss"]))
    assert abs(float(crit.reconstr_loss) - float(g["reconstr_loss"])) < 2e-5
    assert abs(float(crit.ssim_loss) - float(g["ssim_loss"])) < 2e-5
    assert abs(float(crit.smooth_loss) - float(g["smooth_loss"])) < 2e-4
    gd = g["grad_depth"] * 2.0
    assert float((depth.grad - gd).abs().max()) < 4e-6 + 2e-4 * float(gd.abs().max())


def test_unsup_loss_vs_oracle_ragged_and_errors(emul_lib):
    """odd image sizes, batch 3, N = 7, a different smoothness weight; argument checks."""
    from mvs_amd.jdacs.losses.unsup_loss import UnSupLoss
    gen = torch.Generator().manual_seed(3)
    b, n, h, w = 3, 7, 52, 76          # quarter resolution 13 x 19
    imgs = F.avg_pool2d(torch.randn(b * n, 3, h, w, generator=gen), 5, 1, 2).view(b, n, 3, h, w) * 3
    K, E = R.synthetic_cameras(n, h // 4, w // 4, w)
    cams = torch.zeros(b, n, 2, 4, 4)
    cams[:, :, 0] = E
    cams[:, :, 1, :3, :3] = K
    depth = (600.0 + 40.0 * torch.rand(b, h // 4, w // 4, generator=gen))
    da, db = depth.clone().requires_grad_(True), depth.clone().requires_grad_(True)
    crit = UnSupLoss(smooth_lambda=0.5)
    la = crit(imgs, cams, da)
    lb = R.unsup_loss(imgs, cams, db, smooth_lambda=0.5)
    la.backward()
    lb.backward()
    assert abs(float(la) - float(lb)) < 3e-5 * abs(float(lb))
    assert float((da.grad - db.grad).abs().max()) < 4e-6 + 2e-4 * float(db.grad.abs().max())
    with pytest.raises(ValueError):
        crit(imgs[:, :3], cams[:, :3], depth)          # two source views: no top-3
    with pytest.raises(ValueError):
        crit(imgs, cams[:, :5], depth)
    with pytest.raises(ValueError):
        crit(imgs, cams, depth[:, :-1])


def test_cal_depth_hypo_golden_and_oracle(emul_lib):
    """calDepthHypo through the fused kernel pair vs the fixture from the imported reference and vs the
    oracle on a ragged batch-2 case."""
    from mvs_amd.jdacs_ms.models import modules as M
    g = load_golden("g7_cvpmvsnet_e2e")
    hyp = M.calDepthHypo(None, g["depth_up"], g["ref_in"], g["src_in"], g["ref_ex"], g["src_ex"], None, None, 0)
    assert hyp.shape == g["hypos0"].shape and hyp.dtype == torch.float32
    assert float((hyp - g["hypos0"]).abs().max()) < 1e-3
    gen = torch.Generator().manual_seed(4)
    b, h, w = 2, 37, 53
    K, E = R.synthetic_cameras(3, h, w, 4 * w)
    ref_in, src_in = K.unsqueeze(0).repeat(b, 1, 1), K.view(1, 1, 3, 3).repeat(b, 2, 1, 1)
    ref_ex, src_ex = E[0].unsqueeze(0).repeat(b, 1, 1), E[1:].unsqueeze(0).repeat(b, 1, 1, 1).clone()
    src_ex[1, :, :3, 3] *= 1.4
    depth = 600.0 + 80.0 * torch.rand(b, h, w, generator=gen)
    a = M.calDepthHypo(None, depth, ref_in, src_in, ref_ex, src_ex, None, None, 1)
    e = R.cal_depth_hypo(depth, ref_in, src_in, ref_ex, src_ex)
    assert float((a - e).abs().max()) < 1e-3


@pytest.mark.skipif(os.environ.get("MVS_EMUL_FULL") != "1", reason="2.5 minutes of emulation; set MVS_EMUL_FULL=1 (the GPU version is test_gpu_parity.py::test_config3_self_supervised_step_vs_gpu_oracle)")
def test_jdacs_self_supervised_step_end_to_end(emul_lib):
    """BASELINE config 3 in miniature (N = 4 views, 32x64 images, D = 8): MVSNet forward -> UnSupLoss on its depth map ->
    backward into the network, every kernel of the path in one graph, vs the oracle's MVSNet + UnSupLoss."""
    from mvs_amd.jdacs.losses.unsup_loss import UnSupLoss
    from mvs_amd.jdacs.models.mvsnet import MVSNet
    torch.manual_seed(5)
    net = MVSNet(refine=False)
    with torch.no_grad():
        net.cost_regularization.prob.weight.mul_(30.0)
    oracle = R.OracleMVSNet(refine=False)
    oracle.load_state_dict(net.state_dict())
    net.train()
    oracle.train()
    b, n, h, w, d = 1, 4, 32, 64, 8
    imgs, proj, dv = R.synthetic_mvsnet_inputs(b, n, h, w, d, seed=2, depth_min=600.0, interval=8.0)
    imgs = F.avg_pool2d(imgs.view(b * n, 3, h, w), 5, 1, 2).view(b, n, 3, h, w) * 3
    K, E = R.synthetic_cameras(n, h // 4, w // 4, w)
    cams = torch.zeros(b, n, 2, 4, 4)
    cams[:, :, 0] = E
    cams[:, :, 1, :3, :3] = K
    la = UnSupLoss()(imgs, cams, net(imgs, proj, dv)["depth"])
    lb = R.unsup_loss(imgs, cams, oracle(imgs, proj, dv)["depth"])
    la.backward()
    lb.backward()
    assert abs(float(la) - float(lb)) < 1e-4 * abs(float(lb))
    pa, pb = dict(net.named_parameters()), dict(oracle.named_parameters())
    checked = 0
    for k in ("cost_regularization.prob.weight", "cost_regularization.conv0.conv.weight", "feature.conv0.conv.weight",
              "cost_regularization.conv7.0.weight"):
        if pb[k].grad is not None and float(pb[k].grad.abs().max()) > 0:
            assert rel_l1(pa[k].grad, pb[k].grad) < 2e-2, k
            checked += 1
    assert checked >= 3


@pytest.mark.parametrize("cin,cout,ks,stride,hw", [(3, 8, 3, 1, (11, 37)), (8, 8, 3, 1, (8, 32)), (8, 8, 3, 1, (9, 35)), (4, 3, 3, 1, (5, 66)), (8, 16, 5, 2, (18, 70)),
                                                    (16, 16, 3, 1, (9, 33)), (16, 32, 5, 2, (17, 41)), (32, 32, 3, 1, (10, 20))])
def test_conv2d_family(emul_lib, cin, cout, ks, stride, hw):
    """SURVEY 8(f)-3 first cut: every 2-D convolution shape of FeatureNet (mvsnet.py:17-34) -- forward (+ bias), input and
    weight gradient -- vs ATen, on ragged image sizes."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(cin * 3 + cout + ks)
    x = torch.randn(1 if ks == 5 else 2, cin, *hw, generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, ks, ks, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=stride, padding=ks // 2)
    xa, wa, ba = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = ops.Conv2dFn.apply(xa, wa, ba, stride)
    assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert float((y - yr).abs().max()) < 2e-4
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    y.backward(gy)
    assert float((xa.grad - xr.grad).abs().max()) < 3e-4
    assert float((wa.grad - wr.grad).abs().max()) < 1e-3 * max(1.0, float(wr.grad.abs().max()))
    assert float((ba.grad - br.grad).abs().max()) < 1e-3 * max(1.0, float(br.grad.abs().max()))
    y2 = ops.conv2d_forward(x, w, None, stride)
    assert float((y2 - F.conv2d(x, w, None, stride=stride, padding=ks // 2)).abs().max()) < 2e-4
    if ks == 3 and min(cin, cout) <= 8:   # forward and / or input gradient above ran as pixel-pair GEMMs (knob conv2d_pp); here without
        emul_lib.call("mvs_set_tuning", b"conv2d_pp", 0)
        try:
            y0 = ops.conv2d_forward(x, w, b, stride)
            gx0 = ops.conv2d_dgrad(gy, w, tuple(x.shape), stride)
        finally:
            emul_lib.call("mvs_set_tuning", b"conv2d_pp", 1)
        assert float((y0 - yr).abs().max()) < 2e-4 and float((gx0 - xr.grad).abs().max()) < 3e-4
        assert float((y0 - y.detach()).abs().max()) < 2e-5      # same products, another summation order
    if stride == 2:   # the other forms of the stride-2 input gradient: direct VALU (0), four parity-class passes (1); default: ONE pass, compacted taps (2)
        for form in (0, 1):
            emul_lib.call("mvs_set_tuning", b"conv2d_s2_mfma", form)
            try:
                gx = ops.conv2d_dgrad(gy, w, tuple(x.shape), 2)
            finally:
                emul_lib.call("mvs_set_tuning", b"conv2d_s2_mfma", 2)
            assert float((gx - xr.grad).abs().max()) < 3e-4, form


@pytest.mark.skipif(os.environ.get("MVS_EMUL_FULL") != "1", reason="45 s of emulation; set MVS_EMUL_FULL=1 (the GPU version is test_gpu_parity.py::test_featurenet_hip_convs_vs_stock; test_conv2d_family runs by default)")
def test_featurenet_through_hip_convs(emul_lib, monkeypatch):
    """FeatureNet (mvsnet.py:17-34) with its convolutions through csrc/conv2d.hip (ConvBnReLU.hip_conv) vs the stock path:
    three views batched with per-view BatchNorm statistics, forward + parameter gradients."""
    import copy
    from mvs_amd.jdacs.models import module as MM
    from mvs_amd.jdacs.models.mvsnet import FeatureNet
    torch.manual_seed(2)
    a = FeatureNet().train()
    b = copy.deepcopy(a).train()
    x = torch.randn(3, 3, 16, 24)
    monkeypatch.setattr(MM.ConvBnReLU, "hip_conv", False)
    yb = b(x, 3)
    yb.square().mean().backward()
    monkeypatch.setattr(MM.ConvBnReLU, "hip_conv", True)
    monkeypatch.setattr(MM, "conv2d_maybe_hip", lambda conv, t: __import__("mvs_amd").ops.Conv2dFn.apply(t, conv.weight, conv.bias, conv.stride[0]))
    import mvs_amd.jdacs.models.mvsnet as MV
    monkeypatch.setattr(MV, "conv2d_maybe_hip", MM.conv2d_maybe_hip)
    ya = a(x, 3)
    ya.square().mean().backward()
    assert ya.shape == (3, 32, 4, 6)
    assert float((ya - yb).abs().max()) < 1e-3
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert rel_l1(p.grad, q.grad) < 2e-2, k


def test_conv2d_wgrad_persistent_workgroups_walk_several_tiles(emul_lib):
    """The weight gradient's persistent workgroups accumulate over several tiles (forced here with 3 workgroups for 12 tiles;
    at FeatureNet sizes there are thousands of tiles for 256 workgroups)."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 8, 20, 70, generator=g).contiguous(memory_format=torch.channels_last)   # 3 x 3 tiles per image
    gy = torch.randn(2, 16, 20, 70, generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(16, 8, 3, 3, requires_grad=True)
    F.conv2d(x, w, padding=1).backward(gy)
    emul_lib.call("mvs_set_tuning", b"wgrad2d_groups", 3)
    try:
        gw = ops.conv2d_wgrad(x, gy, (16, 8, 3, 3), 1)
    finally:
        emul_lib.call("mvs_set_tuning", b"wgrad2d_groups", 256)
    assert float((gw - w.grad).abs().max()) < 1e-3 * max(1.0, float(w.grad.abs().max()))


# ---- bf16-storage inference path (BASELINE configs[4]) -------------------------------------------------------------------
BF16_CONV_CASES = [(32, 8, 1, False, (5, 6, 20)), (32, 8, 1, False, (7, 3, 33)), (16, 16, 1, False, (4, 5, 18)), (64, 64, 1, False, (3, 4, 17)), (8, 1, 1, False, (6, 5, 19)), (16, 1, 1, False, (5, 9, 18)),
                   (8, 16, 2, False, (6, 8, 34)), (32, 64, 2, False, (4, 6, 18)), (64, 32, 2, True, (2, 3, 9)), (16, 8, 2, True, (3, 4, 17)), (16, 8, 2, True, (5, 5, 9))]


@pytest.mark.parametrize("cin,cout,stride,transposed,dims", BF16_CONV_CASES)
def test_conv3d_bf16_inference(emul_lib, cin, cout, stride, transposed, dims):
    """bf16 activations / fp32 accumulation vs torch's fp32 convolution of the SAME bf16-rounded operands: what is left is the
    summation order and the bf16 rounding of the output (<= 2^-8 relative)."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(7)
    d, h, w = dims
    x = torch.randn(2, cin, d, h, w, generator=g).bfloat16().contiguous(memory_format=torch.channels_last_3d)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    wt = torch.randn(wshape, generator=g) * (0.3 / (cin ** 0.5))
    scale = 0.5 + torch.rand(cout, generator=g)
    shift = torch.randn(cout, generator=g) * 0.1
    wr = wt.bfloat16().float()
    if transposed:
        ref = F.conv_transpose3d(x.float(), wr, stride=2, padding=1, output_padding=1)
    else:
        ref = F.conv3d(x.float(), wr, stride=stride, padding=1)
    ref_bn = torch.relu(ref * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1))
    skip = torch.randn(ref.shape, generator=g).bfloat16().contiguous(memory_format=torch.channels_last_3d)
    with torch.no_grad():
        y = ops.conv3d_forward_bf16(x, wt, stride, transposed, scale=scale, shift=shift, skip=skip, relu=True)
        yb = ops.conv3d_forward_bf16(x, wt, stride, transposed, shift=shift, out_f32=True)     # bias only, fp32 out (prob layer form)
    assert y.dtype == torch.bfloat16 and y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last_3d)
    exp = ref_bn + skip.float()
    assert float((y.float() - exp).abs().max()) <= 2 ** -7 * float(exp.abs().max()) + 1e-5
    assert float((yb - (ref + shift.view(1, -1, 1, 1, 1))).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    if (cin, cout, stride) == (32, 8, 1):   # the default above is the depth-slice-pair form (knob bf16_dp); here one slice per MFMA
        emul_lib.call("mvs_set_tuning", b"bf16_dp", 0)
        try:
            with torch.no_grad():
                y1 = ops.conv3d_forward_bf16(x, wt, stride, transposed, scale=scale, shift=shift, skip=skip, relu=True)
        finally:
            emul_lib.call("mvs_set_tuning", b"bf16_dp", 1)
        assert float((y1.float() - exp).abs().max()) <= 2 ** -7 * float(exp.abs().max()) + 1e-5
        assert float((y1.float() - y.float()).abs().max()) <= 2 ** -7 * float(exp.abs().max())
    if transposed and cout == 8:   # the default above is the W-parity-merged form (knob tr2pw); here one MFMA per parity class
        emul_lib.call("mvs_set_tuning", b"tr2pw", 0)
        try:
            with torch.no_grad():
                y8 = ops.conv3d_forward_bf16(x, wt, stride, transposed, scale=scale, shift=shift, skip=skip, relu=True)
        finally:
            emul_lib.call("mvs_set_tuning", b"tr2pw", 1)
        assert float((y8.float() - exp).abs().max()) <= 2 ** -7 * float(exp.abs().max()) + 1e-5
        assert float((y8.float() - y.float()).abs().max()) <= 2 ** -7 * float(exp.abs().max())
    if cout == 1:   # the default above is the direct four-outputs-per-thread form (knob cout1_d4, bit 1); here the MFMA form
        emul_lib.call("mvs_set_tuning", b"cout1_d4", 0)
        try:
            with torch.no_grad():
                y4 = ops.conv3d_forward_bf16(x, wt, stride, transposed, scale=scale, shift=shift, skip=skip, relu=True)
                yb4 = ops.conv3d_forward_bf16(x, wt, stride, transposed, shift=shift, out_f32=True)
        finally:
            emul_lib.call("mvs_set_tuning", b"cout1_d4", 2)
        assert float((y4.float() - exp).abs().max()) <= 2 ** -7 * float(exp.abs().max()) + 1e-5
        assert float((yb4 - (ref + shift.view(1, -1, 1, 1, 1))).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


def test_plane_sweep_variance_bf16_volume(emul_lib):
    from mvs_amd import ops
    g = torch.Generator().manual_seed(3)
    for c, ns, per_pixel in ((32, 2, False), (16, 6, True), (32, 3, False)):
        b, d, h, w = 1, 5, 11, 19
        rot, trans = _cams(b, ns, h, w, g)
        ref = torch.randn(b, c, h, w, generator=g)
        srcs = [torch.randn(b, c, h, w, generator=g) for _ in range(ns)]
        depth = (450 + 30 * torch.rand(b, 1, h, w, generator=g) + 20.0 * torch.arange(d).view(1, d, 1, 1)) if per_pixel \
            else (430 + 35.0 * torch.arange(d)).unsqueeze(0)
        with torch.no_grad():
            v32 = ops.plane_sweep_variance(ref, srcs, rot, trans, depth)
            v16 = ops.plane_sweep_variance(ref, srcs, rot, trans, depth, out_dtype=torch.bfloat16)
        assert v16.dtype == torch.bfloat16 and v16.shape == v32.shape
        assert torch.equal(v16, v32.bfloat16())        # same arithmetic, rounded once at the store


@pytest.mark.skipif(os.environ.get("MVS_EMUL_FULL") != "1", reason="3.5 minutes of emulation; set MVS_EMUL_FULL=1 (the GPU version is test_gpu_parity.py::test_bf16_inference_path)")
def test_mvsnet_bf16_inference_vs_fp32(emul_lib):
    """End to end on a tiny case: eval-mode MVSNet with bf16 storage vs the fp32 path (the oracle of this path)."""
    from mvs_amd.jdacs.models.mvsnet import MVSNet
    torch.manual_seed(0)
    net = MVSNet(refine=False)
    with torch.no_grad():
        net.cost_regularization.prob.weight.mul_(50.0)
    imgs, proj, dv = R.synthetic_mvsnet_inputs(1, 3, 64, 96, 16, seed=1)
    from conftest import calibrate_batchnorm
    calibrate_batchnorm(net, imgs, proj, dv)    # running statistics := batch statistics (non-degenerate eval model)
    with torch.no_grad():
        o32 = net(imgs, proj, dv)
        net.storage_dtype = torch.bfloat16
        o16 = net(imgs, proj, dv)
    assert o16["depth"].dtype == torch.float32
    assert float(o32["depth"].std()) > 2 * float(dv[0, 1] - dv[0, 0])     # the model is not degenerate
    assert rel_l1(o16["depth"], o32["depth"]) < 5e-3
    net.train()
    with pytest.raises(RuntimeError, match="inference path"):
        net(imgs, proj, dv)


def test_geo_consistency_filter_golden(emul_lib):
    """the HIP geometric-consistency filter vs the fixture the reference's own functions produced."""
    import numpy as np
    from conftest import GOLDEN
    from mvs_amd.jdacs.fusion import geo_filter as GF
    z = np.load(os.path.join(GOLDEN, "g10_geo_filter.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    nsrc = z["depth_src"].shape[0]
    dref = t(z["depth_ref"])
    srcs = [t(z["depth_src"][v]) for v in range(nsrc)]
    for v in range(nsrc):
        mask, rep, xs, ys = GF.check_geometric_consistency(dref, z["K"][0], z["E"][0], srcs[v], z["K"][v + 1], z["E"][v + 1])
        assert float((mask.numpy() != z["mask%d" % (v + 1)]).mean()) < 1e-3     # a pixel exactly at a threshold may flip
        same = mask.numpy() == z["mask%d" % (v + 1)]
        assert np.allclose(rep.numpy()[same], z["reproj%d" % (v + 1)][same], rtol=1e-6, atol=1e-4)
        assert np.allclose(xs.numpy(), z["x_src%d" % (v + 1)], atol=1e-4) and np.allclose(ys.numpy(), z["y_src%d" % (v + 1)], atol=1e-4)
    r = GF.filter_depth_view(dref, t(z["conf_ref"]), z["K"][0], z["E"][0], srcs, list(z["K"][1:]), list(z["E"][1:]))
    assert float((r["geo_count"].numpy() != z["geo_count"]).mean()) < 2e-3
    ok = r["geo_count"].numpy() == z["geo_count"]
    assert np.allclose(r["depth_avg"].numpy()[ok], z["depth_avg"][ok], rtol=1e-6)
    assert float((r["final_mask"].numpy() != z["final_mask"]).mean()) < 2e-3


def test_filter_depth_scan_level_golden(emul_lib, tmp_path):
    """SURVEY 8(f)-4, the scan-level tail of filter_depth (eval.py:340-447: pair file, cameras, images, PFM maps -> masks + fused
    coloured point cloud) vs the fixture made by EXECUTING the reference's own function (tests/golden/make_golden_filter_depth.py)."""
    from conftest import run_filter_depth_golden
    run_filter_depth_golden(tmp_path, "cpu")


@pytest.mark.parametrize("cin,cout,hw", [(3, 64, (9, 34)), (64, 64, (5, 17)), (64, 32, (6, 12)), (32, 16, (8, 35)), (4, 32, (9, 21)), (32, 1, (8, 33))])
def test_conv2d_lrelu_block_and_wide_channels(emul_lib, cin, cout, hw):
    """the `conv` block of the CVP feature pyramid (Conv2d 3x3 + bias + LeakyReLU 0.1; widths up to 64:
    jdacs-ms/models/network.py:16-41) and RefineNet's channel counts (4 -> 32 -> 1: jdacs/models/mvsnet.py:77-92): forward,
    input, weight and bias gradients vs ATen."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(cin + 7 * cout)
    x = torch.randn(1 if cin == 64 else 2, cin, *hw, generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (0.5 / cin ** 0.5)
    b = torch.randn(cout, generator=g) * 0.3
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.leaky_relu(F.conv2d(xr, wr, br, padding=1), 0.1)
    xa, wa, ba = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = ops.Conv2dLReLUFn.apply(xa, wa, ba, 0.1)
    assert y.shape == yr.shape and float((y - yr).abs().max()) < 3e-4
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    y.backward(gy)
    assert float((xa.grad - xr.grad).abs().max()) < 5e-4
    assert float((wa.grad - wr.grad).abs().max()) < 1e-3 * max(1.0, float(wr.grad.abs().max()))
    assert float((ba.grad - br.grad).abs().max()) < 1e-3 * max(1.0, float(br.grad.abs().max()))


@pytest.mark.skipif(os.environ.get("MVS_EMUL_FULL") != "1", reason="3 minutes of emulation; set MVS_EMUL_FULL=1 (the per-layer cases above run by default)")
def test_feature_pyramid_through_hip_convs(emul_lib):
    """FeaturePyramid (3 levels) with every block through csrc/conv2d.hip == the stock-PyTorch path, values and gradients."""
    from mvs_amd.jdacs_ms.models.network import FeaturePyramid
    torch.manual_seed(0)
    fp = FeaturePyramid()
    img = torch.randn(1, 3, 24, 40)
    ref = fp(img, 3)
    sum(f.square().mean() for f in ref).backward()
    gref = {k: p.grad.clone() for k, p in fp.named_parameters()}
    fp.zero_grad()
    # the emulation serves CPU tensors: bypass the is_cuda gate of the module by calling the fused op the same way it does
    from mvs_amd import ops
    from mvs_amd.jdacs_ms.models.network import _PYRAMID_LAYERS
    import torch.nn.functional as Fn

    def trunk(x):
        x = x.contiguous(memory_format=torch.channels_last)
        for name, *_ in _PYRAMID_LAYERS:
            blk = getattr(fp, name)
            x = ops.Conv2dLReLUFn.apply(x, blk[0].weight, blk[0].bias, blk[1].negative_slope)
        return x
    levels, im = [trunk(img)], img
    for _ in range(2):
        im = Fn.interpolate(im, scale_factor=0.5, mode="bilinear", align_corners=None).detach()
        levels.append(trunk(im))
    for a, b in zip(levels, ref):
        assert float((a - b).abs().max()) < 1e-4
    sum(f.square().mean() for f in levels).backward()
    for k, p in fp.named_parameters():
        assert float((p.grad - gref[k]).abs().max()) < 2e-3 * max(1e-6, float(gref[k].abs().max())), k


def test_fusibile_fusion_kernel_vs_numpy_oracle(emul_lib):
    """the fusion kernel (csrc/fusibile.hip) on the host emulation vs oracle/fusibile_np.py (numpy restatement of
    fusibile.cu:138-277) on a synthetic 5-view scene: same points, normals and colours for every reference camera; a pixel may
    differ only where a consistency test sits exactly on its threshold (device vs numpy acos / division rounding)."""
    from mvs_amd.jdacs.fusion import depthfusion as DF
    from oracle import fusibile_np as FO
    Ps, nd, img, _, _ = FO.synthetic_scene(5, 20, 28, seed=3)
    cams_o = FO.fusibile_cameras(Ps)
    cams, f = DF.fusibile_cameras(Ps)
    assert abs(f - float(cams_o["f"])) < 1e-3 and float(abs(cams - cams_o["cams"]).max()) < 2e-3 * float(abs(cams_o["cams"]).max())
    lib = emul_lib
    nd_t, img_t = torch.from_numpy(nd).contiguous(), torch.from_numpy(img).contiguous()
    cams_t = torch.from_numpy(cams_o["cams"]).contiguous()
    subset = torch.arange(5, dtype=torch.int32)
    total = 0
    for ref in range(5):
        out = torch.empty((20, 28, 12), dtype=torch.float32)
        lib.call("mvs_fusibile_fuse", nd_t.data_ptr(), img_t.data_ptr(), cams_t.data_ptr(), subset.data_ptr(), 5, 5, 20, 28, ref,
                 float(cams_o["f"]), 0.25, float(np.float32(360.0) * np.float32(np.pi) / np.float32(180.0)), 2, 1, out.data_ptr(), None)
        exp, count = FO.fuse_view(nd, img, cams_o["cams"], list(range(5)), ref, cams_o["f"], 0.25, 2 * np.pi, 2, True)
        got = out.numpy()
        same_support = ((got[..., 0] != 0) == (exp[..., 0] != 0))
        assert same_support.mean() > 0.995, (ref, same_support.mean())
        both = same_support & (exp[..., 0] != 0)
        assert np.abs(got[both] - exp[both]).max() < 2e-3 * np.abs(exp[both]).max()
        total += int((exp[..., 0] != 0).sum())
    assert total > 5 * 20 * 28 * 0.3       # the scene is consistent: a good part of every view survives
    # bad arguments are rejected with a message, not a crash
    with pytest.raises(ValueError, match="reference camera"):
        lib.call("mvs_fusibile_fuse", nd_t.data_ptr(), None, cams_t.data_ptr(), subset.data_ptr(), 5, 5, 20, 28, 7, 1.0, 0.25, 1.0, 2, 0,
                 out.data_ptr(), None)


def test_fusion_chain_on_files_vs_oracle(emul_lib, tmp_path):
    """PFM -> probability_filter -> gipuma folder -> run_fusibile (the program's folder / camera / .dmb reading, the kernel on the
    emulation, compaction, .ply layout): the written final3d_model.ply equals the oracle's bytes for the same files' content."""
    from conftest import build_fusion_folders
    from mvs_amd.jdacs.fusion import depthfusion as DF
    from oracle import fusibile_np as FO
    point_folder, ins = build_fusion_folders(tmp_path, 4, 20, 28, seed=4)
    ply = DF.run_fusibile(point_folder, os.path.join(point_folder, "cams"), os.path.join(point_folder, "images"), 0.25, 2, 360.0,
                          device="cpu", timestamp="20260927-000000")
    assert ply.endswith(os.path.join("consistencyCheck-20260927-000000", "final3d_model.ply"))
    nthr = float(np.float32(360.0) * np.float32(np.pi) / np.float32(180.0))
    exp = FO.fuse_all(ins["nd"], ins["img"], ins["cams"]["cams"], ins["cams"]["f"], 0.25, nthr, 2)
    assert exp.shape[0] > 100
    assert open(ply, "rb").read() == FO.ply_bytes(exp)


_full = pytest.mark.skipif(os.environ.get("MVS_EMUL_FULL") != "1", reason="a non-default kernel: 30 s of emulation per case; set MVS_EMUL_FULL=1")


def _close(a, b, tol=2e-5):
    return float((a - b).abs().max()) <= tol * max(1e-3, float(b.abs().max()))


def test_fused_regulariser_node_on_a_small_program(emul_lib):
    """The one-node regulariser (ops.UNetRegulariserFn) on a three-block U-Net small enough for the default CPU suite -- stride-1
    conv, stride-2 conv, stride-2 transposed conv with the skip added after its ReLU, bias-only prob layer -- against the per-layer
    autograd graph built from the same modules: logits and BatchNorm buffers bit-identical, input / parameter gradients to fp32 rounding."""
    from mvs_amd import nn3d, ops
    torch.manual_seed(5)

    def build():
        torch.manual_seed(5)
        return torch.nn.ModuleList([nn3d.ConvBnReLU3D(8, 8, stride=1), nn3d.ConvBnReLU3D(8, 16, stride=2), nn3d.DeconvBnReLU3D(16, 8, stride=2),
                                    nn3d.ProbConv3d(8)]).train()
    x0 = torch.randn(1, 8, 4, 4, 16)
    gout = torch.randn(1, 1, 4, 4, 16)
    res = {}
    for fused in (True, False):
        m = build()
        x = x0.clone().requires_grad_(True)
        if fused:
            y = ops.unet_regulariser(x, [(m[0].conv, m[0].bn, False, 1, -1, -1), (m[1].conv, m[1].bn, False, 2, 0, -1),
                                         (m[2][0], m[2][1], True, 2, 1, 0)], m[3])
        else:
            y0 = m[0](x)
            y = m[3](m[2](m[1](y0), skip=y0))
        y.backward(gout)
        res[fused] = (y.detach(), x.grad, [p.grad for p in m.parameters()], [b.clone() for b in m.buffers()])
    # forward: the same kernels in the same order -> bit-identical; backward: the fused node sums the BatchNorm backward statistics
    # in the input-gradient epilogues (per tile) where the per-layer graph runs a reduction pass (per stride): fp32 rounding apart
    assert torch.equal(res[True][0], res[False][0]) and _close(res[True][1], res[False][1])
    for a, b in zip(res[True][2], res[False][2]):
        assert _close(a, b)
    for a, b in zip(res[True][3], res[False][3]):
        assert torch.equal(a, b)


@pytest.mark.skipif(os.environ.get("MVS_EMUL_FULL") != "1", reason="3.5 minutes of emulation (64-channel layers); set MVS_EMUL_FULL=1")
def test_fused_regulariser_node_equals_per_layer_graph(emul_lib):
    """ops.UNetRegulariserFn (the whole regulariser as one autograd node: skip gradients summed in the dgrad epilogue through
    mvs_conv3d_dgrad_acc / mvs_convT3d_dgrad_acc, one node on the tape) vs the per-layer graph on the CVP regulariser (stride-1 and
    stride-2 transposed blocks with skips, shared code with MVSNet's): logits and the BatchNorm buffers bit-identical, input gradient
    and EVERY parameter gradient equal to fp32 rounding (the BatchNorm backward statistics are summed per tile in the input-gradient
    epilogues here, per stride by a reduction pass there); a frozen parameter gets no gradient.  (The goldens run through the fused node by default:
    test_costregnet_golden / test_costregnet_cvp_golden here with MVS_EMUL_FULL=1, and on the GPU.)"""
    from mvs_amd import ops
    from mvs_amd.jdacs_ms.models.network import CostRegNet
    torch.manual_seed(3)
    ref = CostRegNet().train()
    x0 = torch.randn(1, 16, 2, 4, 16)
    gout = torch.randn(1, 2, 4, 16)
    res = {}
    for fused in (True, False):
        net = CostRegNet().train()
        net.load_state_dict(ref.state_dict())
        net.conv2.bn.bias.requires_grad_(False)
        x = x0.clone().requires_grad_(True)
        old = ops.FUSED_REGULARISER
        ops.FUSED_REGULARISER = fused
        try:
            y = net(x)
            y.backward(gout)
        finally:
            ops.FUSED_REGULARISER = old
        res[fused] = (y.detach(), x.grad, {k: p.grad for k, p in net.named_parameters()}, {k: v.clone() for k, v in net.state_dict().items()})
    assert torch.equal(res[True][0], res[False][0]) and _close(res[True][1], res[False][1])
    for k in res[True][2]:
        a, b = res[True][2][k], res[False][2][k]
        assert (a is None) == (b is None), k
        if a is not None:
            assert _close(a, b), k
    assert res[True][2]["conv2.bn.bias"] is None
    for k in res[True][3]:
        assert torch.equal(res[True][3][k], res[False][3][k]), k
    assert int(res[True][3]["conv0.bn.num_batches_tracked"]) == 1


@pytest.mark.parametrize("ns,hw,d", [(2, (13, 21), 9), pytest.param(4, (10, 19), 20, marks=_full), (6, (7, 19), 11)])
def test_plane_sweep_xcd_compact_order_and_merged_regather(emul_lib, ns, hw, d):
    """Knob sweep_xcd (workgroup ids re-dealt so that each XCD owns a contiguous run of tiles: a bijection for grid sizes that are
    not multiples of 8, here with 2 batch entries and several depth slabs) and fwd_dl=2 (merged re-gather phase): forward (fp32 and
    bf16 volume) and backward bit-identical to the default order / loop."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(77 + ns)
    b, c = 2, 32
    h, w = hw
    rot, trans = _cams(b, ns, h, w, g)
    trans = trans * torch.tensor([3.0, -2.0, 1.0])
    ref = torch.randn(b, c, h, w, generator=g)
    srcs = [torch.randn(b, c, h, w, generator=g) for _ in range(ns)]
    depth = (430 + 21.0 * torch.arange(d)).unsqueeze(0).repeat(b, 1)
    gvar = torch.randn(b, c, d, h, w, generator=g)
    res = {}
    try:
        for key, knobs in (("base", {}), ("xcd", {b"sweep_xcd": 1}), ("both", {b"sweep_xcd": 1, b"fwd_dl": 2})):
            emul_lib.call("mvs_set_tuning", b"sweep_xcd", knobs.get(b"sweep_xcd", 0))
            emul_lib.call("mvs_set_tuning", b"fwd_dl", knobs.get(b"fwd_dl", 2))
            emul_lib.call("mvs_set_tuning", b"dslab", 4)        # several slabs per tile: the slab index goes through the re-deal too
            emul_lib.call("mvs_set_tuning", b"bwd_dslab", 4)
            fr = [t.clone().requires_grad_(True) for t in [ref] + srcs]
            var = ops.plane_sweep_variance(fr[0], fr[1:], rot, trans, depth)
            grads = torch.autograd.grad(var, fr, gvar)
            with torch.no_grad():
                v16 = ops.plane_sweep_variance(ref, srcs, rot, trans, depth, out_dtype=torch.bfloat16)
            res[key] = (var.detach(), v16, grads)
    finally:
        for k, v in ((b"sweep_xcd", 0), (b"fwd_dl", 2), (b"dslab", 0), (b"bwd_dslab", 0)):
            emul_lib.call("mvs_set_tuning", k, v)
    for key in ("xcd", "both"):
        assert torch.equal(res[key][0], res["base"][0]), key
        assert torch.equal(res[key][1], res["base"][1]), key
        for ga, gb in zip(res[key][2], res["base"][2]):
            # the backward's atomics land in a different order under sweep_xcd: equal up to fp32 summation order
            assert float((ga - gb).abs().max()) <= 1e-4 * max(1.0, float(gb.abs().max())), key


@pytest.mark.parametrize("cin,cout,ks,stride,hw", [(3, 8, 3, 1, (9, 21)), (8, 16, 5, 2, (11, 22)), (16, 32, 5, 2, (9, 14)), (32, 32, 3, 1, (6, 13))])
def test_conv2d_folded_batchnorm_relu_eval(emul_lib, cin, cout, ks, stride, hw):
    """Inference form of ConvBnReLU (module.py:15-22 in eval mode): BatchNorm's running statistics folded into weights + bias
    (module._fold_bn), ReLU as LeakyReLU(0) in the same csrc/conv2d.hip pass -- vs conv2d -> batch_norm(eval) -> relu of ATen."""
    from mvs_amd import ops
    from mvs_amd.jdacs.models.module import ConvBnReLU, _fold_bn
    torch.manual_seed(cin + cout)
    m = ConvBnReLU(cin, cout, ks, stride, ks // 2).eval()
    with torch.no_grad():
        m.bn.running_mean.normal_(0, 0.3); m.bn.running_var.uniform_(0.5, 2.0)
        m.bn.weight.uniform_(0.5, 1.5); m.bn.bias.normal_(0, 0.2)
        x = torch.randn(2, cin, *hw).contiguous(memory_format=torch.channels_last)
        exp = F.relu(m.bn(m.conv(x)))
        w, b = _fold_bn(m.conv, m.bn)
        got = ops.conv2d_forward(x, w, b, stride, negative_slope=0.0)
    assert got.shape == exp.shape
    assert float((got - exp).abs().max()) < 2e-5 * max(1.0, float(exp.abs().max()))
    assert float(got.min()) >= 0.0


@pytest.mark.parametrize("cin,cout,ks,stride,hw", [(3, 8, 3, 1, (11, 37)), (8, 8, 3, 1, (9, 35)), (8, 16, 5, 2, (12, 66)), (16, 16, 3, 1, (9, 33)),
                                                    (16, 32, 5, 2, (11, 35)), (32, 32, 3, 1, (6, 18))])
def test_conv2d_forward_with_batchnorm_partial_sums(emul_lib, cin, cout, ks, stride, hw):
    """mvs_conv2d_fwd_stats (the convolution of a training-mode ConvBnReLU, module.py:15-22, with BatchNorm's statistics pass folded
    into its epilogue) + mvs_bn_relu_fwd_slots: the convolution equals the plain kernel bit for bit, the slot rows sum to
    the per-image channel sums (ragged images: tiles overhang), and BatchNorm + ReLU from the rows equals BatchNorm + ReLU with its
    own statistics pass, running statistics included (3 images = 3 statistics groups)."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(cin * 5 + cout + ks)
    n = 3
    x = torch.randn(n, cin, *hw, generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, ks, ks, generator=g) * 0.2
    y0 = ops.conv2d_forward(x, w, None, stride)
    y, parts = ops.conv2d_forward(x, w, None, stride, want_stats=True, groups=n)
    assert torch.equal(y, y0)
    assert parts.dtype == torch.float64 and parts.shape[0] == n and tuple(parts.shape[2:]) == (2, cout)
    # the same with the weight image written ahead of time by the batch packer, from a channels-last AND a contiguous parameter
    w_cl = w.contiguous(memory_format=torch.channels_last)
    ws_cl, ws_ct = ops.pack_conv2d_weights([w_cl, w], [stride, stride], x)
    for wt, ws_ in ((w_cl, ws_cl), (w, ws_ct)):
        y2, parts2 = ops.conv2d_forward(x, wt, None, stride, want_stats=True, groups=n, packed_ws=ws_)
        assert torch.equal(y2, y0) and torch.equal(parts2.sum(1), parts.sum(1))
    per_img = parts.sum(1).float()
    assert torch.allclose(per_img[:, 0], y.sum(dim=(2, 3)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(per_img[:, 1], (y * y).sum(dim=(2, 3)), rtol=1e-4, atol=1e-3)
    gamma, beta = 0.5 + torch.rand(cout, generator=g), torch.randn(cout, generator=g) * 0.2
    rm_a, rv_a, rm_b, rv_b = torch.zeros(cout), torch.ones(cout), torch.zeros(cout), torch.ones(cout)
    za = ops.BnReLUFn.apply(y, gamma, beta, rm_a, rv_a, True, 1e-5, 0.1, n)
    zb = ops.BnReLUFn.apply(y, gamma, beta, rm_b, rv_b, True, 1e-5, 0.1, n, parts)
    assert float((za - zb).abs().max()) < 1e-5 * max(1.0, float(za.abs().max()))
    assert torch.allclose(rm_a, rm_b, rtol=1e-5, atol=1e-6) and torch.allclose(rv_a, rv_b, rtol=1e-5, atol=1e-6)
    ref = torch.cat([F.relu(F.batch_norm(y[i:i + 1], None, None, gamma, beta, True, 0.1, 1e-5)) for i in range(n)], 0)
    assert float((zb - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape,empty", [((1, 7, 9), False), ((2, 33, 41), False), ((1, 5, 5), True)])
def test_masked_smooth_l1_loss(emul_lib, shape, empty):
    """mvs_masked_smooth_l1_fwd / _bwd (mvsnet_loss, mvsnet.py:164-166) vs the reference's formulation
    F.smooth_l1_loss(est[mask], gt[mask]): value, gradient, both branches of the loss, an empty mask (nan like the reference)."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    est = (torch.randn(shape, generator=g) * 2).requires_grad_(True)
    gt = torch.randn(shape, generator=g)
    mask = torch.zeros(shape) if empty else (torch.rand(shape, generator=g) > 0.3).float()
    loss = ops.MaskedSmoothL1.apply(est, gt, mask)
    est_r = est.detach().clone().requires_grad_(True)
    ref = F.smooth_l1_loss(est_r[mask > 0.5], gt[mask > 0.5], reduction="mean")
    if empty:
        assert torch.isnan(loss) and torch.isnan(ref)
        return
    assert abs(float(loss) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    (loss * 3.0).backward()
    (ref * 3.0).backward()
    assert float((est.grad - est_r.grad).abs().max()) < 1e-6


def test_conv3d_wgrad_wide_reduction_of_many_partial_images(emul_lib):
    """More than 32 persistent workgroups -> conv_wgrad_reduce_wide_kernel (one launch: 16 slices per output element, fixed order)
    instead of the serial one-thread-per-element reduce: weight gradient vs ATen on a volume of 36 tiles, ragged in W."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(17)
    x = torch.randn(1, 8, 12, 16, 41, generator=g)
    w = torch.randn(16, 8, 3, 3, 3, generator=g) * 0.2
    xr, wr = x.clone(), w.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    gw = ops.conv3d_wgrad(x, gy, tuple(w.shape), 1, False)
    assert float((gw - wr.grad).abs().max()) < 1e-3 * max(1.0, float(wr.grad.abs().max()))


@pytest.mark.parametrize("cin,cout,stride,transposed,dims", [(8, 16, 2, False, (4, 8, 20)), (16, 16, 1, False, (3, 5, 18)), (16, 8, 2, True, (2, 4, 10)),
                                                             (8, 8, 1, False, (4, 3, 17)), (32, 32, 1, False, (2, 3, 17))])
def test_conv3d_wgrad_quarter_size_tiles(emul_lib, cin, cout, stride, transposed, dims):
    """Knob wgrad_small: the generic weight-gradient kernel on the *_SMALL geometries (3x9x33- instead of 5x9x33-voxel halos for the
    stride-2 L0 layers: three workgroups per CU instead of one) -- same result as the full-size tiles, and vs ATen."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(cin + cout + stride)
    x = torch.randn(2, cin, *dims, generator=g)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = torch.randn(wshape, generator=g) * 0.2
    wr = w.clone().requires_grad_(True)
    yr = F.conv_transpose3d(x, wr, stride=stride, padding=1, output_padding=stride - 1) if transposed else F.conv3d(x, wr, stride=stride, padding=1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    outs = {}
    try:
        for mode in (0, 3):
            emul_lib.call("mvs_set_tuning", b"wgrad_small", mode)
            outs[mode] = ops.conv3d_wgrad(x, gy, wshape, stride, transposed)
    finally:
        emul_lib.call("mvs_set_tuning", b"wgrad_small", 0)
    scale = max(1.0, float(wr.grad.abs().max()))
    assert float((outs[3] - wr.grad).abs().max()) < 1e-3 * scale
    assert float((outs[3] - outs[0]).abs().max()) < 2e-4 * scale


@pytest.mark.parametrize("cin,cout,ks,stride,hw", [(3, 8, 3, 1, (40, 130)), (8, 16, 5, 2, (44, 132)), (16, 32, 3, 1, (24, 200))])
def test_conv2d_wgrad_wide_reduction(emul_lib, cin, cout, ks, stride, hw):
    """More than 16 persistent workgroups -> conv2d_wgrad_reduce_wide_kernel (16 slices per element of the partial-image layout,
    channel padding skipped): weight gradient vs ATen on images of 20+ tiles."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(1, cin, *hw, generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(cout, cin, ks, ks, requires_grad=True)
    y = F.conv2d(x, w, stride=stride, padding=ks // 2)
    gy = torch.randn(y.shape, generator=g).contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    gw = ops.conv2d_wgrad(x, gy, tuple(w.shape), stride)
    assert float((gw - w.grad).abs().max()) < 1e-3 * max(1.0, float(w.grad.abs().max()))


def _stock_extractor(net, x, groups):
    """the reference's extractor on stock torch modules: every view through the blocks on its own (mvsnet.py:115), train mode"""
    from mvs_amd.jdacs.models.mvsnet import _FEATURE_LAYERS
    outs = []
    for v in x.chunk(groups, 0):
        for name, *_ in _FEATURE_LAYERS:
            m = getattr(net, name)
            v = F.relu(m.bn(m.conv(v)))
        outs.append(net.feature(v))
    return torch.cat(outs, 0)


# (the one-launch weight gradients without the consumer-side BatchNorm -- MVS_FEATURE_FUSED_APPLY=0 -- are covered kernel by kernel in
#  test_conv2d_weight_gradients_of_all_layers_in_one_launch and end to end on the GPU)
@pytest.mark.parametrize("wgrad_batch,fused,dgrad_bn", [(False, False, False), (True, True, False), pytest.param(True, True, True, marks=_full)],
                         ids=["library_wgrad", "consumer_side_batchnorm", "consumer_side_batchnorm+dgrad_statistics"])
def test_training_extractor_as_one_autograd_node(emul_lib, wgrad_batch, fused, dgrad_bn):
    """ops.FeatureExtractorFn (FeatureNet in training as ONE autograd node: mvsnet.py:17-34 + module.py:15-22 for every block) on the
    emulated kernels vs the stock modules applied view by view: output, input gradient, every parameter gradient and the
    running statistics, 2 views of 12x40 (ragged tiles), with the one-launch weight gradients, with the library's, and with
    BatchNorm + ReLU of block i applied inside block i+1's convolution and weight gradient (FEATURE_FUSED_APPLY: mvs_bn_finalize_slots,
    mvs_conv2d_fwd_stats_xf, mvs_conv2d_wgrad_batch_xf -- every instantiation of both kernels is on this path), and with the
    BatchNorm backward statistics of the block below a conv2d.hip input gradient summed in its epilogue (FEATURE_DGRAD_BNSTATS:
    mvs_conv2d_dgrad_bnstats, pixel-pair and plain kernels)."""
    import copy
    from mvs_amd import ops
    from mvs_amd.jdacs.models.mvsnet import FeatureNet, _FEATURE_LAYERS
    torch.manual_seed(3)
    ref = FeatureNet().train()
    net = copy.deepcopy(ref)
    groups = 2
    x = torch.randn(groups, 3, 12, 40)
    xr = x.clone().requires_grad_(True)
    xa = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yr = _stock_extractor(ref, xr, groups)
    gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(4))
    yr.backward(gy)
    blocks = [getattr(net, name) for name, *_ in _FEATURE_LAYERS]
    cfg, params = [], []
    for m in blocks:
        cfg.append((m.conv.stride[0], m.conv.padding[0], float(m.bn.eps), float(m.bn.momentum), m._hip_dgrad()))
        params += [m.conv.weight, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var]
    params += [net.feature.weight, net.feature.bias]
    old = ops.FEATURE_WGRAD_BATCH, ops.FEATURE_FUSED_APPLY, ops.FEATURE_DGRAD_BNSTATS
    ops.FEATURE_WGRAD_BATCH, ops.FEATURE_FUSED_APPLY, ops.FEATURE_DGRAD_BNSTATS = wgrad_batch, fused, dgrad_bn
    try:
        with ops.slot_scope():
            ya = ops.FeatureExtractorFn.apply(xa, groups, tuple(cfg), *params)
        assert ya.grad_fn.fused == fused
        ya.backward(gy.contiguous(memory_format=torch.channels_last))
    finally:
        ops.FEATURE_WGRAD_BATCH, ops.FEATURE_FUSED_APPLY, ops.FEATURE_DGRAD_BNSTATS = old
    assert float((ya - yr).abs().max()) < 1e-4 * max(1.0, float(yr.abs().max()))
    assert float((xa.grad - xr.grad).abs().max()) < 2e-3 * max(1e-6, float(xr.grad.abs().max()))
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert float((p.grad - q.grad).abs().max()) < 2e-3 * max(1e-6, float(q.grad.abs().max())), k
    for (k, u), (_, v) in zip(net.named_buffers(), ref.named_buffers()):
        if u.dtype.is_floating_point:       # (num_batches_tracked is the module wrapper's business: FeatureNet.forward)
            assert torch.allclose(u, v, rtol=1e-4, atol=1e-6), k


@pytest.mark.parametrize("channels_last_weights", [pytest.param(False, marks=_full), True], ids=["contiguous_weights", "channels_last_weights"])
def test_training_extractor_one_c_call_per_pass_equals_per_layer_calls(emul_lib, channels_last_weights):
    """mvs_feature_fwd / mvs_feature_bwd (csrc/feature_pass.cpp: the node's forward / backward pass as ONE C call each over pointer
    tables into two arenas) against the same node issuing the per-layer C calls from Python (ops.FEATURE_C_ENTRY = False): the same
    kernels in the same order => output, input gradient, every parameter gradient and the running statistics BIT-IDENTICAL.  With
    channels-last parameters (module.to(memory_format=torch.channels_last): what bench.py trains) the C entry reads them in place
    (mvs_conv2d_fwd_wl / mvs_conv2d_dgrad_wl) and writes the weight gradients in the parameters' layout.  A three-block chain (3x3, 5x5
    stride 2, 3x3 + the closing convolution: every branch of the two entry points; the emulated one-launch weight gradient of the full
    FeatureNet alone takes 18 s) -- the full FeatureNet through the C entry against the STOCK modules is
    test_training_extractor_as_one_autograd_node[consumer_side_batchnorm].  Replaces the module calls of FeatureNet.forward
    (/root/reference/jdacs/models/mvsnet.py:17-34)."""
    import copy
    from mvs_amd import ops
    from mvs_amd.jdacs.models.module import ConvBnReLU
    torch.manual_seed(7)

    class Chain(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.b0, self.b1, self.b2 = ConvBnReLU(3, 8, 3, 1, 1), ConvBnReLU(8, 16, 5, 2, 2), ConvBnReLU(16, 16, 3, 1, 1)
            self.feature = torch.nn.Conv2d(16, 16, 3, 1, 1)
    ref = Chain().train()
    if channels_last_weights:
        ref = ref.to(memory_format=torch.channels_last)
    groups = 2
    x = torch.randn(groups, 3, 10, 36).contiguous(memory_format=torch.channels_last)
    res = {}
    for c_entry in (True, False):
        net = copy.deepcopy(ref)
        cfg, params = [], []
        for m in (net.b0, net.b1, net.b2):
            cfg.append((m.conv.stride[0], m.conv.padding[0], float(m.bn.eps), float(m.bn.momentum), m._hip_dgrad()))
            params += [m.conv.weight, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var]
        params += [net.feature.weight, net.feature.bias]
        xa = x.clone().requires_grad_(True)
        old = ops.FEATURE_C_ENTRY
        ops.FEATURE_C_ENTRY = c_entry
        try:
            with ops.slot_scope():
                ya = ops.FeatureExtractorFn.apply(xa, groups, tuple(cfg), *params)
            assert ya.grad_fn.c_entry == c_entry and ya.grad_fn.fused
            gy = torch.randn(ya.shape, generator=torch.Generator().manual_seed(4)).contiguous(memory_format=torch.channels_last)
            ya.backward(gy, retain_graph=c_entry)
            g1 = {k: p.grad.clone() for k, p in net.named_parameters()}
            gx1 = xa.grad.clone()
            g2 = None
            if c_entry:                                        # a second backward through the same graph: fresh statistic accumulators
                for p_ in net.parameters():
                    p_.grad = None
                ya.backward(gy)
                g2 = {k: p.grad.clone() for k, p in net.named_parameters()}
        finally:
            ops.FEATURE_C_ENTRY = old
        res[c_entry] = (ya.detach(), gx1, g1, {k: v.clone() for k, v in net.state_dict().items()}, g2)
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert a[2][k].stride() == b[2][k].stride(), k             # the parameter's own memory layout
        assert torch.equal(a[2][k], b[2][k]), k
        assert torch.equal(a[4][k], a[2][k]), k                    # the second backward reproduces the first
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k


def test_training_extractor_with_a_frozen_weight_keeps_its_activations(emul_lib):
    """A convolution weight that needs no gradient (fine-tuning with a frozen layer): the node does not take the consumer-side
    BatchNorm path (whose backward has no normalised activations to hand to the library's per-layer weight gradients) and the
    other parameters still get the stock gradients."""
    import copy
    from mvs_amd import ops
    from mvs_amd.jdacs.models.mvsnet import FeatureNet, _FEATURE_LAYERS
    torch.manual_seed(5)
    ref = FeatureNet().train()
    net = copy.deepcopy(ref)
    for m in (ref, net):
        m.conv3.conv.weight.requires_grad_(False)
    x = torch.randn(1, 3, 8, 36)
    yr = _stock_extractor(ref, x, 1)
    gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(6))
    yr.backward(gy)
    blocks = [getattr(net, name) for name, *_ in _FEATURE_LAYERS]
    cfg, params = [], []
    for m in blocks:
        cfg.append((m.conv.stride[0], m.conv.padding[0], float(m.bn.eps), float(m.bn.momentum), m._hip_dgrad()))
        params += [m.conv.weight, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var]
    params += [net.feature.weight, net.feature.bias]
    assert ops.FEATURE_FUSED_APPLY and ops.FEATURE_WGRAD_BATCH       # the defaults
    with ops.slot_scope():
        ya = ops.FeatureExtractorFn.apply(x.contiguous(memory_format=torch.channels_last), 1, tuple(cfg), *params)
    assert not ya.grad_fn.fused
    ya.backward(gy.contiguous(memory_format=torch.channels_last))
    assert net.conv3.conv.weight.grad is None
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if q.grad is not None:
            assert float((p.grad - q.grad).abs().max()) < 2e-3 * max(1e-6, float(q.grad.abs().max())), k


WGRAD_BATCH_LAYERS = [  # (Cin, Cout, ks, stride, weight channels-last): the six instantiations of conv2d_wgrad_batch_kernel
    (3, 8, 3, 1, False), (8, 8, 3, 1, True), (8, 16, 5, 2, False), (16, 16, 3, 1, True), (16, 32, 5, 2, True), (32, 32, 3, 1, False)]


@pytest.mark.parametrize("budget,hw", [(6, (9, 70)), (40, (13, 37))])
def test_conv2d_weight_gradients_of_all_layers_in_one_launch(emul_lib, budget, hw):
    """mvs_conv2d_wgrad_batch: FeatureNet's layer shapes (jdacs/models/mvsnet.py:21-32) as ONE launch + one reduction launch vs
    ATen's weight gradients; ragged tiles in both directions, two images, several tiles per workgroup (budget 6: one workgroup per
    layer walks all its tiles) and one tile per workgroup, contiguous and channels-last parameter layouts."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(budget)
    xs, gys, ws, refs, strides = [], [], [], [], []
    for cin, cout, ks, stride, wcl in WGRAD_BATCH_LAYERS:
        x = torch.randn(2, cin, *hw, generator=g).contiguous(memory_format=torch.channels_last)
        w = torch.zeros(cout, cin, ks, ks, requires_grad=True)
        y = F.conv2d(x, w, stride=stride, padding=ks // 2)
        gy = torch.randn(y.shape, generator=g).contiguous(memory_format=torch.channels_last)
        y.backward(gy)
        xs.append(x); gys.append(gy); refs.append(w.grad); strides.append(stride)
        ws.append(w.detach().contiguous(memory_format=torch.channels_last) if wcl else w.detach())
    assert ops.conv2d_wgrad_batch_serves(xs, ws, strides)
    emul_lib.call("mvs_set_tuning", b"wgrad2d_batch", budget)
    ops._WGRAD_BATCH_PLANS.clear()
    try:
        gws = ops.conv2d_wgrad_batch(xs, gys, ws, strides)
    finally:
        emul_lib.call("mvs_set_tuning", b"wgrad2d_batch", 2048)
        ops._WGRAD_BATCH_PLANS.clear()
    for gw, ref, w, cfg in zip(gws, refs, ws, WGRAD_BATCH_LAYERS):
        assert gw.shape == ref.shape and gw.stride() == w.stride(), cfg
        assert float((gw - ref).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max())), cfg
    # a layer without an instantiation is reported, not computed
    assert not ops.conv2d_wgrad_batch_serves([torch.zeros(1, 12, 8, 8)], [torch.zeros(8, 12, 3, 3)], [1])
    with pytest.raises(ValueError):
        ops.conv2d_wgrad_batch([torch.zeros(1, 12, 8, 8)], [torch.zeros(1, 8, 8, 8)], [torch.zeros(8, 12, 3, 3)], [1])


@pytest.mark.parametrize("dims,xcd", [((4, 4, 18), 1), ((5, 6, 16), 0), ((3, 9, 33), 1)])
def test_conv_cout8_weight_gradient_two_chunks_per_workgroup(emul_lib, dims, xcd):
    """Knob wgrad8_nch = 2 (conv0, mvsnet.py:40: 32 -> 8): one workgroup stages the X halo of BOTH 16-channel chunks and the output
    gradient tile once -- the same weight gradient as the one-chunk form, and vs ATen; ragged tiles, both tile orders."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(sum(dims))
    x = torch.randn(2, 32, *dims, generator=g)
    w = torch.zeros(8, 32, 3, 3, 3, requires_grad=True)
    y = F.conv3d(x, w, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    outs = {}
    emul_lib.call("mvs_set_tuning", b"xcd", xcd)
    try:
        for nch in (1, 2):
            emul_lib.call("mvs_set_tuning", b"wgrad8_nch", nch)
            outs[nch] = ops.conv3d_wgrad(x, gy, tuple(w.shape), 1, False)
    finally:
        emul_lib.call("mvs_set_tuning", b"wgrad8_nch", 2)
        emul_lib.call("mvs_set_tuning", b"xcd", 1)
    scale = max(1.0, float(w.grad.abs().max()))
    assert float((outs[2] - w.grad).abs().max()) < 1e-3 * scale
    assert float((outs[2] - outs[1]).abs().max()) < 2e-4 * scale


# the persistent LDS-DMA implicit GEMM (csrc/conv3d_pers.hip): (cin, cout, stride, transposed, op, dims) -- op "fwd" / "dgrad"
PERS_CASES = [
    (16, 16, 1, False, "fwd", (5, 6, 21)),      # conv2: stride-1, 16-channel chunk, ragged 4 x 4 x 16 tiles in all directions
    (16, 16, 1, False, "dgrad", (4, 6, 20)),    # its input gradient (flipped taps) with summand + BatchNorm backward statistics
    (8, 16, 2, False, "fwd", (6, 10, 36)),      # conv1: stride 2, 8 channels (two taps per k-step)
    (16, 8, 2, True, "dgrad", (3, 5, 18)),      # conv11's input gradient: the same geometry through mvs_convT3d_dgrad
    (32, 8, 1, False, "dgrad", (4, 7, 19)),     # conv0's input gradient: 8 -> 32 as two Cout tiles per workgroup
    (16, 8, 2, True, "fwd", (5, 6, 19)),        # conv11: transposed stride 2, 16 -> 8 as W-parity-merged GEMMs (four classes per tile)
    (8, 16, 2, False, "dgrad", (10, 12, 38)),   # conv1's input gradient: the same geometry with summand + backward statistics
]


@pytest.mark.parametrize("groups", [3, 0], ids=["three_persistent_workgroups", "one_tile_per_workgroup"])
@pytest.mark.parametrize("cin,cout,stride,transposed,op,dims", PERS_CASES)
def test_conv3d_persistent_lds_dma_kernel_equals_one_tile_kernel(emul_lib, cin, cout, stride, transposed, op, dims, groups):
    """conv_pers_kernel walks several tiles per workgroup through a double-buffered LDS-DMA halo with the weight image resident in
    LDS; outputs must equal the one-tile kernel's BIT FOR BIT (same packed image, same MFMA order over k), the statistic slots to
    rounding (another summation order), with the epilogue's summand and backward statistics, at volume borders and ragged tiles."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(cin + 5 * cout + stride)
    b = 2
    x = torch.randn(b, cin, *dims, generator=g)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = torch.randn(wshape, generator=g) * 0.2

    def run():
        if op == "fwd":
            y, slots = ops.conv3d_forward(x, w, stride, transposed, want_stats=True)
            return y, slots.clone()
        yshape = (F.conv_transpose3d(x, w, stride=stride, padding=1, output_padding=stride - 1) if transposed
                  else F.conv3d(x, w, stride=stride, padding=1)).shape
        gy = torch.randn(yshape, generator=torch.Generator().manual_seed(7))
        raw = torch.randn(x.shape, generator=torch.Generator().manual_seed(8))
        add = torch.randn(x.shape, generator=torch.Generator().manual_seed(9))
        mean, var = raw.mean(dim=(0, 2, 3, 4)), raw.var(dim=(0, 2, 3, 4), unbiased=False)
        invstd = torch.rsqrt(var + 1e-5)
        stats = torch.stack([mean, invstd, 0.7 * invstd, 0.1 - mean * 0.7 * invstd]).contiguous()
        slots = torch.zeros((8, 2, cin), dtype=torch.float64)
        gx = ops.conv3d_dgrad(gy, w, tuple(x.shape), stride, transposed, add=add, bn=(raw, stats, slots))
        return gx, slots

    emul_lib.call("mvs_set_tuning", b"conv_pers", 0)
    try:
        ref, ref_slots = run()
        emul_lib.call("mvs_set_tuning", b"conv_pers", 1)
        emul_lib.call("mvs_set_tuning", b"conv_pers_min", 0)
        emul_lib.call("mvs_set_tuning", b"conv_pers_groups", groups)
        got, got_slots = run()
    finally:
        emul_lib.call("mvs_set_tuning", b"conv_pers", 1)
        emul_lib.call("mvs_set_tuning", b"conv_pers_min", 1024)
        emul_lib.call("mvs_set_tuning", b"conv_pers_groups", 0)
    assert torch.equal(got, ref), float((got - ref).abs().max())
    s_ref, s_got = ref_slots.sum(-3), got_slots.sum(-3)
    assert torch.allclose(s_got, s_ref, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("groups", [3, 0], ids=["three_persistent_workgroups", "one_tile_per_workgroup"])
@pytest.mark.parametrize("cin,cout,stride,transposed,dims", [
    (16, 16, 1, False, (5, 6, 21)),     # conv2: stride 1, 16 X channels, ragged tiles
    (8, 16, 2, False, (6, 10, 36)),     # conv1: stride 2, 8 X channels (tap pairs), X = the layer input
    (16, 8, 2, True, (3, 5, 18)),       # conv11 (transposed): X = the OUTPUT gradient (8 channels, fine grid), G = the input
    (8, 8, 2, False, (4, 8, 20)),       # 8 gradient channels: half of the G tile's 16-float rows stays zero
])
def test_conv3d_persistent_weight_gradient(emul_lib, cin, cout, stride, transposed, dims, groups):
    """conv_wgrad_pers_kernel (double-buffered LDS-DMA tiles, eight waves, K split over two wave groups) against autograd and
    against the register-staged kernel it replaces for these layers."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(3 * cin + cout)
    x = torch.randn(2, cin, *dims, generator=g)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = (torch.randn(wshape, generator=g) * 0.2).requires_grad_(True)
    y = F.conv_transpose3d(x, w, stride=stride, padding=1, output_padding=stride - 1) if transposed else F.conv3d(x, w, stride=stride, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    emul_lib.call("mvs_set_tuning", b"wgrad_pers", 0)
    try:
        old = ops.conv3d_wgrad(x, gy, wshape, stride, transposed)
        emul_lib.call("mvs_set_tuning", b"wgrad_pers", 1)
        emul_lib.call("mvs_set_tuning", b"conv_pers_min", 0)
        emul_lib.call("mvs_set_tuning", b"conv_pers_groups", groups)
        new = ops.conv3d_wgrad(x, gy, wshape, stride, transposed)
    finally:
        emul_lib.call("mvs_set_tuning", b"wgrad_pers", 1)
        emul_lib.call("mvs_set_tuning", b"conv_pers_min", 1024)
        emul_lib.call("mvs_set_tuning", b"conv_pers_groups", 0)
    scale = max(1.0, float(w.grad.abs().max()))
    assert float((new - w.grad).abs().max()) < 1e-3 * scale
    assert float((new - old).abs().max()) < 2e-4 * scale
    assert not torch.equal(new, torch.zeros_like(new))


@pytest.mark.parametrize("waves,b,dims,groups", [
    (2, 2, (3, 5, 17), 3),         # sixteen waves; ragged tiles, batch 2, three persistent workgroups walk eight tiles
    pytest.param(1, 2, (3, 5, 17), 192, marks=_full),   # eight waves (knob wgrad8_gs = 1; on the GPU: test_conv0_weight_gradient_forms_vs_fp64_autograd); one tile per workgroup
    pytest.param(2, 1, (8, 8, 32), 3, marks=_full),     # whole tiles
    (2, 1, (1, 1, 1), 192),        # one voxel: 26 of 27 taps see only the zero page
    (1, 1, (3, 9, 17), 3),         # eight waves; one voxel past a tile in H and W
    pytest.param(2, 1, (12, 12, 48), 5, marks=_full),   # 27 tiles, the middle one takes the interior path (no bounds checks); the GPU suite runs it at config 2's size
], ids=["sixteen_waves_ragged_batch_2", "eight_waves_one_tile_per_workgroup", "whole_tiles", "one_voxel", "one_past_a_tile", "interior_tile"])
def test_conv0_weight_gradient_output_gradient_shifted_form(emul_lib, waves, b, dims, groups):
    """conv_c8_wgrad_gs_kernel (conv0, 32 -> 8: X read unshifted without a halo, G with the halo as the shifted operand, 16x16x4 MFMA with
    N = (tap pair, co), double-buffered LDS-DMA tiles, eight or sixteen waves) against autograd and the 4x4x1-MFMA kernel it replaces."""
    from mvs_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(b, 32, *dims, generator=g)
    w = (torch.randn(8, 32, 3, 3, 3, generator=g) * 0.2).requires_grad_(True)
    y = F.conv3d(x, w, stride=1, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    emul_lib.call("mvs_set_tuning", b"wgrad8_gs", 0)
    try:
        old = ops.conv3d_wgrad(x, gy, (8, 32, 3, 3, 3), 1, False)
        emul_lib.call("mvs_set_tuning", b"wgrad8_gs", waves)
        emul_lib.call("mvs_set_tuning", b"wgrad8_groups", groups)
        new = ops.conv3d_wgrad(x, gy, (8, 32, 3, 3, 3), 1, False)
    finally:
        emul_lib.call("mvs_set_tuning", b"wgrad8_gs", 2)
        emul_lib.call("mvs_set_tuning", b"wgrad8_groups", 192)
    scale = max(1.0, float(w.grad.abs().max()))
    assert float((new - w.grad).abs().max()) < 1e-3 * scale
    assert float((new - old).abs().max()) < 2e-4 * scale
    assert not torch.equal(new, torch.zeros_like(new))
    # every tap of every (ci, co) is a different sum: a wrong tap pairing or operand shift cannot hide behind the tolerance
    assert float((new - w.grad).abs().max()) < 0.05 * float(w.grad.abs().mean())


@pytest.mark.parametrize("which", ["small", pytest.param("mvsnet", marks=_full), pytest.param("cvp", marks=_full)])
def test_regulariser_pass_level_c_entry_equals_per_layer_calls(emul_lib, which):
    """mvs_unet_fwd / mvs_unet_bwd (ONE C call per pass over pointer tables into three arenas) against the same autograd node issuing
    the per-layer C calls from Python: the same kernels in the same order => logits, BatchNorm buffers, input gradient and every
    parameter gradient BIT-IDENTICAL; a frozen weight gets no gradient (its launch is skipped); a second backward through the same
    graph gets fresh statistic accumulators.  "small": a three-block U-Net (stride-1, stride-2, transposed stride-2 with skip) for the
    default suite; the two networks' regularisers take minutes of emulation (MVS_EMUL_FULL=1) and run on the GPU."""
    from mvs_amd import nn3d, ops
    if which == "small":
        def build():
            torch.manual_seed(5)
            return torch.nn.ModuleList([nn3d.ConvBnReLU3D(8, 8, stride=1), nn3d.ConvBnReLU3D(8, 16, stride=2),
                                        nn3d.DeconvBnReLU3D(16, 8, stride=2), nn3d.ProbConv3d(8)]).train()
        x0 = torch.randn(1, 8, 4, 4, 16, generator=torch.Generator().manual_seed(1))
        run = lambda m, x: ops.unet_regulariser(x, [(m[0].conv, m[0].bn, False, 1, -1, -1), (m[1].conv, m[1].bn, False, 2, 0, -1),
                                                    (m[2][0], m[2][1], True, 2, 1, 0)], m[3])
        frozen = "1.conv.weight"
    else:
        if which == "mvsnet":
            from mvs_amd.jdacs.models.mvsnet import CostRegNet
            x0 = torch.randn(1, 32, 8, 8, 16, generator=torch.Generator().manual_seed(1))
        else:
            from mvs_amd.jdacs_ms.models.network import CostRegNet
            x0 = torch.randn(1, 16, 2, 4, 16, generator=torch.Generator().manual_seed(1))

        def build():
            torch.manual_seed(5)
            return CostRegNet().train()
        run = lambda m, x: m(x)
        frozen = "conv2.conv.weight"
    res = {}
    for c_entry in (True, False):
        net = build()
        dict(net.named_parameters())[frozen].requires_grad_(False)
        x = x0.clone().requires_grad_(True)
        old = ops.C_ENTRY
        ops.C_ENTRY = c_entry
        try:
            y = run(net, x)
            gout = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
            y.backward(gout, retain_graph=c_entry)
            g1 = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
            gx1 = x.grad.clone()
            g2 = None
            if c_entry:                                        # second backward through the same graph
                for p in net.parameters():
                    p.grad = None
                y.backward(gout)
                g2 = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        finally:
            ops.C_ENTRY = old
        res[c_entry] = (y.detach(), gx1, g1, {k: v.clone() for k, v in net.state_dict().items()}, g2)
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert set(a[2]) == set(b[2]) and frozen not in a[2] and len(a[2]) >= 8
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
        assert torch.equal(a[4][k], a[2][k]), k               # the second backward reproduces the first
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k


def _run_program(emul_lib, c_entry, layers_of, x0):
    from mvs_amd import ops
    net = layers_of()
    x = x0.clone().requires_grad_(True)
    old = ops.C_ENTRY
    ops.C_ENTRY = c_entry
    try:
        y = net.run(x)
        y.backward(torch.randn(y.shape, generator=torch.Generator().manual_seed(2)))
    finally:
        ops.C_ENTRY = old
    return y.detach(), x.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()}


def test_regulariser_c_entry_two_readers_of_the_volume_and_unsupported_programs(emul_lib):
    """ADVICE r5.  (1) a program in which TWO blocks read the volume x: mvs_unet_bwd used to write gx twice with add = null (the last
    write won, silently); it now accumulates, and the C entry equals the per-layer path.  (2) a block that is the skip operand of two
    later blocks (needs an explicit gradient add, which mvs_unet_bwd refuses): `_UnetPlan.ok` must send it down the per-layer path at
    FORWARD time -- the result then equals C_ENTRY = False trivially, and no MVS_ERR_UNSUPPORTED surfaces in backward()."""
    from mvs_amd import nn3d, ops
    x0 = torch.randn(1, 8, 4, 4, 16, generator=torch.Generator().manual_seed(1))

    class TwoReaders(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(7)
            self.a, self.b, self.c = nn3d.ConvBnReLU3D(8, 8), nn3d.ConvBnReLU3D(8, 8), nn3d.ConvBnReLU3D(8, 8)
            self.prob = nn3d.ProbConv3d(8)
            self.train()

        def run(self, x):     # a = f(x); b = g(x) + a; c = h(b)
            return ops.unet_regulariser(x, [(self.a.conv, self.a.bn, False, 1, -1, -1), (self.b.conv, self.b.bn, False, 1, -1, 0),
                                            (self.c.conv, self.c.bn, False, 1, 1, -1)], self.prob)

    prog = ((False, 1, -1, -1, 1e-5, 0.1), (False, 1, -1, 0, 1e-5, 0.1), (False, 1, 1, -1, 1e-5, 0.1))
    plan = ops._UnetPlan(emul_lib, prog, (1, 8, 4, 4, 16), [(8, 8, 3, 3, 3)] * 3, 1)
    assert plan.ok
    a, b = _run_program(emul_lib, True, TwoReaders, x0), _run_program(emul_lib, False, TwoReaders, x0)
    assert torch.equal(a[0], b[0])
    assert torch.equal(a[1], b[1]), "x.grad: two readers of the volume"
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k

    class TwoSkips(TwoReaders):
        def run(self, x):     # a = f(x); b = g(a) + a; c = h(b) + a: block 0 is the skip operand of blocks 1 and 2
            return ops.unet_regulariser(x, [(self.a.conv, self.a.bn, False, 1, -1, -1), (self.b.conv, self.b.bn, False, 1, 0, 0),
                                            (self.c.conv, self.c.bn, False, 1, 1, 0)], self.prob)

    prog2 = ((False, 1, -1, -1, 1e-5, 0.1), (False, 1, 0, 0, 1e-5, 0.1), (False, 1, 1, 0, 1e-5, 0.1))
    assert not ops._UnetPlan(emul_lib, prog2, (1, 8, 4, 4, 16), [(8, 8, 3, 3, 3)] * 3, 1).ok
    a, b = _run_program(emul_lib, True, TwoSkips, x0), _run_program(emul_lib, False, TwoSkips, x0)
    assert torch.equal(a[1], b[1])
    # the skip consumer BEFORE the input consumer in the backward order is the supported direction; the other one is refused too
    prog3 = ((False, 1, -1, -1, 1e-5, 0.1), (False, 1, -1, 0, 1e-5, 0.1), (False, 1, 0, 1, 1e-5, 0.1))   # block 0: skip of 1, input of 2
    assert not ops._UnetPlan(emul_lib, prog3, (1, 8, 4, 4, 16), [(8, 8, 3, 3, 3)] * 3, 1).ok


def test_regulariser_c_entry_rejects_bad_programs(emul_lib):
    """C-ABI error behaviour of mvs_unet_fwd: a block that reads a later block, too many blocks, null tables."""
    import ctypes as C
    from mvs_amd import _lib
    blocks = (_lib.MvsUnetBlock * 2)()
    blocks[0].src, blocks[0].skip, blocks[0].stride = 1, -1, 1          # reads block 1: not an earlier block
    blocks[1].src, blocks[1].skip, blocks[1].stride = 0, -1, 1
    null = (C.c_void_p * 2)()
    with pytest.raises(ValueError, match="earlier blocks"):
        emul_lib.call("mvs_unet_fwd", 2, blocks, 1, None, null, null, null, null, null, null, null, null, null, null, (C.c_int * 2)(), None, None,
                      1, None, None, None)
    blocks[0].src = -1
    with pytest.raises(ValueError, match="null pointer"):
        emul_lib.call("mvs_unet_fwd", 2, blocks, 1, None, null, null, null, null, null, null, null, null, null, null, (C.c_int * 2)(), None, None,
                      1, None, None, None)
    with pytest.raises(ValueError, match="blocks"):
        emul_lib.call("mvs_unet_fwd", 33, blocks, 1, None, null, null, null, null, null, null, null, null, null, null, (C.c_int * 2)(), None, None,
                      1, None, None, None)
