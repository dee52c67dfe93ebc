"""Oracle (oracle/ref_torch.py) vs fixtures generated from the imported reference.  CPU only."""
import os

import pytest
import torch

from conftest import load_golden, rel_l1, state_dict_from
from oracle import ref_torch as R

torch.set_num_threads(4)
ATOL_VOL = 1e-5  # variance / warped volumes: abs 1e-5 + rel 1e-4 (SURVEY 8(c) caveat iii)


def close(a, b, atol=ATOL_VOL, rtol=1e-4):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    assert bool((err <= atol + rtol * b.abs()).all()), "max err %g" % float(err.detach().max())


def test_g1_homo_warping():
    for tag in "ab":
        g = load_golden("g1_homo_warping_" + tag)
        src = g["src_fea"].clone().requires_grad_(True)
        out = R.homo_warping(src, g["src_proj"], g["ref_proj"], g["depth_values"])
        close(out, g["out"])
        out.backward(g["grad_out"])
        close(src.grad, g["grad_src"], atol=1e-4)
        assert float((g["out"] == 0).float().mean()) > 0.02  # fixture exercises out-of-image samples


def test_aten_sampler_mode_equals_gather_form():
    """bench.py times the oracle with sampler="aten" (the reference's own F.grid_sample call, module.py:131-136) as the CPU
    baseline / reference GPU path; that form must be the same function as the hand-written gather the parity tests use --
    forward and gradient, on the reference-generated fixture (out-of-image samples included) and per-pixel hypotheses."""
    for tag in "ab":
        g = load_golden("g1_homo_warping_" + tag)
        outs, grads = [], []
        for sampler in ("gather", "aten"):
            old = R.set_sampler(sampler)
            try:
                src = g["src_fea"].clone().requires_grad_(True)
                out = R.homo_warping(src, g["src_proj"], g["ref_proj"], g["depth_values"])
                out.backward(g["grad_out"])
            finally:
                R.set_sampler(old)
            outs.append(out.detach())
            grads.append(src.grad.clone())
        # measured 1.04e-6 / 1.3e-6 of the largest value (fp32 rounding of ix = ((g+1)*W-1)/2 inside ATen vs the same expression here)
        assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * max(1.0, float(outs[0].abs().max()))
        assert float((grads[0] - grads[1]).abs().max()) <= 2e-6 * max(1.0, float(grads[0].abs().max()))
        assert bool((outs[1] == g["out"]).all())     # and the aten form reproduces the reference fixture bit for bit
    assert R.SAMPLER == "gather"
    gen = torch.Generator().manual_seed(2)
    fea = torch.randn(2, 8, 12, 20, generator=gen)
    rot, trans = R.relative_projection(g["src_proj"][:1].repeat(2, 1, 1), g["ref_proj"][:1].repeat(2, 1, 1))
    depth = 450 + 30 * torch.rand(2, 1, 12, 20, generator=gen) + 20.0 * torch.arange(5).view(1, 5, 1, 1)
    a = R.warp_features(fea, rot, trans, depth, sampler="gather")
    b = R.warp_features(fea, rot, trans, depth, sampler="aten")
    assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))
    with pytest.raises(ValueError):
        R.set_sampler("nearest")


def test_g3_proj_cost_and_ms_warp():
    g = load_golden("g3_proj_cost")
    ref = g["ref_fea"].clone().requires_grad_(True)
    srcs = [g["src_fea0"].clone().requires_grad_(True), g["src_fea1"].clone().requires_grad_(True)]
    cost = R.proj_cost(2, ref, srcs, g["ref_in"], g["src_in"], g["ref_ex"], g["src_ex"], g["hypos"])
    close(cost, g["cost"], atol=1e-4)
    cost.backward(g["grad_out"])
    close(ref.grad, g["grad_ref"], atol=2e-3, rtol=1e-3)
    close(srcs[0].grad, g["grad_src0"], atol=2e-3, rtol=1e-3)
    close(srcs[1].grad, g["grad_src1"], atol=2e-3, rtol=1e-3)
    w = R.homo_warping_ms(g["src_fea0"], g["ref_in"], g["src_in"][:, 0], g["ref_ex"], g["src_ex"][:, 0], g["planes"])
    close(w, g["warped_ms"])


def _check_regnet(g, net, has_second):
    net.load_state_dict(state_dict_from(g))
    net.train()
    x = g["x"].clone().requires_grad_(True)
    y = net(x)
    y = y if y.dim() == g["y_train"].dim() else y.unsqueeze(1)
    close(y, g["y_train"], atol=2e-4, rtol=1e-3)
    y.backward(g["grad_out"].view_as(y))
    close(x.grad, g["grad_x"], atol=1e-3, rtol=2e-2)
    for k, p in net.named_parameters():
        ref = g["grad." + k]
        assert rel_l1(p.grad, ref) < 2e-3, k
    sd = net.state_dict()
    for k, v in g.items():
        if k.startswith("after1.") and "num_batches" not in k:
            close(sd[k[7:]], v, atol=1e-5, rtol=1e-4)
    if has_second:
        with torch.no_grad():
            net(g["x2"])
    net.eval()
    with torch.no_grad():
        ye = net(g["x"])
    ye = ye if ye.dim() == g["y_eval"].dim() else ye.unsqueeze(1)
    close(ye, g["y_eval"], atol=2e-4, rtol=1e-3)


def test_g4_costregnet_mvs():
    _check_regnet(load_golden("g4_costregnet_mvs"), R.OracleCostRegNet(), True)


def test_g4_costregnet_cvp():
    _check_regnet(load_golden("g4_costregnet_cvp"), R.OracleCostRegNetMS(), False)


def test_g5_softargmin():
    g = load_golden("g5_softargmin")
    lg = g["logits"].clone().requires_grad_(True)
    depth, conf, _ = R.softargmin_conf(lg, g["depth_values"])
    close(depth, g["depth"], atol=1e-3, rtol=1e-6)
    close(conf, g["conf"], atol=1e-6)
    depth.backward(g["grad_depth"])
    close(lg.grad, g["grad_logits"], atol=1e-4, rtol=1e-4)


def test_g6_mvsnet_end_to_end():
    g = load_golden("g6_mvsnet_e2e")
    net = R.OracleMVSNet(refine=False)
    net.load_state_dict(state_dict_from(g))
    net.train()
    out = net(g["imgs"], g["proj"], g["depth_values"], return_intermediates=True)
    close(out["variance"], g["train_variance"])
    close(out["logits"].unsqueeze(1), g["train_logits"], atol=1e-3, rtol=1e-3)
    assert rel_l1(out["depth"], g["train_depth"]) < 1e-5
    assert float(g["train_depth"].std()) > 1.0  # non-degenerate golden (SURVEY 8(c) caveat ii)
    close(out["photometric_confidence"], g["train_conf"], atol=1e-4)
    wts = torch.linspace(0.5, 1.5, out["depth"].numel()).view_as(out["depth"])
    (out["depth"] * wts).mean().backward()
    for k, p in net.named_parameters():
        if k.endswith("prob.bias"):  # softmax is shift invariant: true gradient is 0, value is roundoff
            assert float(p.grad.abs().max()) < 1e-4
            continue
        assert rel_l1(p.grad.detach(), g["grad." + k]) < 5e-3, k
    # eval with the fixture's calibrated running stats
    sd = net.state_dict()
    for k, v in g.items():
        if k.startswith("cal."):
            sd[k[4:]] = v
    net.load_state_dict(sd)
    net.eval()
    with torch.no_grad():
        oe = net(g["imgs"], g["proj"], g["depth_values"], return_intermediates=True)
    close(oe["variance"], g["eval_variance"])
    assert rel_l1(oe["depth"], g["eval_depth"]) < 1e-5
    close(oe["photometric_confidence"], g["eval_conf"], atol=1e-4)


def test_g7_cvpmvsnet_end_to_end():
    g = load_golden("g7_cvpmvsnet_e2e")
    net = R.OracleCVPMVSNet(R.cvp_args(nsrc=2, nscale=2, mode="train"))
    net.load_state_dict(state_dict_from(g), strict=False)
    net.train()
    with torch.no_grad():
        out = net(g["ref_img"], g["src_imgs"], g["ref_in"], g["src_in"], g["ref_ex"], g["src_ex"],
                  g["depth_min"], g["depth_max"])
        hyp = R.cal_depth_hypo(g["depth_up"], g["ref_in"], g["src_in"], g["ref_ex"], g["src_ex"])
    close(hyp, g["hypos0"], atol=1e-3, rtol=1e-6)
    assert rel_l1(out["depth_est_list"][1], g["depth1"]) < 1e-5
    assert rel_l1(out["depth_est_list"][0], g["depth0"]) < 1e-4
    close(out["prob_confidence"], g["conf"], atol=1e-3)


@pytest.mark.parametrize("name", ["g8_unsup_loss", "g8_unsup_loss_n4"])
def test_unsup_loss_oracle_vs_golden(name):
    """the oracle's restatement of UnSupLoss against the fixture generated by the imported reference
    (tests/golden/make_golden_unsup.py): loss, its three terms, d loss / d depth, and the first view's warp + mask."""
    g = load_golden(name)
    imgs, cams = g["imgs"].float(), g["cams"]
    depth = g["depth"].clone().requires_grad_(True)
    total, reconstr, ssim, smooth = R.unsup_loss(imgs, cams, depth, return_terms=True)
    total.backward()
    assert abs(float(total) - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    assert abs(float(reconstr) - float(g["reconstr_loss"])) < 1e-5
    assert abs(float(ssim) - float(g["ssim_loss"])) < 1e-5
    assert abs(float(smooth) - float(g["smooth_loss"])) < 1e-4
    assert float((depth.grad - g["grad_depth"]).abs().max()) < 2e-6 + 1e-4 * float(g["grad_depth"].abs().max())
    kinv, proj = R.unsup_view_transform(cams[:, 0], cams[:, 1])
    warped, mask = R.unsup_inverse_warp(R.quarter_image(imgs[:, 1]), kinv, proj, g["depth"])
    assert torch.equal(mask, g["mask1"])
    assert float((warped - g["warped1"]).abs().max()) < 2e-4


def test_g3b_ms_homo_warping_with_gradient():
    """jdacs-ms homo_warping (modules.py:62-104) incl. d/d src_feature, two cases generated by the imported reference."""
    g = load_golden("g3b_ms_homo_warping")
    for tag in "ab":
        src = g[tag + "_src"].clone().requires_grad_(True)
        w = R.homo_warping_ms(src, g[tag + "_ref_in"], g[tag + "_src_in"], g[tag + "_ref_ex"], g[tag + "_src_ex"], g[tag + "_planes"])
        close(w, g[tag + "_warped"])
        w.backward(g[tag + "_grad_out"])
        close(src.grad, g[tag + "_grad_src"], atol=1e-4)


def test_g10_geometric_consistency_filter():
    """oracle/geo_filter_np.py vs the outputs of the reference's own functions (jdacs/eval.py:169-224, executed by
    tests/golden/make_golden_geo.py) and of the aggregation lines eval.py:379-388."""
    import numpy as np
    from oracle import geo_filter_np as G
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_geo_filter.npz"))
    nsrc = z["depth_src"].shape[0]
    for v in range(1, nsrc + 1):
        mask, rep, xs, ys = G.check_geometric_consistency(z["depth_ref"], z["K"][0], z["E"][0], z["depth_src"][v - 1], z["K"][v], z["E"][v])
        assert np.array_equal(mask, z["mask%d" % v]) and np.array_equal(rep, z["reproj%d" % v])
        assert np.array_equal(xs, z["x_src%d" % v]) and np.array_equal(ys, z["y_src%d" % v])
        assert 0.05 < mask.mean() < 0.999 or v == 1          # the fixture exercises both outcomes
    r = G.filter_depth_view(z["depth_ref"], z["conf_ref"], z["K"][0], z["E"][0], list(z["depth_src"]), list(z["K"][1:]), list(z["E"][1:]))
    assert np.array_equal(r["geo_count"], z["geo_count"]) and np.array_equal(r["final_mask"], z["final_mask"])
    assert np.array_equal(r["depth_avg"], z["depth_avg"])
