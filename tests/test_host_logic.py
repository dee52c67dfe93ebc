"""CPU: host-side logic of the drop-in modules that needs no kernel (loss form, state_dict layout,
depth-hypothesis helpers, argument validation)."""
import pytest
import torch

from conftest import load_golden, state_dict_from
from oracle import ref_torch as R


def test_mvsnet_loss_matches_boolean_index_form():
    from mvs_amd.jdacs.models.mvsnet import mvsnet_loss
    g = torch.Generator().manual_seed(0)
    est = (600 + 50 * torch.randn(2, 16, 20, generator=g)).requires_grad_(True)
    gt = 600 + 50 * torch.randn(2, 16, 20, generator=g)
    gt[0, :2] = est.detach()[0, :2] + 0.3          # exercise the quadratic branch of smooth-L1 too
    mask = (torch.rand(2, 16, 20, generator=g) > 0.3).float()
    a = mvsnet_loss(est, gt, mask)
    est2 = est.detach().clone().requires_grad_(True)
    b = R.mvsnet_loss(est2, gt, mask)
    a.backward()
    b.backward()
    assert torch.allclose(a, b, rtol=1e-6)
    assert torch.allclose(est.grad, est2.grad, rtol=1e-5, atol=1e-9)


def test_state_dict_layout_matches_reference_fixtures():
    from mvs_amd.jdacs.models.mvsnet import MVSNet
    from mvs_amd.jdacs_ms.models.network import CVPMVSNet
    g6 = state_dict_from(load_golden("g6_mvsnet_e2e"))
    net = MVSNet(refine=False)
    sd = net.state_dict()
    assert set(sd) == set(g6) and all(sd[k].shape == g6[k].shape for k in sd)
    assert sum(p.numel() for p in net.parameters()) == 338129 and len(sd) == 106
    net.load_state_dict(g6)  # strict
    assert len(MVSNet(refine=True).state_dict()) > 106  # refine_network.* present like the reference
    g7 = state_dict_from(load_golden("g7_cvpmvsnet_e2e"))
    cvp = CVPMVSNet(R.cvp_args(2, 2, "test"))
    sd = cvp.state_dict()
    assert len(sd) == 80 and sum(p.numel() for p in cvp.parameters()) == 551585
    assert all(k in sd and sd[k].shape == v.shape for k, v in g7.items())


def test_same_seed_gives_reference_initialisation():
    """Parameter containers are the stock nn modules in the reference's construction order, so
    torch.manual_seed(s) + constructor reproduces the reference's (== the oracle's) initial weights."""
    from mvs_amd.jdacs.models.mvsnet import MVSNet
    torch.manual_seed(3)
    a = MVSNet(refine=False).state_dict()
    torch.manual_seed(3)
    b = R.OracleMVSNet(refine=False).state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_cvp_depth_hypothesis_helpers():
    from mvs_amd.jdacs_ms.models import modules as M
    g = load_golden("g7_cvpmvsnet_e2e")
    with pytest.raises(RuntimeError):     # calDepthHypo is a kernel (csrc/depth_hypo.hip): no CPU fallback
        M.calDepthHypo(None, g["depth_up"], g["ref_in"], g["src_in"], g["ref_ex"], g["src_ex"], None, None, 0)
    planes = M.calSweepingDepthHypo(g["ref_in"], None, None, None, g["depth_min"], g["depth_max"])
    assert planes.shape == (1, 48) and float(planes[0, 0]) == 425.0 and float(planes[0, -1]) == 425.0 + 47 * 13.5
    k = M.conditionIntrinsics(g["ref_in"], (1, 3, 64, 96), [(1, 16, 64, 96), (1, 16, 32, 48)])
    assert k.shape == (1, 2, 3, 3) and torch.allclose(k[0, 1, :2], g["ref_in"][0, :2] / 2) and k[0, 1, 2, 2] == 1


def test_shape_validation_raises_early():
    from mvs_amd.jdacs.models.mvsnet import CostRegNet
    from mvs_amd.nn3d import ConvBnReLU3D
    with pytest.raises(ValueError, match="divisible by 8"):
        CostRegNet()(torch.zeros(1, 32, 12, 16, 16))
    with pytest.raises(ValueError):
        ConvBnReLU3D(8, 8, kernel_size=5)
