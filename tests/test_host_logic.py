"""CPU: host-side logic of the drop-in modules that needs no kernel (loss form, state_dict layout,
depth-hypothesis helpers, argument validation)."""
import os

import numpy as np

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


def test_pfm_bytes_match_the_reference_and_round_trip(tmp_path):
    """the on-disk format of the path's outputs.  The fixture holds the bytes the reference's save_pfm wrote."""
    import numpy as np
    from mvs_amd.jdacs.datasets.data_io import read_pfm, save_pfm
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_pfm.npz"))
    for key, scale in (("depth", 1), ("color", 2.5)):
        p = str(tmp_path / (key + ".pfm"))
        save_pfm(p, g[key], scale)
        assert open(p, "rb").read() == g[key + "_bytes"].tobytes()
        back, sc = read_pfm(p)
        assert np.array_equal(back, g[key]) and sc == scale
    save_pfm(str(tmp_path / "one.pfm"), g["depth"][:, :, None])
    assert read_pfm(str(tmp_path / "one.pfm"))[0].shape == (32, 40)
    with pytest.raises(Exception, match="float32"):
        save_pfm(str(tmp_path / "x.pfm"), g["depth"].astype(np.float64))
    with pytest.raises(Exception, match="dimensions"):
        save_pfm(str(tmp_path / "x.pfm"), np.zeros((2, 3, 4), np.float32))
    (tmp_path / "bad.pfm").write_bytes(b"P6\n1 1\n255\n")
    with pytest.raises(Exception, match="Not a PFM"):
        read_pfm(str(tmp_path / "bad.pfm"))


def test_eval_plumbing_helpers(tmp_path):
    """checkpoint with DataParallel's `module.` prefix -> model; outputs dict -> PFM files named like jdacs/eval.py:155-164."""
    import numpy as np
    from mvs_amd.jdacs.datasets.data_io import read_pfm
    from mvs_amd.jdacs.models.mvsnet import MVSNet
    from mvs_amd.jdacs.utils import load_checkpoint, save_depth_outputs, tensor2float, tensor2numpy
    torch.manual_seed(1)
    src = MVSNet(refine=False)
    ckpt = {"epoch": 3, "model": {"module." + k: v for k, v in src.state_dict().items()}}
    torch.save(ckpt, str(tmp_path / "model_000003.ckpt"))
    dst = MVSNet(refine=False)
    res = load_checkpoint(dst, str(tmp_path / "model_000003.ckpt"))
    assert not res.missing_keys and not res.unexpected_keys
    assert all(torch.equal(v, dst.state_dict()[k]) for k, v in src.state_dict().items())
    outputs = {"depth": torch.rand(2, 8, 10) + 500, "photometric_confidence": torch.rand(2, 8, 10)}
    names = ["scan1/{}/00000000{}", "scan1/{}/00000001{}"]
    paths = save_depth_outputs(outputs, names, str(tmp_path / "out"))
    assert len(paths) == 4 and paths[0].endswith("scan1/depth_est/00000000.pfm") and paths[3].endswith("scan1/confidence/00000001.pfm")
    assert np.array_equal(read_pfm(paths[2])[0], outputs["depth"][1].numpy())
    nested = tensor2numpy({"a": [torch.ones(2), (torch.zeros(1),)], "b": np.ones(3)})
    assert isinstance(nested["a"][1][0], np.ndarray) and nested["a"][0].sum() == 2
    assert tensor2float({"loss": torch.tensor(1.5), "lr": 0.1}) == {"loss": 1.5, "lr": 0.1}
    with pytest.raises(NotImplementedError):
        tensor2numpy("x")


def test_flat_bucket_keeps_channels_last_params_trainable():
    """parameters flattened into one store must stay views of it -- also channels-last conv weights --
    and one bucket + optimiser step must move EVERY parameter."""
    import mvs_amd  # noqa: F401
    from mvs_amd import dist as mdist
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                                torch.nn.Conv2d(8, 4, 5, stride=2, padding=2))
    model.to(memory_format=torch.channels_last)          # layout conversion BEFORE the bucket is built
    bucket = mdist.FlatGradBucket(model.parameters(), flatten_params=True)
    assert model[0].weight.is_contiguous(memory_format=torch.channels_last)   # physical layout survived the flattening
    opt = torch.optim.Adam([bucket.flat_param], lr=1e-2)
    before = [p.detach().clone() for p in model.parameters()]
    x = torch.randn(2, 3, 8, 8).contiguous(memory_format=torch.channels_last)
    bucket.zero()
    model(x).square().mean().backward()
    ref_grads = [p.grad.clone() for p in model.parameters()]
    bucket.gather()
    opt.step()
    for p, b, g in zip(model.parameters(), before, ref_grads):
        assert not torch.equal(p.detach(), b), "a parameter did not move"
        # Adam's first step moves every element by lr * sign(grad) (up to eps)
        moved = (p.detach() - b)
        nz = g.abs() > 1e-6
        assert torch.allclose(moved[nz], -1e-2 * torch.sign(g[nz]), atol=1e-4)
    # a re-allocation of parameter storage after the bucket was built must be caught, not silently ignored
    model[3].weight.data = model[3].weight.data.clone()
    bucket.zero()
    model(x).square().mean().backward()
    with pytest.raises(RuntimeError, match="no longer aliases"):
        bucket.gather()

def test_flat_bucket_gather_fast_path_equals_the_concatenation():
    """FlatGradBucket.gather(): the one-launch copy into cached strided views of the bucket (every gradient present, in its parameter's
    layout) writes exactly what the concatenation of the gradients in storage order writes; a missing gradient, a gradient in another
    layout or another dtype take the general path and give the same bucket as before the fast path existed."""
    import mvs_amd  # noqa: F401
    from mvs_amd import dist as mdist
    torch.manual_seed(1)
    model = torch.nn.Sequential(torch.nn.Conv3d(4, 8, 3, padding=1), torch.nn.BatchNorm3d(8), torch.nn.Conv2d(8, 4, 5, stride=2, padding=2),
                                torch.nn.Conv3d(8, 1, 3, padding=1))
    model[0].weight.data = model[0].weight.data.contiguous(memory_format=torch.channels_last_3d)
    model[2].weight.data = model[2].weight.data.contiguous(memory_format=torch.channels_last)
    bucket = mdist.FlatGradBucket(model.parameters(), flatten_params=True)
    params = list(model.parameters())

    def expected():
        parts = []
        for p in params:
            if p.grad is None:
                parts.append(torch.zeros(p.numel()))
            else:
                parts.append(torch.empty_strided(p.shape, p.stride()).copy_(p.grad).as_strided((p.numel(),), (1,)))
        return torch.cat(parts)

    for step in range(3):                                   # the cached views are reused from the second step on
        for p in params:
            p.grad = torch.empty_strided(p.shape, p.stride()).normal_()
        want = expected()
        bucket.gather()
        assert torch.equal(bucket.flat, want)
    assert len(bucket._grad_views) == len(params)
    params[0].grad = torch.randn(params[0].shape)           # a contiguous gradient for a channels-last parameter: general path
    assert params[0].grad.stride() != params[0].stride()
    want = expected()
    bucket.gather()
    assert torch.equal(bucket.flat, want)
    params[3].grad = None                                   # a parameter without a gradient: zeros in its slice
    want = expected()
    bucket.gather()
    assert torch.equal(bucket.flat, want)
    for p in params:                                        # and back on the fast path
        p.grad = torch.empty_strided(p.shape, p.stride()).normal_()
    want = expected()
    bucket.gather()
    assert torch.equal(bucket.flat, want)


def test_sliced_optimizer_params_equal_the_single_flat_tensor():
    """FlatGradBucket.optimizer_params(): Adam over the flat store as N slices gives bit-identical parameters to Adam over the one
    flat tensor (element-wise update), for several steps, and the slices alias the store / the bucket."""
    import mvs_amd  # noqa: F401
    from mvs_amd import dist as mdist

    def run(sliced):
        torch.manual_seed(3)
        model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                                    torch.nn.Conv2d(8, 5, 3, padding=1))
        bucket = mdist.FlatGradBucket(model.parameters(), flatten_params=True)
        params = bucket.optimizer_params(7) if sliced else [bucket.flat_param]
        if sliced:
            assert sum(p.numel() for p in params) == bucket.flat_param.numel() and len(params) >= 6
            assert params[1].data_ptr() == bucket.flat_param.data_ptr() + 4 * params[0].numel()
            assert params[1].grad.data_ptr() == bucket.flat.data_ptr() + 4 * params[0].numel()
        opt = torch.optim.Adam(params, lr=1e-2, betas=(0.9, 0.999))
        g = torch.Generator().manual_seed(4)
        for _ in range(3):
            x = torch.randn(2, 3, 6, 6, generator=g)
            bucket.zero()
            model(x).square().mean().backward()
            bucket.gather()
            opt.step()
        return bucket.flat_param.detach().clone()

    assert torch.equal(run(False), run(True))
    with pytest.raises(ValueError):
        mdist.FlatGradBucket(torch.nn.Linear(2, 2).parameters()).optimizer_params()



def test_mvsnet_construction_fixes_feature_layout_once():
    """The feature extractor is converted to channels-last at construction; forward() must not touch parameter storage."""
    import mvs_amd  # noqa: F401
    from mvs_amd.jdacs.models.mvsnet import MVSNet
    net = MVSNet(refine=False)
    assert net.feature.conv0.conv.weight.is_contiguous(memory_format=torch.channels_last)
    import inspect
    assert "memory_format=torch.channels_last)" not in inspect.getsource(MVSNet._forward).replace(
        "contiguous(memory_format=torch.channels_last)", "")


@pytest.mark.parametrize("how", ["launcher", "sitecustomize"])
@pytest.mark.parametrize("sub,stmt,expect", [
    ("jdacs", "from models.mvsnet import MVSNet, mvsnet_loss\nfrom models.module import homo_warping, ConvBnReLU3D\nm = MVSNet(refine=False)",
     "mvs_amd.jdacs.models.mvsnet"),
    ("jdacs-ms", "from models.network import CVPMVSNet, sL1_loss, MSE_loss\nfrom models.modules import proj_cost, calDepthHypo\nm = CVPMVSNet",
     "mvs_amd.jdacs_ms.models.network"),
])
def test_dropin_runs_reference_import_lines_unchanged(tmp_path, sub, stmt, expect, how):
    """`from models.mvsnet import MVSNet` (jdacs/train.py:28) / `from models.network import CVPMVSNet` (jdacs-ms/train.py:24) in an
    UNCHANGED script resolve to the drop-in modules, while submodules that are not on the path (models/augmentations.py ...) still
    come from the script's own `models/` package.  Stand-in tree here: the reference itself is not on the GPU box."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    work = tmp_path / sub
    (work / "models").mkdir(parents=True)
    (work / "models" / "__init__.py").write_text("")
    (work / "models" / "augmentations.py").write_text("MARK = 'from the working tree'\n")
    # the working tree's own hot-path files must NOT be the ones that get imported
    for name in ("mvsnet", "module", "network", "modules"):
        (work / "models" / (name + ".py")).write_text("raise ImportError('the reference file was imported, not the drop-in')\n")
    (work / "train.py").write_text(stmt + "\nfrom models.augmentations import MARK\n"
                                   "print((m if isinstance(m, type) else type(m)).__module__, '|', MARK)\n")
    env = dict(os.environ)
    if how == "launcher":
        cmd = [sys.executable, os.path.join(root, "dropin", "run.py"), "train.py"]
    else:
        env["PYTHONPATH"] = os.pathsep.join([os.path.join(root, "dropin"), root])
        cmd = [sys.executable, "train.py"]
    r = subprocess.run(cmd, cwd=str(work), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().endswith("%s | from the working tree" % expect), r.stdout


def test_convbnrelu_gate_for_unsupported_bn_shapes():
    """channel counts the BatchNorm kernels do not serve (12, 24, 48 ...) and momentum=None must take the stock
    modules instead of raising; on CPU tensors the stock path is taken anyway, so this checks the gate's source of truth."""
    import inspect
    from mvs_amd.jdacs.models.module import ConvBnReLU
    src = inspect.getsource(ConvBnReLU.forward)
    assert "in (4, 8, 16, 32, 64)" in src and "bn.momentum is not None" in src
    m = ConvBnReLU(3, 12)
    y = m(torch.randn(2, 3, 8, 8))
    assert y.shape == (2, 12, 8, 8) and float(y.min()) >= 0
    from mvs_amd import nn3d
    blk = nn3d.ConvBnReLU3D(8, 8)
    blk.bn.momentum = None
    with pytest.raises(ValueError, match="momentum=None"):
        blk(torch.randn(1, 8, 4, 4, 4))


def test_async_wgrad_use_counts_are_per_outstanding_use():
    """Two graphs built before either backward (forward A, forward B, A.backward(), B.backward()): a weight used twice in B (CVP's
    shared regulariser) must stay on the synchronous path in B's backward although A's backward ran in between."""
    from types import SimpleNamespace
    from mvs_amd import ops

    class W:   # the bookkeeping only looks at these
        is_cuda = True
        device = SimpleNamespace(index=0)

        def __init__(self, ptr):
            self._ptr = ptr

        def data_ptr(self):
            return self._ptr

    old = ops._ASYNC_WGRAD
    ops.set_async_wgrad(True)
    ops._WEIGHT_USES.pop(0, None)
    ops._WEIGHT_MULTI.pop(0, None)
    try:
        single, shared = W(1000), W(2000)
        ops._note_weight_use(single)                                   # graph A
        ops._note_weight_use(shared); ops._note_weight_use(shared)     # graph B: two uses of the same weight
        ops._weight_use_done(0, 1000)                                  # A.backward()
        ops._end_of_backward(0)
        assert 2000 in ops._WEIGHT_MULTI[0] and 1000 not in ops._WEIGHT_MULTI[0]
        ops._weight_use_done(0, 2000)                                  # B.backward(): first node -- still multi-use
        assert 2000 in ops._WEIGHT_MULTI[0]
        ops._weight_use_done(0, 2000)
        assert 2000 not in ops._WEIGHT_MULTI[0] and not ops._WEIGHT_USES[0]
    finally:
        ops.set_async_wgrad(old)
        ops._WEIGHT_USES.pop(0, None)
        ops._WEIGHT_MULTI.pop(0, None)


def test_flat_bucket_reattaches_optimizer_slices_after_zero_grad():
    """optimizer.zero_grad() (set_to_none=True by default) drops the slices' .grad views of the bucket; gather() re-attaches them,
    otherwise the next opt.step() silently skips every slice."""
    from mvs_amd import dist as mdist
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    bucket = mdist.FlatGradBucket(net.parameters(), flatten_params=True)
    opt = torch.optim.SGD(bucket.optimizer_params(4), lr=0.1)
    before = bucket.flat_param.detach().clone()
    for _ in range(2):
        opt.zero_grad()                      # the hazard: slices lose their .grad
        bucket.zero()
        net(torch.randn(4, 5)).sum().backward()
        bucket.gather()
        opt.step()
    assert all(p.grad is not None for p, _, _ in bucket._opt_slices)
    assert float((bucket.flat_param.detach() - before).abs().max()) > 0


def test_gipuma_format_glue_reproduces_the_reference_bytes(tmp_path):
    """jdacs/fusion/depthfusion.py's format functions, byte for byte against files the reference's OWN functions wrote
    (tests/golden/make_golden_fusion.py executes them out of the reference's syntax tree): disp.dmb, the fake normals.dmb (with the
    channel-planar quirk), read-back, camera txt -> load_cam -> "<view>.jpg.P" text, probability_filter over 49 views."""
    from mvs_amd.jdacs.fusion import depthfusion as DF
    g = {k: v.numpy() if hasattr(v, "numpy") else v for k, v in load_golden("g11_gipuma_formats").items()}
    rd = lambda p: np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
    DF.write_pfm(str(tmp_path / "d.pfm"), g["depth"])
    DF.mvsnet_to_gipuma_dmb(str(tmp_path / "d.pfm"), str(tmp_path / "disp.dmb"))
    assert np.array_equal(rd(tmp_path / "disp.dmb"), g["disp_dmb_bytes"])
    DF.fake_gipuma_normal(str(tmp_path / "disp.dmb"), str(tmp_path / "normals.dmb"))
    assert np.array_equal(rd(tmp_path / "normals.dmb"), g["normals_dmb_bytes"])
    assert np.array_equal(DF.read_gipuma_dmb(str(tmp_path / "disp.dmb")), g["disp_read_back"])
    assert np.array_equal(DF.read_gipuma_dmb(str(tmp_path / "normals.dmb")), g["normals_read_back"])
    DF.write_gipuma_dmb(str(tmp_path / "c.dmb"), g["colour"])
    assert np.array_equal(rd(tmp_path / "c.dmb"), g["colour_dmb_bytes"])
    for tag in "abc":
        (tmp_path / ("cam_%s.txt" % tag)).write_bytes(g["cam_txt_" + tag].tobytes())
        with open(tmp_path / ("cam_%s.txt" % tag)) as fh:
            assert np.array_equal(DF.load_cam(fh), g["cam_loaded_" + tag])
        DF.mvsnet_to_gipuma_cam(str(tmp_path / ("cam_%s.txt" % tag)), str(tmp_path / ("cam_%s.P" % tag)))
        assert np.array_equal(rd(tmp_path / ("cam_%s.P" % tag)), g["cam_P_bytes_" + tag]), tag
        P = DF.read_p_file(str(tmp_path / ("cam_%s.P" % tag)))                      # what the fusion program reads back
        want = (g["cam_loaded_" + tag][1][:3, :3] @ g["cam_loaded_" + tag][0][:3]).astype(np.float32)
        assert P.shape == (3, 4) and np.allclose(P, want, rtol=1e-6, atol=1e-3)
    scan = tmp_path / "scan"
    (scan / "depth_est").mkdir(parents=True)
    (scan / "confidence").mkdir()
    for v in range(49):
        DF.write_pfm(str(scan / "depth_est" / ("%08d.pfm" % v)), g["pf_depth"][v])
        DF.write_pfm(str(scan / "confidence" / ("%08d.pfm" % v)), g["pf_prob"][v])
    DF.probability_filter(str(scan), 0.8)
    for v in range(49):
        assert np.array_equal(rd(scan / "depth_est" / ("%08d_prob_filtered.pfm" % v)), g["pf_filtered_bytes"][v]), v


def test_write_depth_img_png_matches_the_reference(tmp_path):
    """jdacs/eval.py:110-123: (depth - 500) / 2 through PIL's float -> 8-bit conversion, clipping below 0 and above 255."""
    from PIL import Image
    from mvs_amd.jdacs.utils import save_depth_outputs, write_depth_img
    g = {k: v.numpy() if hasattr(v, "numpy") else v for k, v in load_golden("g11_gipuma_formats").items()}
    path = tmp_path / "a" / "b" / "00000000.pfm.png"
    assert write_depth_img(str(path), g["png_depth"]) == 1
    assert np.array_equal(np.asarray(Image.open(path)), g["png_pixels"])
    assert int(g["png_pixels"].min()) == 0 and int(g["png_pixels"].max()) == 255         # the fixture exercises both clips
    if Image.__version__.encode() == g["pil_version"].tobytes():                      # same encoder -> same file
        assert np.array_equal(np.frombuffer(path.read_bytes(), dtype=np.uint8), g["png_bytes"])
    out = {"depth": torch.from_numpy(g["png_depth"])[None], "photometric_confidence": torch.rand(1, 12, 16)}
    written = save_depth_outputs(out, ["scan1/{}/00000003{}"], str(tmp_path / "o"), depth_png=True)
    assert [os.path.basename(p) for p in written] == ["00000003.pfm", "00000003.pfm.png", "00000003.pfm"]
    assert np.array_equal(np.asarray(Image.open(written[1])), g["png_pixels"])


def test_fusibile_oracle_fuses_a_consistent_scene_onto_its_surface():
    """The numpy restatement of the fusion program on a synthetic 5-view scene: the fused points lie on the surface the depth maps
    were rendered from, the noisy view contributes fewer points, holes contribute none, the two camera preparations (product:
    scipy RQ; oracle: flipped QR) agree, and the .ply bytes have the program's layout."""
    from mvs_amd.jdacs.fusion import depthfusion as DF
    from oracle import fusibile_np as FO
    Ps, nd, img, Ks, Es = FO.synthetic_scene(5, 24, 32, seed=1)
    co = FO.fusibile_cameras(Ps)
    cams, f = DF.fusibile_cameras(Ps)
    assert np.allclose(cams, co["cams"], rtol=2e-4, atol=2e-3) and abs(f - float(co["f"])) < 1e-3
    assert np.allclose(cams[:, :12].reshape(5, 3, 4), np.stack(Ps), rtol=1e-4, atol=5e-2)     # one K for all views here: P is rebuilt as it was
    per_view = []
    for ref in range(5):
        out, count = FO.fuse_view(nd, img, co["cams"], list(range(5)), ref, co["f"], 0.25, 2 * np.pi, 2)
        pts = FO.compact(out)
        per_view.append(pts.shape[0])
        if pts.shape[0]:
            err = np.abs(FO.scene_surface(pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)) - pts[:, 2])
            # one pixel is ~20 units wide at this resolution and the program averages points re-projected from TRUNCATED pixel
            # positions (fusibile.cu:224-226) on a surface with slopes up to ~0.5: a few units of scatter are the algorithm's own
            assert np.median(err) < 5.0 and err.max() < 30.0, (ref, np.median(err), err.max())
        if ref == 3:                                                       # the view with a hole: nothing fused from inside it
            assert not out[24 // 5:24 // 2, 32 // 3:2 * 32 // 3, :3].any()
    assert per_view[2] < min(per_view[0], per_view[1])                     # the noisy view agrees with fewer neighbours
    allp = FO.fuse_all(nd, img, co["cams"], co["f"], 0.25, 2 * np.pi, 2)
    assert allp.shape[0] == sum(per_view)
    ply = FO.ply_bytes(allp)
    assert ply == DF.ply_bytes(allp)
    head, _, body = ply.partition(b"end_header\n")
    assert b"element vertex %d\n" % allp.shape[0] in head and len(body) == 15 * allp.shape[0]


def test_folded_eval_batchnorm_cache_follows_parameter_updates():
    """ConvBnReLU._folded (inference: BatchNorm folded into the convolution): the cached (w', b') is rebuilt after in-place updates of
    any of its five source tensors (optimizer step, load_state_dict) and conv(x, w') + b' equals the eval-mode block before ReLU."""
    import torch.nn.functional as F
    from mvs_amd.jdacs.models.module import ConvBnReLU
    torch.manual_seed(3)
    m = ConvBnReLU(4, 8).eval()
    x = torch.randn(1, 4, 6, 7)
    with torch.no_grad():
        w1, b1 = m._folded()
        assert m._folded()[0] is w1                      # cached
        assert float((F.conv2d(x, w1, b1, padding=1) - m.bn(m.conv(x))).abs().max()) < 1e-5
        m.bn.running_mean.add_(0.5)
        m.conv.weight.mul_(1.5)
        w2, b2 = m._folded()
        assert w2 is not w1
        assert float((F.conv2d(x, w2, b2, padding=1) - m.bn(m.conv(x))).abs().max()) < 1e-5
        sd = {k: v.clone() + 0.1 for k, v in m.state_dict().items() if v.dtype.is_floating_point}
        m.load_state_dict(sd, strict=False)
        w3, b3 = m._folded()
        assert float((F.conv2d(x, w3, b3, padding=1) - m.bn(m.conv(x))).abs().max()) < 1e-5


def test_folded_batchnorm_cache_is_derived_data():
    """the eval-mode folded (conv, BatchNorm) weights are dropped by train(), by .to()/.double() and are not carried by
    deepcopy / pickle; refold() drops them by hand (parameter updates through .data do not bump the version counters)."""
    import copy
    import pickle
    from mvs_amd.jdacs.models.module import ConvBnReLU
    m = ConvBnReLU(3, 8).eval()
    with torch.no_grad():
        w0, _ = m._folded()
    assert "_fold_cache" in m.__dict__
    assert "_fold_cache" not in copy.deepcopy(m).__dict__
    assert "_fold_cache" not in pickle.loads(pickle.dumps(m)).__dict__
    m.train()
    assert "_fold_cache" not in m.__dict__
    m.eval()
    with torch.no_grad():
        m._folded()
        m.conv.weight.data.mul_(2.0)          # invisible to the (data_ptr, version) key
        assert torch.equal(m._folded()[0], w0)
        m.refold()
        assert torch.allclose(m._folded()[0], 2.0 * w0)
    m.double()
    assert "_fold_cache" not in m.__dict__


def test_reset_weight_uses_forgets_forward_passes_without_backward():
    from mvs_amd import ops
    w = torch.nn.Parameter(torch.zeros(4, 4))
    ops._WEIGHT_USES.setdefault(0, {})[w.data_ptr()] = 2
    ops._WEIGHT_MULTI.setdefault(0, set()).add(w.data_ptr())
    ops.reset_weight_uses()
    assert not ops._WEIGHT_USES and not ops._WEIGHT_MULTI
