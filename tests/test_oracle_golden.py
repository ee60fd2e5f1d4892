"""The oracle (oracle/*.py) against outputs of the imported reference (tests/golden/*.npz).

CPU only.  Float tensors: <= 2e-6 rel (same torch ops; a different host CPU may pick another
oneDNN kernel, so not asserted bit-exact).  Integer tensors (ind, masks, labels): bit-exact.
"""
import numpy as np
import pytest
import torch

from bevfusion_multimodal_3d_object_detection_amd import synth
from oracle import ref_model, ref_targets
from tests.conftest import load_golden, rel_err
from tests.golden import cases

TOL = 2e-6


def _check(out, gold, keys=None):
    for k in (keys or out.keys()):
        g = gold[k]
        o = out[k].detach().numpy() if isinstance(out[k], torch.Tensor) else np.asarray(out[k])
        assert o.shape == g.shape, (k, o.shape, g.shape)
        if np.issubdtype(g.dtype, np.integer) or g.dtype == np.bool_:
            assert np.array_equal(o, g), k
        else:
            assert rel_err(o, g) <= TOL, (k, rel_err(o, g))


@pytest.mark.parametrize("c", cases.DETECTOR_CASES, ids=lambda c: c["name"])
@torch.no_grad()
def test_detector(c):
    m = ref_model.make_detector(c["modality"], c["bev_h"], c["bev_w"])
    synth.fill_state_dict_(m, c["seed"])
    m.eval()
    gold = load_golden("detector_" + c["name"])
    assert sum(p.numel() for p in m.parameters()) == int(gold["n_params"])
    assert len(m.state_dict()) == int(gold["n_state"])
    imgs, pts, radars = cases.detector_inputs(c)
    _check(m(imgs, pts, radars or None), gold, ("heatmap", "offset", "size", "rot", "vel"))


def test_param_counts_match_published():
    """demo.ipynb:419,519 -- 52,398,483 (camera+lidar) and 55,197,715 (camera+lidar+radar)."""
    gold = load_golden("param_counts")
    assert int(gold["camera_lidar"]) == 52_398_483 and int(gold["camera_lidar_radar"]) == 55_197_715
    for mod, key in (("camera+lidar", "camera_lidar"), ("camera+lidar+radar", "camera_lidar_radar"),
                     ("camera_only", "camera_only")):
        assert sum(p.numel() for p in ref_model.make_detector(mod).parameters()) == int(gold[key])


def test_state_dict_keys_match_reference():
    import os
    from tests.conftest import GOLDEN
    want = open(os.path.join(GOLDEN, "state_dict_keys_clr.txt")).read().split("\n")[:-1]
    got = sorted(f"{k}:{tuple(v.shape)}" for k, v in ref_model.make_detector("all").state_dict().items())
    assert got == want


@torch.no_grad()
def test_camera_encoder():
    c = cases.CAMERA_ENCODER_CASE
    m = ref_model.CameraEncoder(); synth.fill_state_dict_(m, c["seed"]); m.eval()
    _check({"out": m(synth.normal(c["shape"], c["seed"] + 1))}, load_golden("camera_encoder"))


@torch.no_grad()
def test_pointnet():
    c = cases.POINTNET_CASE
    m = ref_model.PointMLPMax(c["cin"], [64, 128, 256, 512, 1024]); synth.fill_state_dict_(m, c["seed"]); m.eval()
    _check({"out": m(cases.pointnet_input(c))}, load_golden("pointnet"))


@pytest.mark.parametrize("method", ["concat", "max", "mean"])
@torch.no_grad()
def test_radar(method):
    c = cases.RADAR_CASE
    m = ref_model.MultiRadar(7, 256, c["num_radars"], method); synth.fill_state_dict_(m, c["seed"]); m.eval()
    _check({"out": m(cases.radar_input(c))}, load_golden("radar_" + method))


@torch.no_grad()
def test_vfe():
    c = cases.VFE_CASE
    m = ref_model.VFE(c["cin"], c["cout"]); synth.fill_state_dict_(m, c["seed"]); m.eval()
    _check({"out": m(synth.normal(c["shape"], c["seed"] + 1))}, load_golden("vfe"))


@pytest.mark.parametrize("c", cases.FUSION_CASES, ids=lambda c: c["name"])
@torch.no_grad()
def test_fusion(c):
    m = ref_model.BEVFusion(c["cam"], c["lid"], c["rad"], bev_h=c["bev_h"], bev_w=c["bev_w"])
    synth.fill_state_dict_(m, c["seed"]); m.eval()
    _check({"out": m(*cases.fusion_inputs(c))}, load_golden("fusion_" + c["name"]))


@torch.no_grad()
def test_head():
    c = cases.HEAD_CASE
    torch.manual_seed(0)
    m = ref_model.CenterHead()
    x = synth.normal(c["shape"], c["seed"] + 1)
    o = m(x)["heatmap"]                      # default init: sigmoid(-ln 99) = 0.01 everywhere
    assert 0.0099 < float(o.min()) and float(o.max()) < 0.0101
    synth.fill_state_dict_(m, c["seed"]); m.eval()
    _check(m(x), load_golden("head"))


@pytest.mark.parametrize("c", cases.TARGET_CASES, ids=lambda c: c["name"])
def test_targets_and_loss(c):
    boxes, labels = cases.target_inputs(c)
    t = ref_targets.make_targets(boxes, labels, bev_size=c["bev_size"])
    gold = load_golden("targets_" + c["name"])
    for k in ("ind", "mask", "reg_mask"):                      # the integer grid-index pin: bit-exact
        assert np.array_equal(t[k].numpy(), gold[k]), k
    for k in ("heatmap", "offset", "size", "rot", "vel", "target_offset", "target_size", "target_rot", "target_vel"):
        assert np.array_equal(t[k].numpy(), gold[k]), k       # same float ops in the same order: exact
    losses = ref_targets.centernet_loss(cases.loss_predictions(c), t)
    _check(losses, load_golden("loss_" + c["name"]))


@pytest.mark.parametrize("c", cases.DECODE_CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("tag,vox", [("ct", 2.048), ("fd", 0.512)])
def test_decode(c, tag, vox):
    gold = load_golden(f"decode_{tag}_{c['name']}")
    dets = ref_targets.decode(cases.decode_predictions(c), c["thresh"], c["K"], voxel_size=vox)
    for b, d in enumerate(dets):
        for k, v in d.items():
            g = gold[f"{k}_{b}"]
            assert tuple(v.shape) == g.shape, (k, b)
            if k == "labels":
                assert np.array_equal(v.numpy(), g) and (g == 0).all()     # the reference's label bug
            elif v.numel():
                assert rel_err(v, g) <= TOL, (k, b)
