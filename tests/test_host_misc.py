"""Host-side logic that needs no GPU: launch-limit chunking, conv-mode switch, checkpoint dict layout, refusal of CPU
tensors by the new entry points."""
import pytest
import torch

from bevfusion_multimodal_3d_object_detection_amd import _lib as L
from bevfusion_multimodal_3d_object_detection_amd import checkpoint, engine, fusion, preprocess, synth, training


def test_image_chunks_are_balanced_and_respect_the_limit(monkeypatch):
    assert training._image_chunk(48, 225 * 400 * 64) == 48                       # config 4 fits one launch
    assert training._image_chunk(96, 225 * 400 * 64) == 48                       # 2.2 GB of layer-1 input: two chunks
    monkeypatch.setattr(training, "BUF_LIMIT", 1000 * 4)
    for n in range(1, 40):
        per = training._image_chunk(n, 300)
        assert 1 <= per <= 3 or n <= 3
        assert per * 300 * 4 <= 1000 * 4
        assert -(-n // per) == -(-n // 3)                                         # as few launches as the limit allows


def test_conv_mode_switch_validates_and_changes_engine_signature():
    default = engine.conv_mode()
    assert default == "wino"                     # fused fp32 Winograd for the 3x3 / stride 1 layers
    with pytest.raises(ValueError):
        engine.set_conv_mode("bf16x9")
    m = fusion.create_detector("camera_only", "bev", "centernet", bev_h=16, bev_w=16)
    eng = m.camera_encoder._eng()
    before = eng._signature()
    engine.set_conv_mode("f32x3")
    try:
        assert eng._signature() != before                                         # next forward repacks
    finally:
        engine.set_conv_mode(default)
    assert eng._signature() == before


def test_checkpoint_dict_layout_on_cpu(tmp_path):
    m = fusion.create_detector("camera_only", "bev", "centernet", bev_h=16, bev_w=16)
    synth.fill_state_dict_(m, 3)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01)
    path = tmp_path / "best_model.pth"
    checkpoint.save_checkpoint(path, m, opt, epoch=7, config={"use_camera": True}, best_map=0.5)
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {"epoch", "model_state_dict", "optimizer_state_dict", "config", "best_map"}     # ref train_detect.py:786-792
    assert list(raw["model_state_dict"]) == list(m.state_dict())
    m2 = fusion.create_detector("camera_only", "bev", "centernet", bev_h=16, bev_w=16)
    ck = checkpoint.load_checkpoint(path, m2)
    assert ck["epoch"] == 7 and ck["best_map"] == 0.5
    for a, b in zip(m.state_dict().values(), m2.state_dict().values()):
        assert torch.equal(a, b)
    torch.save({"weights": 1}, tmp_path / "other.pth")
    with pytest.raises(KeyError):
        checkpoint.load_checkpoint(tmp_path / "other.pth")


def test_new_entry_points_refuse_cpu_tensors():
    with pytest.raises(L.BevfError):
        preprocess.preprocess_camera_images(torch.zeros(1, 8, 8, 3, dtype=torch.uint8))
    with pytest.raises(L.BevfError):
        preprocess.filter_pad_lidar(torch.zeros(10, 4))
    with pytest.raises(L.BevfError):
        preprocess.preprocess_camera_images(torch.zeros(1, 8, 8, 3))                          # not uint8


def test_oracle_dense_scatter_is_the_references_index_assignment():
    """oracle/ref_voxelize.dense_scatter against the statement it restates (ref src/encoders.py:407-410,
    `feature_grid[b, :, c0, c1, c2] = features.T`) on unique coordinates, where that statement is well defined."""
    from oracle import ref_voxelize
    B, C, D, H, W = 2, 5, 2, 6, 7
    perm = torch.stack([torch.randperm(D * H * W, generator=torch.Generator().manual_seed(b))[:40] for b in range(B)])
    coords = torch.stack([perm // (H * W), (perm // W) % H, perm % W], dim=2)
    feats = synth.normal((B, 40, C), 12)
    grid = torch.zeros(B, C, D, H, W)
    for b in range(B):
        c = coords[b].long()
        grid[b, :, c[:, 0], c[:, 1], c[:, 2]] = feats[b].T
    assert torch.equal(ref_voxelize.dense_scatter(feats, coords, (D, H, W)), grid)


def test_train_mode_dispatch_of_the_modules_is_host_logic_that_fails_loudly_without_a_gpu():
    """forward() of an encoder / fusion / head picks the train-mode tape by the BatchNorm flags alone; with CPU tensors both
    routes refuse (no CPU fallback); an eval-mode BatchNorm inside a training module is "frozen" (running statistics, constants in the backward)."""
    from bevfusion_multimodal_3d_object_detection_amd import encoders, centernet_target
    enc = encoders.PointNetLiDAREncoder(input_channels=4, feat_dim=1024)             # a fresh module is in train mode
    assert enc.training and training.any_bn_training(enc)
    with pytest.raises(L.BevfError, match="MI355X only"):
        enc(torch.zeros(1, 8, 4))
    enc.eval()
    assert not training.any_bn_training(enc)
    with pytest.raises(L.BevfError, match="MI355X only"):
        enc(torch.zeros(1, 8, 4))
    enc.train()
    enc.bn3.eval()
    assert training.bn_is_frozen(enc.bn3) and not training.bn_is_frozen(enc.bn2) and training.wants_train_path(enc)
    cam = encoders.ResNetCameraEncoder(backbone="resnet18", pretrained=False, freeze_bn=True)
    assert cam.training and not training.any_bn_training(cam)                        # frozen BatchNorm everywhere: running statistics
    assert training.wants_train_path(cam)                                            # ... but the weights still train: the tape
    with torch.no_grad():
        assert not training.wants_train_path(cam)                                    # nothing to record, nothing to update: eval engine
    with pytest.raises(L.BevfError, match="no gradient path"):
        training._no_input_grad(torch.zeros(2, requires_grad=True), "the camera images")
    if not torch.cuda.is_available():
        with pytest.raises(L.BevfError, match="no 'cuda' device"):
            centernet_target.example_usage()
