"""Input pipeline kernels against the oracle: the image leg bit-exact with Pillow + torch fp32
normalisation, the LiDAR leg bit-exact with numpy.  Unpinned by the reference (it holds no vectors for these)."""
import numpy as np
import pytest
import torch

from bevfusion_multimodal_3d_object_detection_amd import preprocess, synth
from oracle import ref_preprocess as rp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,size", [((90, 160), (45, 80)), ((97, 131), (44, 80)), ((30, 40), (64, 96)),
                                        ((448, 800), (448, 800)), ((900, 1600), (448, 800))])
def test_camera_preprocess_is_bit_exact(gpu, shape, size):
    n = 3 if shape[0] < 900 else 6
    rs = np.random.RandomState(shape[1])
    imgs = rs.randint(0, 256, (n, *shape, 3), dtype=np.uint8)
    imgs[0, : shape[0] // 3] = 255
    imgs[0, shape[0] // 3: shape[0] // 2] = 0
    ref = rp.camera_preprocess(imgs, size)
    out = preprocess.preprocess_camera_images(torch.from_numpy(imgs).cuda(), size)
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert torch.equal(out.cpu(), ref)


def test_camera_preprocess_batch_layout_and_errors(gpu):
    imgs = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (2, 6, 60, 100, 3), dtype=np.uint8)).cuda()
    out = preprocess.preprocess_camera_images(imgs, (30, 50))
    assert tuple(out.shape) == (2, 6, 3, 30, 50)                  # (B, cams, 3, h, w): what the detector takes
    flat = preprocess.preprocess_camera_images(imgs.view(12, 60, 100, 3), (30, 50))
    assert torch.equal(out.view(12, 3, 30, 50), flat)
    with pytest.raises(Exception):
        preprocess.preprocess_camera_images(imgs.float(), (30, 50))
    with pytest.raises(Exception):
        preprocess.preprocess_camera_images(imgs.cpu(), (30, 50))


@pytest.mark.parametrize("N,max_points,with_choice", [(0, 64, False), (1, 8, False), (5000, 35000, False),
                                                      (50000, 35000, False), (50000, 35000, True), (40000, 1000, True)])
def test_lidar_filter_pad_matches_numpy(gpu, N, max_points, with_choice):
    pts = torch.stack([synth.uniform((N,), 1, -60.0, 60.0), synth.uniform((N,), 2, -60.0, 60.0),
                       synth.uniform((N,), 3, -6.0, 4.0), synth.uniform((N,), 4, 0.0, 255.0)], 1) if N else torch.zeros(0, 4)
    if N > 10:
        pts[3, 0] = 51.2                                          # exactly on the border: dropped (strict <)
        pts[4, 1] = float("nan")
    p = pts.numpy()
    n_in = int(((p[:, 0] > -51.2) & (p[:, 0] < 51.2) & (p[:, 1] > -51.2) & (p[:, 1] < 51.2) & (p[:, 2] > -5.0) &
                (p[:, 2] < 3.0)).sum())
    choice = None
    if with_choice and n_in >= max_points:
        choice = np.random.RandomState(N).choice(n_in, max_points, replace=False)
    ref, n = rp.lidar_filter_pad(p, max_points, choice)
    out, cnt = preprocess.filter_pad_lidar(pts.cuda(), max_points, choice=None if choice is None else torch.from_numpy(choice))
    assert int(cnt) == n == n_in
    assert tuple(out.shape) == (max_points, 4)
    assert np.array_equal(out.cpu().numpy(), ref, equal_nan=True)


def test_preprocessed_frames_feed_the_detector(gpu):
    """uint8 frames + raw sweep -> preprocess -> detector, against the oracle fed with the oracle's preprocessing."""
    from bevfusion_multimodal_3d_object_detection_amd import fusion
    from oracle import ref_model
    from tests.conftest import rel_err
    rs = np.random.RandomState(5)
    frames = rs.randint(0, 256, (1, 2, 120, 200, 3), dtype=np.uint8)
    sweep = np.stack([rs.uniform(-60, 60, 3000), rs.uniform(-60, 60, 3000), rs.uniform(-6, 4, 3000),
                      rs.uniform(0, 255, 3000)], 1).astype(np.float32)
    m = fusion.create_detector("camera+lidar", "bev", "centernet", bev_h=50, bev_w=50)
    synth.fill_state_dict_(m, 21)
    ora = ref_model.make_detector("camera+lidar", 50, 50)
    ora.load_state_dict(m.state_dict())
    ora.eval()
    imgs_ref = rp.camera_preprocess(frames[0], (64, 96)).unsqueeze(0)
    pts_ref = torch.from_numpy(rp.lidar_filter_pad(sweep, 2048)[0]).unsqueeze(0)
    with torch.no_grad():
        ref = ora(imgs_ref, pts_ref, None)
    imgs = preprocess.preprocess_camera_images(torch.from_numpy(frames).cuda(), (64, 96))
    pts = preprocess.filter_pad_lidar(torch.from_numpy(sweep).cuda(), 2048)[0].unsqueeze(0)
    out = m.cuda().eval()(imgs, pts, None)
    for k in ref:
        assert rel_err(out[k].cpu(), ref[k]) <= 1e-4, k
