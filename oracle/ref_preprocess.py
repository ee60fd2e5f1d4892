"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of the reference's input pipeline
(ref src/train_detect.py:123-189): the image leg runs through Pillow itself, exactly as torchvision's
`T.Resize` + `T.ToTensor` + `T.Normalize` do for a PIL image (torchvision is absent from this image; its PIL path is
`img.resize(size[::-1], Image.BILINEAR)` followed by `pic -> float / 255` and `(t - mean) / std`).
Pinned by: the installed Pillow (image leg) and numpy (LiDAR leg) -- the reference's own tests hold no vectors for
these functions ("parity unpinned by the reference")."""
import numpy as np
import torch
from PIL import Image

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def resize_u8(img_hwc: np.ndarray, size=(448, 800)) -> np.ndarray:
    """ref src/train_detect.py:128 T.Resize((448, 800)) on a PIL image."""
    return np.asarray(Image.fromarray(img_hwc, "RGB").resize((size[1], size[0]), Image.BILINEAR))


def camera_preprocess(imgs_u8: np.ndarray, size=(448, 800)) -> torch.Tensor:
    """(n,H,W,3) uint8 -> (n,3,h,w) fp32; ref src/train_detect.py:127-143."""
    out = []
    mean = torch.tensor(MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(3, 1, 1)
    for im in imgs_u8:
        r = torch.from_numpy(resize_u8(im, size).copy()).permute(2, 0, 1).contiguous()
        t = r.to(torch.float32).div(255)                        # T.ToTensor
        out.append((t - mean) / std)                            # T.Normalize: sub_ then div_
    return torch.stack(out)


def lidar_filter_pad(points: np.ndarray, max_points: int, choice=None):
    """ref src/train_detect.py:150-159, 181-189; `choice` stands for np.random.choice(N, max_points, replace=False)."""
    m = (points[:, 0] > -51.2) & (points[:, 0] < 51.2) & (points[:, 1] > -51.2) & (points[:, 1] < 51.2) & \
        (points[:, 2] > -5.0) & (points[:, 2] < 3.0)
    p = points[m]
    n = p.shape[0]
    if n >= max_points:
        p = p[choice] if choice is not None else p[:max_points]
    else:
        p = np.concatenate([p, np.zeros((max_points - n, points.shape[1]), dtype=points.dtype)], axis=0)
    return p, n
