"""TEST INFRASTRUCTURE -- sequential restatement of hard voxelisation.

There is no reference implementation to follow (the reference has no voxel assignment): this
is the standard deterministic algorithm (VoxelNet / mmdet3d `hard_voxelize` semantics) written as a plain
loop, and it is what the HIP voxeliser is held bit-exact against.  **Parity unpinned by the reference.**
"""
import numpy as np
import torch


def hard_voxelize(points: torch.Tensor, pc_range, voxel_size, max_points: int, max_voxels: int):
    pts = points.numpy().astype(np.float32)
    B, N, C = pts.shape
    lo = np.asarray(pc_range[:3], dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    grid = np.round((np.asarray(pc_range[3:], dtype=np.float32) - lo) / vs).astype(np.int64)      # (gx, gy, gz)
    feats = np.zeros((B, max_voxels, max_points, C), np.float32)
    coords = np.zeros((B, max_voxels, 3), np.int64)
    npts = np.zeros((B, max_voxels), np.int32)
    nvox = np.zeros(B, np.int32)
    for b in range(B):
        c = np.floor((pts[b, :, :3] - lo) / vs)                       # float32 arithmetic, like the kernel
        ok = np.all((c >= 0) & (c < grid.astype(np.float32)), axis=1)
        ci = c.astype(np.int64)
        table = {}
        for n in range(N):
            if not ok[n]:
                continue
            key = (ci[n, 2], ci[n, 1], ci[n, 0])                      # (z, y, x)
            v = table.get(key)
            if v is None:
                v = len(table)
                table[key] = v
                if v < max_voxels:
                    coords[b, v] = key
            if v < max_voxels and npts[b, v] < max_points:
                feats[b, v, npts[b, v]] = pts[b, n]
                npts[b, v] += 1
        nvox[b] = min(len(table), max_voxels)
    return torch.from_numpy(feats), torch.from_numpy(coords), torch.from_numpy(npts), torch.from_numpy(nvox)


def dense_scatter(features: torch.Tensor, coords: torch.Tensor, grid, num_voxels=None) -> torch.Tensor:
    """ref src/encoders.py:399-410 -- `feature_grid[b, :, c0, c1, c2] = features.T`, written row by row so that the last
    row naming a cell wins (the reference's advanced-index assignment leaves the order of duplicates unspecified; its
    sequential reading is what the kernel pins).  num_voxels: only rows below it take part (not in the reference)."""
    B, Nv, C = features.shape
    D, H, W = grid
    out = torch.zeros(B, C, D, H, W)
    for b in range(B):
        n = Nv if num_voxels is None else int(num_voxels[b])
        for v in range(n):
            z, y, x = (int(t) for t in coords[b, v])
            if 0 <= z < D and 0 <= y < H and 0 <= x < W:
                out[b, :, z, y, x] = features[b, v]
    return out
