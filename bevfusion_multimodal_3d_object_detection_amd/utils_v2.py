"""Evaluation metrics of the reference's drivers: `compute_metrics(predictions, ground_truths)` ->
{"mAP", "NDS", "AP_per_class"} as ref src/utils_v2.py:94-205 computes them (centre-distance matching at 2 m, 11-point
interpolated AP per frame and class, mean translation / scale / orientation errors of the matched pairs).  Host code:
per (frame, class) it is a sequential greedy assignment over at most `max_detections` boxes.

Reference behaviour kept on purpose: APs are averaged per frame before per class; a class never seen scores 0;
`NDS = mean(5*mAP, 1-min(mATE/4,1), 1-min(mASE,1), 1-min(mAOE/pi,1))`; ties in the scores are ordered by
`np.argsort(-scores)`.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

CLASS_NAMES = ("car", "truck", "bus", "trailer", "construction_vehicle", "pedestrian", "motorcycle", "bicycle",
               "traffic_cone", "barrier")


def compute_center_distance_matrix(pred_boxes: np.ndarray, gt_boxes: np.ndarray) -> np.ndarray:
    """(N,>=2), (M,>=2) -> (N,M) Euclidean distance of the box centres in the ground plane (ref src/utils_v2.py:7-10)."""
    d = pred_boxes[:, None, :2] - gt_boxes[None, :, :2]
    return np.sqrt((d ** 2).sum(axis=2))


def _greedy_assign(distance_matrix: np.ndarray, pred_scores: np.ndarray, threshold: float):
    """Predictions in descending score order each take the nearest ground truth that is still free, if it lies within
    `threshold`.  Returns (order, gt index per rank or -1).  Both reference loops (ref src/utils_v2.py:13-37 and
    :52-72) are this assignment."""
    order = np.argsort(-pred_scores)
    taken = np.zeros(distance_matrix.shape[1], dtype=bool)
    got = np.full(order.shape[0], -1, dtype=np.int64)
    for rank, p in enumerate(order):
        if taken.all():
            break
        d = np.where(taken, np.inf, distance_matrix[p])
        g = int(np.argmin(d))
        if d[g] <= threshold:
            got[rank] = g
            taken[g] = True
    return order, got


def match_predictions_to_gt(distance_matrix: np.ndarray, pred_scores: np.ndarray, threshold: float = 2.0) -> List[Tuple[int, int]]:
    order, got = _greedy_assign(distance_matrix, pred_scores, threshold)
    return [(int(order[r]), int(g)) for r, g in enumerate(got) if g >= 0]


def calculate_ap(pred_boxes: np.ndarray, pred_scores: np.ndarray, gt_boxes: np.ndarray, distance_matrix: np.ndarray,
                 threshold: float = 2.0) -> float:
    """11-point interpolated average precision of one class in one frame (ref src/utils_v2.py:43-88)."""
    if len(pred_boxes) == 0 or len(gt_boxes) == 0:
        return 0.0
    _, got = _greedy_assign(distance_matrix, pred_scores, threshold)
    hit = (got >= 0).astype(np.float64)
    tp_cum, fp_cum = np.cumsum(hit), np.cumsum(1.0 - hit)
    recalls = tp_cum / len(gt_boxes)
    precisions = tp_cum / (tp_cum + fp_cum + 1e-10)
    ap = 0.0
    for t in np.linspace(0, 1, 11):
        above = precisions[recalls >= t]
        ap += (above.max() if len(above) > 0 else 0) / 11.0
    return ap


def _to_numpy(v):
    return v if isinstance(v, np.ndarray) else v.cpu().numpy()


def compute_metrics(predictions: List[Dict], ground_truths: List[Dict]) -> Dict[str, float]:
    aps = [[] for _ in CLASS_NAMES]
    ate, ase, aoe = [], [], []
    for pred, gt in zip(predictions, ground_truths):
        gb, gl = gt["boxes"], gt["labels"]
        if isinstance(gl, np.ndarray):                         # the reference drops padding labels of numpy GT only
            keep = gl >= 0
            gb, gl = gb[keep], gl[keep]
        pb, ps, pl = pred["boxes"], pred["scores"], pred["labels"]
        if len(gb) == 0 and len(pb) == 0:
            continue
        if not isinstance(pb, np.ndarray):
            pb, ps, pl = _to_numpy(pb), _to_numpy(ps), _to_numpy(pl)
        if not isinstance(gb, np.ndarray):
            gb, gl = _to_numpy(gb), _to_numpy(gl)
        for c in range(len(CLASS_NAMES)):
            cp, cs, cg = pb[pl == c], ps[pl == c], gb[gl == c]
            if len(cg) == 0 and len(cp) == 0:
                continue
            if len(cg) == 0 or len(cp) == 0:
                aps[c].append(0.0)
                continue
            dist = compute_center_distance_matrix(cp, cg)
            aps[c].append(calculate_ap(cp, cs, cg, dist, threshold=2.0))
            for pi, gi in match_predictions_to_gt(dist, cs, threshold=2.0):
                p, g = cp[pi], cg[gi]
                ate.append(np.linalg.norm(p[:2] - g[:2]))
                ase.append(np.mean(np.abs(p[3:6] - g[3:6]) / (g[3:6] + 1e-6)))
                turn = p[6] - g[6]
                aoe.append(abs(np.arctan2(np.sin(turn), np.cos(turn))))
    class_aps = [float(np.mean(a)) if len(a) > 0 else 0.0 for a in aps]
    m_ap = float(np.mean(class_aps))
    m_ate = float(np.mean(ate)) if ate else 1.0
    m_ase = float(np.mean(ase)) if ase else 1.0
    m_aoe = float(np.mean(aoe)) if aoe else 1.0
    nds = np.mean([5 * m_ap, 1 - min(m_ate / 4.0, 1.0), 1 - min(m_ase / 1.0, 1.0), 1 - min(m_aoe / np.pi, 1.0)])
    return {"mAP": float(m_ap), "NDS": float(nds),
            "AP_per_class": {CLASS_NAMES[i]: float(class_aps[i]) for i in range(len(CLASS_NAMES))}}


def _metrics_report(metrics: Dict) -> List[str]:
    """The lines of the drivers' metrics report: a header, mAP and NDS to four decimals, then one line per class with
    the class name left-justified in 20 columns."""
    lines = ["===== Evaluation Metrics =====", f"mAP : {metrics['mAP']:.4f}", f"NDS : {metrics['NDS']:.4f}", "",
             "--- AP Per Class ---"]
    lines += [f"{name:20s}: {ap:.4f}" for name, ap in metrics["AP_per_class"].items()]
    return lines


def save_and_print_metrics(metrics: dict, save_path: str = "metrics_output.txt"):
    """ref src/utils_v2.py:208-233 (called by src/train_detect.py and src/eval.py after `compute_metrics`): print the
    report to stdout and write the same report to `save_path` (byte-identical to the reference's file:
    tests/golden/metrics_report_*.txt are minted from it)."""
    lines = _metrics_report(metrics)
    print("\n" + "\n".join(lines))
    with open(save_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"\nMetrics saved to {save_path}")
