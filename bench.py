#!/usr/bin/env python3
"""bench.py -- BEV frames/sec of the detector forward on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config 2|3|1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one forward of `FlexibleMultiModal3DDetector` over a batch of B synthetic frames
(config 2 of BASELINE.json by default: 6 x 900x1600 cameras + 35k-point LiDAR, 128x128 BEV,
fp32, random-init weights), inputs resident in HBM, head tensors as outputs.  Frames shard
over ranks as independent replicas (inference has no collective) -> weak scaling.
Rank 0 prints ONE JSON line with the whole-job frames/s, the live roofline of the dominant
kernel (conv_igemm_f32, HIP events on the launch stream inside the timed region) and the CPU
baseline (the oracle on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bevfusion_multimodal_3d_object_detection_amd import engine, fusion, replicas, synth   # noqa: E402

CONFIGS = {
    # BASELINE.json configs[0..2]; (modality, cams, H, W, points, radars, bev)
    1: dict(name="camera_only 6x448x800, BEV 128x128 (reference-runnable sanity shape)",
            modality="camera_only", cams=6, h=448, w=800, points=0, radars=0, bev=128),
    2: dict(name="camera+LiDAR, 6x900x1600 images + 35k-point sweep, BEV 128x128",
            modality="camera+lidar", cams=6, h=900, w=1600, points=35000, radars=0, bev=128),
    3: dict(name="camera+LiDAR+radar, 6x900x1600 + 35k points + 5x125 radar, BEV 128x128",
            modality="camera+lidar+radar", cams=6, h=900, w=1600, points=35000, radars=5, bev=128),
    5: dict(name="bandwidth-stress: camera+LiDAR, 6x900x1600 + 120k-point 10-sweep LiDAR, BEV 256x256",
            modality="camera+lidar", cams=6, h=900, w=1600, points=120000, radars=0, bev=256),
    # config 4: the reference's training step (src/train_detect.py: images resized to 448x800, BEV 50x50 targets)
    4: dict(name="train step camera+LiDAR, 6x448x800 + 35k points, BEV 50x50, 20 GT boxes/frame, AdamW + clip 10",
            modality="camera+lidar", cams=6, h=448, w=800, points=35000, radars=0, bev=50),
}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def build_model(cfg, seed=0):
    model = fusion.create_detector(cfg["modality"], "bev", "centernet", bev_h=cfg["bev"], bev_w=cfg["bev"])
    synth.fill_state_dict_(model, seed)
    return model.eval()


def make_inputs(cfg, batch, seed, dev):
    imgs, pts, radars = synth.frame_inputs(batch, cfg["cams"], cfg["h"], cfg["w"], cfg["points"], 4,
                                           cfg["radars"], 125, 7, seed=seed)
    return (imgs.to(dev) if imgs is not None else None, pts.to(dev) if pts is not None else None,
            [r.to(dev) for r in radars] if radars else None)


def pmc_traffic(config: int, batch: int):
    """HBM-side bytes per conv launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, see
    tools/pmc_summary.py); only valid for the configuration the profile was taken on."""
    path = os.path.join(ROOT, "profiles", f"r01_pmc_traffic_config{config}_b{batch}.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    n = sum(v["launches"] for k, v in d.items() if k.startswith("conv_igemm"))
    tot = sum(v["launches"] * v["hbm_bytes_per_launch"] for k, v in d.items() if k.startswith("conv_igemm"))
    return (tot / n, os.path.relpath(path, ROOT)) if n else None


def host_cores() -> int:
    """Cores this process may really use: min(affinity, cgroup cpu.max quota) -- the GPU box reports 256
    logical CPUs but grants a 16-CPU share; oversubscribing it makes the CPU baseline meaninglessly slow."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cfg, state_dict, budget_s=12.0):
    """The oracle (CPU restatement, oracle/ref_model.py) on the host cores, B=1, same synthetic frame."""
    from oracle import ref_model
    cores = host_cores()
    torch.set_num_threads(cores)
    ora = ref_model.make_detector(cfg["modality"], cfg["bev"], cfg["bev"])
    ora.load_state_dict(state_dict)
    ora.eval()
    imgs, pts, radars = synth.frame_inputs(1, cfg["cams"], cfg["h"], cfg["w"], cfg["points"], 4,
                                           cfg["radars"], 125, 7, seed=0x5EED + 2000)
    with torch.no_grad():
        ora(imgs, pts, radars or None)                      # warm-up frame
        n, t0 = 0, time.perf_counter()
        while n < 2 or (time.perf_counter() - t0 < budget_s and n < 8):
            ora(imgs, pts, radars or None)
            n += 1
        dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} frames at B=1 after 1 warm-up, oracle/ref_model.py (PyTorch-CPU fp32), same config")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="frames per step per GPU (default 8; 2 for config 5, whose B=8 activations pass the 2 GiB buffer limit)")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="train = BASELINE config 4: fwd + targets + loss + bwd + grad all-reduce + clip + AdamW")
    ap.add_argument("--dtype", choices=["fp32", "bf16", "f32x3"], default="fp32",
                    help="fp32 (default): exact fp32 MFMA.  bf16 = BASELINE configs 3/5: bf16 storage, fp32 accumulate.  "
                         "f32x3: fp32 storage, convolutions through an exact 3-way bf16 split on the bf16 MFMA "
                         "(fp32-level error, not bit-identical to fp32; opt-in, never the headline)")
    ap.add_argument("--graph", action="store_true", help="replay the inference forward as one hipGraph (small batches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true", help="skip the per-launch HIP-event brackets")
    args = ap.parse_args()

    rank, local_rank, world = replicas.rank_world()
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # one process per GPU.  BEVF_DIST_BACKEND=gloo + fewer GPUs than ranks is a control-flow rehearsal only
    # (ranks then share a device); the driver's runs use the default: RCCL, one GPU per rank.
    backend = os.environ.get("BEVF_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = replicas.init(backend, dev)             # RCCL; only for the barrier and the MAX of the elapsed time
    if args.mode == "train":
        args.config = 4
    cfg = CONFIGS[args.config]
    if args.batch is None:
        args.batch = 2 if args.config == 5 else 8

    model_cpu = build_model(cfg)
    state = {k: v.clone() for k, v in model_cpu.state_dict().items()}
    model = model_cpu.to(dev)
    if args.dtype == "f32x3":
        if args.mode == "train":
            raise SystemExit("training runs in exact fp32")
        engine.set_conv_mode("f32x3")
    if args.dtype == "bf16":
        if args.mode == "train":
            raise SystemExit("training runs in fp32 (the reference has no mixed precision, SURVEY.md 5)")
        model = model.bfloat16()
    inputs = make_inputs(cfg, args.batch, replicas.frame_seed(0x5EED, args.config, rank), dev)

    if args.mode == "train":
        from bevfusion_multimodal_3d_object_detection_amd import centernet_target as ct
        from bevfusion_multimodal_3d_object_detection_amd import training
        model.train()
        boxes, labels = synth.gt_boxes(args.batch, 20, seed=replicas.frame_seed(0x5EED, args.config, rank))
        gt = {"gt_boxes": boxes.to(dev), "gt_labels": labels.to(dev)}
        crit = ct.CenterNetLoss()
        # ref train_detect.py:725-741 (AdamW lr 1e-4 wd 0.01) and :431 (clip_grad_norm_ 10), clip folded into the update
        opt = training.FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.01, max_grad_norm=10.0)

        if dist is not None:          # DP: gradient all-reduce (RCCL) issued from inside the backward, overlapped with it
            training.set_grad_reducer(replicas.GradReducer(dist))

        def step():
            pred = model(*inputs)
            tgt = ct.prepare_centernet_targets(gt, dev)
            losses = crit(pred, tgt)
            opt.zero_grad()
            losses["total_loss"].backward()
            opt.step()
            return {k: v.detach() for k, v in losses.items()}
    elif args.graph:
        graphed = model.make_graphed(*inputs)
        args.no_kernel_timer = True                     # per-launch event brackets cannot live inside a captured graph

        def step():
            return graphed(*inputs)
    else:
        def step():
            return model(*inputs)

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in out.values()), "non-finite head output"

    # Inference: the per-launch HIP-event brackets (37 launches a step) ride inside the timed region.  Training issues
    # ~700 launches a step and the brackets' host cost would distort `value`, so its roofline is taken over two extra,
    # untimed steps right after the timed region.
    timer = None if args.no_kernel_timer else engine.KernelTimer()
    timer_in_region = timer is not None and args.mode == "infer"
    engine.set_timer(timer if timer_in_region else None)
    replicas.barrier(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    replicas.barrier(dist)
    elapsed = time.perf_counter() - t0
    engine.set_timer(None)
    elapsed = replicas.max_over_ranks(elapsed, dist, dev)
    timer_steps, timer_elapsed = args.steps, elapsed
    if timer is not None and not timer_in_region:
        engine.set_timer(timer)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        timer_steps, timer_elapsed = 2, time.perf_counter() - t1
        engine.set_timer(None)

    if rank == 0:
        line = {
            "metric": f"BEV frames/sec (6-cam+LiDAR, {cfg['bev']}x{cfg['bev']} BEV)",
            "value": replicas.aggregate_fps(args.batch * args.steps, world, elapsed), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": cfg["name"], "batch_per_gpu": args.batch,
                       "parallelism": f"replicas x{world}" if args.mode == "infer" else f"dp{world}",
                       "weights": "random-init (synthetic, seeded)", "launch": "hipGraph replay" if args.graph else "eager",
                       "mode": "inference forward -> 5 head tensors" if args.mode == "infer"
                       else "training step: fwd (train-mode BN) + targets + loss + bwd + grad all-reduce + clip + AdamW"},
        }
        if args.mode == "train":
            line["metric"] = "training frames/sec (camera+LiDAR, per-GPU batch 8)"
        if timer is not None:
            tot = timer.totals()
            conv = tot.get("conv_igemm_f32")
            if conv and args.mode == "train":                          # forward + data-gradient + weight-gradient GEMMs
                for extra in ("conv_dgrad_f32", "conv_wgrad_f32"):
                    e = tot.get(extra)
                    if e:
                        conv = {k: conv[k] + e[k] for k in conv}
            if conv:
                ach = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
                # f32x3 spends six bf16 MFMA products per algorithmic multiply-add: its ceiling is a sixth of the bf16 peak
                peak = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS,
                        "f32x3": PEAK_BF16_MFMA_TFLOPS / 6.0}[args.dtype]
                kname = {"fp32": "conv_igemm_f32", "bf16": "conv_igemm_bf16", "f32x3": "conv_split_f32x3"}[args.dtype]
                line["roofline"] = {"kernel": kname if args.mode == "infer" else "conv_igemm_f32+conv_wgrad_f32",
                                    "bound": "mfma", "achieved": ach,
                                    "traffic_source": None,
                                    "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                    "traffic": None, "launches_per_step": conv["launches"] / timer_steps,
                                    "avg_launch_ms": conv["ms"] / conv["launches"],
                                    "gflop_per_step": conv["flops"] / timer_steps / 1e9,
                                    "share_of_step": conv["ms"] / (1e3 * timer_elapsed),
                                    "measured_over": "the timed region" if timer_in_region else "2 untimed steps after the timed region"}
                tr = pmc_traffic(args.config, args.batch) if args.dtype == "fp32" and args.mode == "infer" else None
                if tr is not None:
                    line["roofline"]["traffic"], line["roofline"]["traffic_source"] = tr
            pool = tot.get("bev_pool")
            if pool:
                gbs = pool["bytes"] / (pool["ms"] * 1e-3) / 1e9
                line["roofline_bev_pool"] = {"kernel": "cam_mean+bilinear_nhwc", "bound": "hbm", "achieved": gbs,
                                             "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                             "traffic": None, "mb_per_step": pool["bytes"] / timer_steps / 1e6}
            stem = tot.get("stem_conv7x7_f32")
            if stem:
                line["stem_tflops"] = stem["flops"] / (stem["ms"] * 1e-3) / 1e12
        if world == 1 and not args.no_cpu_baseline and args.mode == "infer":
            line["cpu_baseline"] = cpu_baseline(cfg, state)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
