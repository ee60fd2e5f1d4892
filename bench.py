#!/usr/bin/env python3
"""bench.py -- BEV frames/sec of the detector forward on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config 2|3|1|5] [--mode infer|train]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one forward of `FlexibleMultiModal3DDetector` over a batch of B synthetic frames
(config 2 of BASELINE.json by default: 6 x 900x1600 cameras + 35k-point LiDAR, 128x128 BEV,
fp32, random-init weights), inputs resident in HBM, head tensors as outputs.  Frames shard
over ranks as independent replicas (inference has no collective) -> weak scaling.

`--gpus N` with N > 1 and no torchrun environment: this process launches N ranks itself
(`python -m torch.distributed.run ... bench.py --gpus N ...`, one per GPU, rendezvous on 127.0.0.1)
BEFORE it touches the GPU, forwards their output and exits with their code.  Every rank
checks that the world it joined has exactly N ranks and fails otherwise.

Rank 0 prints ONE JSON line: the whole-job frames/s of the headline configuration, the live
roofline of its dominant kernels (wino_f32 + conv_igemm_f32, HIP events on the launch stream
inside the timed region), the CPU baseline (the oracle on the host cores, N=1 only) and, at
N=1 under `extra.configs`, short legs of the other BASELINE configs (3: bf16 full fusion,
5: bf16 bandwidth-stress, 4: the training step), each with its own live roofline.  At N>1
the line leaves right after the replica headline; the data-parallel training leg (gradient
all-reduce over RCCL) runs after it and reports on stderr as `[bench extra] {...}`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[0..4]; (modality, cams, H, W, points, radars, bev)
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
PMC_PROFILE_ROUNDS = ("r03", "r02", "r01")
DEFAULT_CONV = "wino"
WINO_EXECUTED = 16.0 / 36.0       # Winograd F(2x2,3x3): 16 multiplies on the matrix pipe per 36 of the direct convolution
# KernelTimer span -> (kernel the span's launches run on, or None = the implicit-GEMM kernel of the leg's dtype / conv mode;
#                      share of the span's ALGORITHMIC (direct-convolution) FLOPs the matrix pipe really executes)
MFMA_SPANS = {
    "conv_wino_f32": ("wino_f32", WINO_EXECUTED),
    "conv_dgrad_wino_f32": ("wino_f32", WINO_EXECUTED),          # training: the data gradient runs on the forward kernel
    "conv_wgrad_wino_f32": ("wino_wgrad_f32", WINO_EXECUTED),
    "conv_igemm_f32": (None, 1.0),
    "conv3x3_bf16": ("conv3x3_bf16", 1.0),                       # bf16 models: the 3x3 / stride 1 layers (csrc/conv3x3_bf16.hip)
    "conv_dgrad_f32": (None, 1.0),
    "conv_wgrad_f32": ("conv_wgrad_f32", 1.0),
    "stem_conv7x7_f32": ("stem", 1.0),
    "pointnet_front_f32": ("pointnet_front", 1.0),
}


def parse_args(argv=None):
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
    ap.add_argument("--conv", choices=["wino", "f32", "wino_x3"], default=DEFAULT_CONV,
                    help="fp32 convolution kernels: f32 = exact implicit GEMM (v_mfma_f32_32x32x2_f32) everywhere; wino = fused fp32 "
                         "Winograd F(2x2,3x3) for the 3x3 / stride 1 layers (fp32 products and accumulation, 2.25x fewer MFMA FLOPs, "
                         "a few 1e-7 relative from the exact kernel), exact implicit GEMM for the rest; wino_x3 = opt-in: Winograd as "
                         "above and every other layer through the exact three-plane bf16 split of --dtype f32x3 (fp32-level error)")
    ap.add_argument("--graph", action="store_true", help="replay the inference forward as one hipGraph (small batches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true", help="skip the per-launch HIP-event brackets")
    ap.add_argument("--extras", choices=["auto", "all", "none"], default="auto",
                    help="extra.configs legs after the headline: auto = configs 3 (bf16), 5 (bf16) and 4 (train) at N=1, "
                         "the data-parallel training leg only at N>1 (it is the one with a collective); none = headline only")
    ap.add_argument("--extra-steps", type=int, default=8)
    return ap.parse_args(argv)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks_if_needed(args, argv) -> None:
    """`--gpus N` without a torchrun environment: start N ranks as children of this process (which has made no GPU
    call and never will) and exit with their code.  A torchrun environment of a different size is an error."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher made WORLD_SIZE={env_world} ranks; "
                             "pass --gpus equal to --nproc-per-node")
        return
    if args.gpus <= 1:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def build_model(cfg, seed=0):
    from bevfusion_multimodal_3d_object_detection_amd import fusion, synth
    model = fusion.create_detector(cfg["modality"], "bev", "centernet", bev_h=cfg["bev"], bev_w=cfg["bev"])
    synth.fill_state_dict_(model, seed)
    return model.eval()


class InputCache:
    """Synthetic frames regenerated from (seed, element index) on the host; the image block (the expensive part:
    207 M normals at B=8) is shared by the legs that use the same image shape."""

    def __init__(self, dev):
        self.dev, self.imgs = dev, {}

    def get(self, cfg, batch, seed):
        from bevfusion_multimodal_3d_object_detection_amd import synth
        key = (batch, cfg["cams"], cfg["h"], cfg["w"], seed)
        if cfg["cams"] and key not in self.imgs:
            self.imgs[key] = synth.frame_inputs(batch, cfg["cams"], cfg["h"], cfg["w"], 0, 4, 0, 125, 7, seed=seed)[0].to(self.dev)
        _, pts, radars = synth.frame_inputs(batch, 0, 0, 0, cfg["points"], 4, cfg["radars"], 125, 7, seed=seed)
        return (self.imgs[key] if cfg["cams"] else None, pts.to(self.dev) if pts is not None else None,
                [r.to(self.dev) for r in radars] if radars else None)

    def drop(self):
        self.imgs.clear()


def pmc_traffic(config: int, dtype: str, mode: str, batch: int, kernel: str):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of THIS leg (FETCH_SIZE x2 + WRITE_SIZE
    in separate passes, MI355X_MICROARCH.md "HBM"; tools/pmc_summary.py); only valid for the configuration the profile was
    taken on.  bench.py cannot read PMC counters itself (they need rocprofv3 around the process), so the figure is labelled
    with its source.  Returns (bytes per launch, source) or None."""
    tag = f"config{config}" + ("" if dtype == "fp32" else f"_{dtype}") + ("_train" if mode == "train" else "") + f"_b{batch}"
    want = {"wino_f32": ("wino_f32<",), "wino_wgrad_f32": ("wino_wgrad_f32",), "conv_wgrad_f32": ("conv_wgrad_f32",),
            "conv_igemm_f32": ("conv_igemm<float", "conv_igemm_hybrid<float"),
            "conv_igemm_bf16": ("conv_igemm<__bf16", "conv_igemm_hybrid<__bf16", "conv_igemm<__hip_bfloat16",
                                "conv_igemm_hybrid<__hip_bfloat16", "conv3x3_bf16"),
            "conv3x3_bf16": ("conv3x3_bf16",),
            "stem": ("stem_",)}.get(kernel)
    if not want:
        return None
    for rnd in PMC_PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{tag}.json")
        if os.path.exists(path):
            d = json.load(open(path))
            hit = [v for k, v in d.items() if k.startswith(want)]
            n = sum(v["launches"] for v in hit)
            if n:
                return (sum(v["launches"] * v["hbm_bytes_per_launch"] for v in hit) / n,
                        "committed profile: " + os.path.relpath(path, ROOT))
    return None


def mfma_roofline(tot, dtype, conv, mode, timer_steps, timer_elapsed, where):
    """The leg's `roofline` object from the HIP-event brackets (engine.KernelTimer.totals()).

    frac = EXECUTED matrix-pipe FLOPs / (kernel time x MFMA peak of the kernel's operand type): at most 1 by construction (asserted).
    The Winograd kernels execute 16/36 of a layer's direct-convolution FLOPs (SURVEY.md 8d tabulates the direct ones); that
    direct-equivalent rate is reported beside it as `algorithmic_equiv_tflops` and never divided by a peak.  The headline
    fields describe the DOMINANT kernel alone (most kernel time); `conv_aggregate` covers every convolution launch."""
    igemm = {"fp32": "conv_igemm_f32", "bf16": "conv_igemm_bf16", "f32x3": "conv_split_f32x3"}[dtype]
    if dtype == "fp32" and conv == "wino_x3":
        igemm = "conv_split_f32x3"
    # f32x3 spends six bf16 MFMA products per algorithmic multiply-add: its ceiling is a sixth of the bf16 peak
    peaks = {"conv_igemm_f32": PEAK_F32_MFMA_TFLOPS, "conv_igemm_bf16": PEAK_BF16_MFMA_TFLOPS, "conv3x3_bf16": PEAK_BF16_MFMA_TFLOPS,
             "conv_split_f32x3": PEAK_BF16_MFMA_TFLOPS / 6.0}
    per = {}
    for span, (kernel, share) in MFMA_SPANS.items():
        t = tot.get(span)
        if not t or t["ms"] <= 0:
            continue
        kernel = kernel or igemm
        if kernel == "stem":
            kernel = "stem_pool7x7_bf16mma" if dtype == "bf16" else "stem_conv7x7_f32"
        d = per.setdefault(kernel, dict(ms=0.0, launches=0, algorithmic=0.0, executed=0.0,
                                        peak=peaks.get(kernel, PEAK_BF16_MFMA_TFLOPS if kernel.endswith("bf16mma") else PEAK_F32_MFMA_TFLOPS)))
        d["ms"] += t["ms"]; d["launches"] += t["launches"]; d["algorithmic"] += t["flops"]; d["executed"] += t["flops"] * share
    convs = {k: v for k, v in per.items() if not k.startswith(("stem", "pointnet_front"))}
    if not convs:
        return None

    def rates(v):
        ex = v["executed"] / (v["ms"] * 1e-3) / 1e12
        frac = ex / v["peak"]
        assert 0.0 < frac <= 1.0, f"roofline fraction {frac:.3f} outside (0, 1]: executed-FLOP accounting is wrong"
        return {"achieved": ex, "peak": v["peak"], "unit": "TFLOP/s", "frac": frac,
                "algorithmic_equiv_tflops": v["algorithmic"] / (v["ms"] * 1e-3) / 1e12,
                "launches_per_step": v["launches"] / timer_steps, "avg_launch_ms": v["ms"] / v["launches"],
                "executed_gflop_per_step": v["executed"] / timer_steps / 1e9,
                "share_of_step": v["ms"] / (1e3 * timer_elapsed)}
    dom = max(convs, key=lambda k: convs[k]["ms"])
    roof = {"kernel": dom, "bound": "mfma"}
    roof.update(rates(convs[dom]))
    roof["traffic"], roof["traffic_source"] = None, None
    roof["measured_over"] = where
    roof["kernels"] = {k: rates(v) for k, v in per.items() if k != dom}
    agg_ms = sum(v["ms"] for v in convs.values())
    agg = {"kernels": sorted(convs), "ms_per_step": agg_ms / timer_steps, "share_of_step": agg_ms / (1e3 * timer_elapsed),
           "executed_tflops": sum(v["executed"] for v in convs.values()) / (agg_ms * 1e-3) / 1e12,
           "algorithmic_equiv_tflops": sum(v["algorithmic"] for v in convs.values()) / (agg_ms * 1e-3) / 1e12}
    if len({v["peak"] for v in convs.values()}) == 1:          # one operand type: the aggregate has a roofline of its own
        agg["frac"] = agg["executed_tflops"] / convs[dom]["peak"]
        assert agg["frac"] <= 1.0
    roof["conv_aggregate"] = agg
    return roof


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


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, state_dict, budget_s=8.0):
    """The oracle (CPU restatement, oracle/ref_model.py) on the host cores, same synthetic frames: B=1 (1 warm-up frame, then
    up to `budget_s` seconds of frames) and one B=8 batch (SURVEY.md 8d asks for both)."""
    import torch
    from bevfusion_multimodal_3d_object_detection_amd import synth
    from oracle import ref_model
    cores = host_cores()
    torch.set_num_threads(cores)
    ora = ref_model.make_detector(cfg["modality"], cfg["bev"], cfg["bev"])
    ora.load_state_dict(state_dict)
    ora.eval()

    def frames(b):
        return synth.frame_inputs(b, cfg["cams"], cfg["h"], cfg["w"], cfg["points"], 4, cfg["radars"], 125, 7, seed=0x5EED + 2000)
    samples = {}
    with torch.no_grad():
        imgs, pts, radars = frames(1)
        ora(imgs, pts, radars or None)                      # warm-up frame
        n, t0 = 0, time.perf_counter()
        while n < 2 or (time.perf_counter() - t0 < budget_s and n < 8):
            ora(imgs, pts, radars or None)
            n += 1
        samples["b1"] = {"frames_per_s": n / (time.perf_counter() - t0), "frames": n, "warmup_frames": 1}
        imgs, pts, radars = frames(8)
        t0 = time.perf_counter()
        ora(imgs, pts, radars or None)                      # one batch of 8 (the threads and allocator are warm from B=1)
        samples["b8"] = {"frames_per_s": 8 / (time.perf_counter() - t0), "frames": 8, "warmup_frames": 0}
    # `value` = the B=1 rate: the one configuration the reference itself runs on a CPU (BASELINE configs[0], ref src/fusion.py:1228-1330)
    # and the figure every earlier round quoted; the B=8 rate (the oracle batches better than the reference's per-frame loop) is
    # beside it and bench.py prints the GPU / CPU ratio against BOTH
    return dict(value=samples["b1"]["frames_per_s"], unit="frames/s", cores=torch.get_num_threads(), kind="port",
                cpu_model=cpu_model(), samples=samples, value_b8=samples["b8"]["frames_per_s"],
                sample=f"oracle/ref_model.py (PyTorch-CPU fp32), same config: {samples['b1']['frames']} frames at B=1 after 1 warm-up "
                       f"frame (= value), then one batch of 8 (= value_b8)")


def run_leg(config, dtype, mode, batch, steps, warmup, ctx, graph=False, kernel_timer=True, keep_state=False, conv="f32"):
    """One timed leg: W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both
    sides, MAX over ranks.  Returns the leg's record (value = whole-job frames/s) and, if asked, the state dict."""
    import torch
    from bevfusion_multimodal_3d_object_detection_amd import engine, replicas, synth
    dev, dist, rank, world = ctx["dev"], ctx["dist"], ctx["rank"], ctx["world"]
    cfg = CONFIGS[config]
    model_cpu = build_model(cfg)
    state = {k: v.clone() for k, v in model_cpu.state_dict().items()} if keep_state else None
    model = model_cpu.to(dev)
    engine.set_conv_mode("f32x3" if dtype == "f32x3" else (conv if dtype == "fp32" else "f32"))
    if dtype == "bf16":
        model = model.bfloat16()
    seed = replicas.frame_seed(0x5EED, config if mode == "train" else 2, rank)     # inference legs share config 2's images
    inputs = ctx["inputs"].get(cfg, batch, seed)

    if mode == "train":
        from bevfusion_multimodal_3d_object_detection_amd import centernet_target as ct
        from bevfusion_multimodal_3d_object_detection_amd import training
        model.train()
        boxes, labels = synth.gt_boxes(batch, 20, seed=seed)
        gt = {"gt_boxes": boxes.to(dev), "gt_labels": labels.to(dev)}
        crit = ct.CenterNetLoss()
        # ref train_detect.py:725-741 (AdamW lr 1e-4 wd 0.01) and :431 (clip_grad_norm_ 10), clip folded into the update
        opt = training.FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.01, max_grad_norm=10.0)
        reducer = replicas.GradReducer(dist) if dist is not None else None
        training.set_grad_reducer(reducer)     # DP: gradient all-reduce (RCCL) issued from inside the backward, overlapped

        def step():
            pred = model(*inputs)
            tgt = ct.prepare_centernet_targets(gt, dev)
            losses = crit(pred, tgt)
            opt.zero_grad()
            losses["total_loss"].backward()
            opt.step()
            return {k: v.detach() for k, v in losses.items()}
    elif graph:
        graphed = model.make_graphed(*inputs)
        kernel_timer = False                            # per-launch event brackets cannot live inside a captured graph

        def step():
            return graphed(*inputs)
    else:
        def step():
            return model(*inputs)

    try:
        for _ in range(max(warmup, 1)):
            out = step()
        torch.cuda.synchronize()
        assert all(torch.isfinite(v).all() for v in out.values()), "non-finite head output"

        # Inference: the per-launch HIP-event brackets (37 launches a step) ride inside the timed region.  Training
        # issues ~700 launches a step and the brackets' host cost would distort `value`, so its roofline is taken
        # over two extra, untimed steps right after the timed region.
        timer = engine.KernelTimer() if kernel_timer else None
        timer_in_region = timer is not None and mode == "infer"
        engine.set_timer(timer if timer_in_region else None)
        replicas.barrier(dist)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        torch.cuda.synchronize()
        replicas.barrier(dist)
        elapsed = time.perf_counter() - t0
        engine.set_timer(None)
        per_rank_elapsed = replicas.gather_over_ranks(elapsed, dist, dev)
        elapsed = replicas.max_over_ranks(elapsed, dist, dev)
        timer_steps, timer_elapsed = steps, elapsed
        if timer is not None and not timer_in_region:
            engine.set_timer(timer)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            timer_steps, timer_elapsed = 2, time.perf_counter() - t1
            engine.set_timer(None)
    finally:
        engine.set_timer(None)
        engine.set_conv_mode("f32")
        if mode == "train":
            training.set_grad_reducer(None)

    rec = {"workload": cfg["name"], "dtype": dtype, "batch_per_gpu": batch, "steps": steps, "warmup": warmup,
           "value": replicas.aggregate_fps(batch * steps, world, elapsed), "unit": "frames/s",
           "ms_per_step": 1e3 * elapsed / steps,
           "mode": "inference forward -> 5 head tensors" if mode == "infer"
           else "training step: fwd (train-mode BN) + targets + loss + bwd + grad all-reduce + clip + AdamW"}
    rec["per_rank_elapsed_s"] = per_rank_elapsed
    if mode == "train" and dist is not None:
        rec["grad_allreduce"] = {"collectives_per_step": reducer.collectives / max(1, warmup + steps + (2 if timer else 0)),
                                 "backend": dist.get_backend(), "ranks": dist.get_world_size()}
    if timer is not None:
        tot = timer.totals()
        roof = mfma_roofline(tot, dtype, conv, mode, timer_steps, timer_elapsed,
                             "the timed region" if timer_in_region else "2 untimed steps after the timed region")
        if roof is not None:
            tr = pmc_traffic(config, dtype, mode, batch, roof["kernel"])
            if tr is not None:
                roof["traffic"], roof["traffic_source"] = tr
            rec["roofline"] = roof
        pool = tot.get("bev_pool")
        if pool:
            gbs = pool["bytes"] / (pool["ms"] * 1e-3) / 1e9
            rec["roofline_bev_pool"] = {"kernel": "cam_mean+bilinear_nhwc", "bound": "hbm", "achieved": gbs,
                                        "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                        "traffic": None, "mb_per_step": pool["bytes"] / timer_steps / 1e6}
    del model, inputs, out
    torch.cuda.empty_cache()
    return rec, state


EXTRA_DEADLINE_STATUS = 3


def install_exit_guard(seconds: float):
    """N > 1, after the headline line has left: if the extra (data-parallel training) leg is still running after `seconds`
    -- i.e. a gradient all-reduce never completed -- report it as an error record on stderr and END THE RANK WITH A NON-ZERO
    STATUS: a hung collective must reach the launcher as a failure, never as success.  The deadline sits
    below the process-group timeout (BEVF_DIST_TIMEOUT_S) so this message, not RCCL's watchdog abort, is what the log shows.
    Returns the timer; cancel() it once the leg has completed."""
    import threading

    def _deadline():
        print("[bench extra] " + json.dumps({"error": f"deadline: the extra leg did not complete within {seconds:.0f} s "
                                                      "(hung collective?); headline already printed, exiting non-zero"}),
              file=sys.stderr, flush=True)
        os._exit(EXTRA_DEADLINE_STATUS)
    guard = threading.Timer(seconds, _deadline)
    guard.daemon = True
    guard.start()
    return guard


def stub_main(args):
    """BEVF_BENCH_STUB=1: the launcher / rank / barrier / MAX-over-ranks control flow with a CPU stand-in for the step
    (CPU-only rehearsal of `--gpus N`, tests/test_multiproc_gloo.py); prints the same JSON skeleton, no measurements."""
    import torch
    from bevfusion_multimodal_3d_object_detection_amd import replicas
    rank, _, world = replicas.rank_world()
    dist = replicas.init(os.environ.get("BEVF_DIST_BACKEND", "gloo"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) are running")
    x = torch.ones(64, 64)
    replicas.barrier(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = (x @ x).clamp_(max=1.0)
    replicas.barrier(dist)
    elapsed = replicas.max_over_ranks(time.perf_counter() - t0, dist, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": replicas.aggregate_fps(args.steps, world, elapsed), "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "stub": True,
                          "dist": {"backend": dist.get_backend() if dist else None,
                                   "ranks": dist.get_world_size() if dist else 1}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    launch_ranks_if_needed(args, argv)              # may not return; nothing above this line touches the GPU
    if os.environ.get("BEVF_BENCH_STUB") == "1":
        return stub_main(args)

    import torch
    from bevfusion_multimodal_3d_object_detection_amd import replicas
    rank, local_rank, world = replicas.rank_world()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) are running")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # one process per GPU.  BEVF_DIST_BACKEND=gloo + fewer GPUs than ranks is a control-flow rehearsal only
    # (ranks then share a device); the driver's runs use the default: RCCL, one GPU per rank.
    backend = os.environ.get("BEVF_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = replicas.init(backend, dev)             # RCCL; inference: only the barrier and the MAX of the elapsed time
    if dist is not None and dist.get_world_size() != args.gpus:
        raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    if args.mode == "train":
        args.config = 4
        if args.dtype != "fp32":
            raise SystemExit("training runs in exact fp32 (the reference has no mixed precision, SURVEY.md 5)")
    cfg = CONFIGS[args.config]
    if args.batch is None:
        args.batch = 2 if args.config == 5 else 8
    ctx = dict(dev=dev, dist=dist, rank=rank, world=world, inputs=InputCache(dev))

    head, state = run_leg(args.config, args.dtype, args.mode, args.batch, args.steps, args.warmup, ctx,
                          graph=args.graph, kernel_timer=not args.no_kernel_timer, keep_state=True, conv=args.conv)

    def emit(extras):
        if rank != 0:
            return
        line = {
            "metric": f"BEV frames/sec (6-cam+LiDAR, {cfg['bev']}x{cfg['bev']} BEV)",
            "value": head["value"], "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": cfg["name"], "batch_per_gpu": args.batch,
                       "parallelism": f"replicas x{world}" if args.mode == "infer" else f"dp{world}",
                       "conv_kernels": args.conv if args.dtype == "fp32" else "f32",
                       "weights": "random-init (synthetic, seeded)", "launch": "hipGraph replay" if args.graph else "eager",
                       "mode": head["mode"]},
            # ranks of the process group the barrier / MAX (and, in training legs, the gradient all-reduce) ran on
            "rccl_ranks": dist.get_world_size() if (dist is not None and dist.get_backend() == "nccl") else (1 if dist is None else 0),
            "dist_backend": dist.get_backend() if dist is not None else None,
        }
        if args.mode == "train":
            line["metric"] = "training frames/sec (camera+LiDAR, per-GPU batch 8)"
        for k in ("roofline", "roofline_bev_pool", "grad_allreduce"):
            if k in head:
                line[k] = head[k]
        if world == 1 and not args.no_cpu_baseline and args.mode == "infer":
            line["cpu_baseline"] = cpu_baseline(cfg, state)
            line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
            line["gpu_over_cpu_b8"] = line["value"] / line["cpu_baseline"]["value_b8"]
        if world > 1:                                        # a straggler must be visible in the line itself
            line["per_rank"] = {"ms_per_step": [1e3 * t / args.steps for t in head["per_rank_elapsed_s"]],
                                "frames_per_s": [args.batch * args.steps / t for t in head["per_rank_elapsed_s"]],
                                "ms_per_step_min": 1e3 * min(head["per_rank_elapsed_s"]) / args.steps,
                                "ms_per_step_max": 1e3 * max(head["per_rank_elapsed_s"]) / args.steps}
        if extras:
            line["extra"] = {"configs": extras}
        print(json.dumps(line), flush=True)

    plan = []
    if args.extras != "none" and not args.graph:
        # config 1 = the shape the reference itself runs (ref src/fusion.py:1228-1330: camera_only, B=1, 6x448x800), then B=8
        plan = [(3, "bf16", "infer", 8), (5, "bf16", "infer", 2), (1, "fp32", "infer", 1), (1, "fp32", "infer", 8),
                (4, "fp32", "train", 8)]
        if args.conv == "wino" and args.dtype == "fp32" and args.mode == "infer" and args.config == 2 and world == 1:
            # the headline's workload once more with the opt-in mixed convolution mode (NOT the headline: its non-Winograd layers
            # multiply through three bf16 planes instead of fp32 FMAs; fp32-level error)
            plan.insert(0, (2, "fp32", "infer", args.batch or 8, "wino_x3"))
        if world > 1 and args.extras == "auto":
            plan = [(4, "fp32", "train", 8)]
        plan = [p for p in plan if len(p) > 4 or (p[0], p[1], p[2]) != (args.config, args.dtype, args.mode)]
    # N > 1: the ONE JSON line (the replica headline the scaling curve is computed from) leaves BEFORE the extra leg, whose
    # gradient all-reduce is the only collective of this program: if RCCL misbehaves there, the headline is already out.  The
    # leg's record then goes to stderr (`[bench extra] {...}`).  N = 1: no collective anywhere, one line at the end with everything.
    guard = None
    if world > 1:
        emit([])
        # The headline is out.  The extra leg's gradient all-reduce is this program's only data-path collective; if it never
        # completes the rank reports that and exits NON-ZERO (install_exit_guard) -- a hang is a failure, not a success.
        if plan:
            guard = install_exit_guard(float(os.environ.get("BEVF_EXTRA_DEADLINE_S", "240")))
    extras = []
    failed = False
    for config, dtype, mode, batch, *conv_override in plan:
        if mode == "train":
            ctx["inputs"].drop()
        try:
            rec, _ = run_leg(config, dtype, mode, batch, args.extra_steps, 3, ctx, conv=conv_override[0] if conv_override else args.conv)
            if conv_override:
                rec["conv_kernels"] = conv_override[0]
        except Exception as e:                               # the headline stays; the failure is recorded AND reflected in the status at N > 1
            rec = {"workload": CONFIGS[config]["name"], "dtype": dtype, "error": f"{type(e).__name__}: {e}"[:300]}
            failed = True
        if world == 1:
            rec.pop("per_rank_elapsed_s", None)
        extras.append(rec)
        if world > 1 and rank == 0:
            print("[bench extra] " + json.dumps(rec), file=sys.stderr, flush=True)
    if guard is not None:
        guard.cancel()
    if world == 1 and plan and args.config == 2 and args.mode == "infer":
        # the non-convolution kernels the north star names (K20 voxeliser / VFE / scatter, input pipeline, BEV pooling, decode), each
        # alone against its algorithmic bytes: tools/frontend_bench.py
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import frontend_bench
            extras.append({"workload": "front-end / non-convolution kernels, each alone (HBM roofline, 8 TB/s)",
                           "kernels": frontend_bench.run(3)})
        except Exception as e:
            extras.append({"workload": "front-end kernels", "error": f"{type(e).__name__}: {e}"[:300]})
    if world == 1:
        emit(extras)
    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:                               # e.g. a peer that left through the deadline above
            print(f"[bench] process-group shutdown: {type(e).__name__}: {e}"[:300], file=sys.stderr, flush=True)
            failed = True
    if world > 1 and failed:
        sys.exit(EXTRA_DEADLINE_STATUS)


if __name__ == "__main__":
    main()
