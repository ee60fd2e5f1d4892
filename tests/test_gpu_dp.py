"""Data-parallel training on the device: two ranks sharing one MI355X (gloo for the exchange, so the test
needs no second GPU), each with its own shard; the in-backward GradReducer must leave both ranks with identical
parameters, equal to a single process stepping on the MEAN of the two shards' gradients."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build():
    from bevfusion_multimodal_3d_object_detection_amd import fusion, synth
    m = fusion.create_detector("camera+lidar", "bev", "centernet", bev_h=50, bev_w=50)
    synth.fill_state_dict_(m, 31)
    return m.cuda().train()


def _shard(rank):
    from bevfusion_multimodal_3d_object_detection_amd import synth
    imgs, pts, _ = synth.frame_inputs(1, 2, 64, 96, 300, 4, seed=900 + rank)
    boxes, labels = synth.gt_boxes(1, 6, seed=950 + rank)
    return imgs.cuda(), pts.cuda(), {"gt_boxes": boxes.cuda(), "gt_labels": labels.cuda()}


def _grads(model, shard):
    from bevfusion_multimodal_3d_object_detection_amd import centernet_target as ct
    imgs, pts, gt = shard
    for p in model.parameters():
        p.grad = None
    losses = ct.CenterNetLoss()(model(imgs, pts, None), ct.prepare_centernet_targets(gt, imgs.device))
    losses["total_loss"].backward()
    return [p.grad.detach().clone() for p in model.parameters()]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from bevfusion_multimodal_3d_object_detection_amd import replicas, training
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    model = _build()
    red = replicas.GradReducer(dist)
    training.set_grad_reducer(red)
    opt = training.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.01, max_grad_norm=10.0)
    g = _grads(model, _shard(rank))
    opt.step()
    torch.cuda.synchronize()
    q.put((rank, red.collectives, [t.cpu().numpy() for t in g[:6]],
           [p.detach().cpu().numpy() for p in list(model.parameters())[:6]], float(opt.last_grad_norm)))     # numpy: pickled by value
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_mean_gradient_step(gpu):
    from bevfusion_multimodal_3d_object_detection_amd import training
    from tests.conftest import rel_err
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, n0, g0, p0, norm0), (_, n1, g1, p1, norm1) = got
    g0, g1, p0, p1 = ([torch.from_numpy(a) for a in v] for v in (g0, g1, p0, p1))
    assert n0 == n1 and n0 >= 3                                   # buckets left while the backward was still running
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)                                  # both ranks hold the same averaged gradients
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)                                  # ... and took the same step
    assert norm0 == norm1
    # single process: mean of the two shards' gradients, then the same optimiser step
    training.set_grad_reducer(None)
    model = _build()
    ga, gb = _grads(model, _shard(0)), _grads(model, _shard(1))
    for p, a, b in zip(model.parameters(), ga, gb):
        p.grad = (a + b) / 2
    for a, ref in zip(g0, [(x + y) / 2 for x, y in zip(ga[:6], gb[:6])]):
        assert rel_err(a, ref.cpu()) <= 1e-5                      # float-atomic accumulation order differs run to run
    opt = training.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.01, max_grad_norm=10.0)
    opt.step()
    # the first AdamW step moves every element by ~lr * sign(g): where |g| is rounding noise (e.g. a convolution bias in
    # front of a train-mode BatchNorm) the sign, hence the element, legitimately differs between two runs whose
    # float-atomic accumulation order differs -- compare the elements whose gradient is well above that noise
    for a, p, ref in zip(p0, list(model.parameters())[:6], [(x + y) / 2 for x, y in zip(ga[:6], gb[:6])]):
        solid = (ref.abs() > 1e-3 * ref.abs().max()).cpu()
        assert float(solid.float().mean()) > 0.5
        assert rel_err(a[solid], p.detach().cpu()[solid]) <= 1e-5
        assert float((a - p.detach().cpu()).abs().max()) <= 2.1e-3               # nowhere more than two steps of lr 1e-3 apart
