"""Training step on the HIP path (ref src/train_detect.py:401-434).

`model.train()` routes `FlexibleMultiModal3DDetector.forward` through `_DetectorTrainFn`, one
`torch.autograd.Function` for the whole detector: the forward runs the same NHWC kernels as inference but
with train-mode BatchNorm (batch statistics, running-stat update) and keeps a tape; the backward walks the
tape with hand-written gradient kernels (MFMA weight / data gradients, BN, pooling, resample, dense layers,
head) and returns the parameter gradients to autograd, so the reference's training loop --
`loss.backward(); clip_grad_norm_(...); optimizer.step()` -- works unchanged.  torch is used for tensor
allocation and for weight layout permutes; no torch compute op touches an activation.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E


def _lib():
    return L.lib()


def _st():
    return torch.cuda.current_stream().cuda_stream


def _ck(rc, what):
    if rc != 0:
        raise L.BevfError(f"{what} failed ({rc}): {_lib().bevf_last_error().decode()}")


def _new(n: int, dev, dtype=torch.float32) -> torch.Tensor:
    return torch.empty(max(int(n), 4), dtype=dtype, device=dev)


class _ZeroPool:
    """One zero-filled buffer per backward pass from which the many small accumulation targets (conv weight
    gradients, bias sums) are carved: one fill launch instead of ~60."""

    def __init__(self, dev, floats: int = 16 << 20):
        self.buf = torch.zeros(floats, device=dev)
        self.off = 0

    def take(self, n: int):
        n4 = (n + 3) // 4 * 4                                 # keep every slice 16-byte aligned
        if n > (2 << 20) or self.off + n4 > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n4]
        self.off += n4
        return t


_ZPOOL: Optional[_ZeroPool] = None


def _zeros(n: int, dev, dtype=torch.float32) -> torch.Tensor:
    n = max(int(n), 4)
    if _ZPOOL is not None and dtype == torch.float32 and _ZPOOL.buf.device == torch.device(dev):
        t = _ZPOOL.take(n)
        if t is not None:
            return t
    return torch.zeros(n, dtype=dtype, device=dev)


_PIXTAB: Dict[tuple, torch.Tensor] = {}


def _pixtab(N, H, W, k, stride, pad, x_cs, dev) -> torch.Tensor:
    key = (N, H, W, k, stride, pad, x_cs, str(dev))
    t = _PIXTAB.get(key)
    if t is None:
        t = torch.empty(_lib().bevf_conv_pixtab_bytes(N, H, W, k, k, stride, pad) // 4, dtype=torch.int32, device=dev)
        _ck(_lib().bevf_conv_pixtab(t.data_ptr(), N, H, W, k, k, stride, pad, x_cs, _st()), "bevf_conv_pixtab")
        _PIXTAB[key] = t
    return t


def _wino_wgrad_table(N, H, W, x_cs, dy_cs, dev) -> torch.Tensor:
    """Per-shape tile table of the Winograd weight gradient (pixel byte offsets per 2x2 tile), cached like _pixtab."""
    key = ("wwtab", N, H, W, x_cs, dy_cs, str(dev))
    t = _PIXTAB.get(key)
    if t is None:
        t = torch.empty(_lib().bevf_wino_wgrad_table_bytes(N, H, W) // 4, dtype=torch.int32, device=dev)
        _ck(_lib().bevf_wino_wgrad_table(t.data_ptr(), N, H, W, x_cs, dy_cs, _st()), "bevf_wino_wgrad_table")
        _PIXTAB[key] = t
    return t


# ---- primitive ops (thin wrappers over the C-ABI; all tensors fp32 cuda, flat NHWC) ---------------------------------

# The conv kernels address their operands with 32-bit byte offsets: a tensor handed to one launch must stay below
# BUF_LIMIT bytes.  Batches past that run as image chunks (outputs are slices of one buffer, weight gradients
# accumulate); tests shrink BUF_LIMIT to exercise the chunking on small shapes.
BUF_LIMIT = (1 << 31) - 1


def _image_chunk(N: int, *per_image_elems: int) -> int:
    n_max = max(1, BUF_LIMIT // (4 * max(per_image_elems)))
    return N if N <= n_max else -(-N // -(-N // n_max))


def _wino_ok(cin, k, stride, pad) -> bool:
    """The fused fp32 Winograd kernel (csrc/conv_wino.hip) serves the 3x3 / stride 1 / pad 1 layers when the conv mode
    says so (engine.set_conv_mode("wino")): forward and data-gradient convolutions of the training step alike."""
    return E.conv_mode() in ("wino", "wino_x3") and (k, stride, pad) == (3, 1, 1) and cin % 32 == 0      # (the split kernels are inference-only)


def _conv_launch(x, w_ohwi, bias, y, n, H, W, cin, cout, k, stride, pad, relu, res=None, stats=None):
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if _wino_ok(cin, k, stride, pad):
        with E._span("conv_wino_f32", flops=2.0 * n * Ho * Wo * cout * k * k * cin):
            L.conv3x3_wino(x, L.wino_filter_transform(w_ohwi, cout, cin), None, bias, y, N=n, H=H, W=W, Cin=cin, x_cs=cin,
                           Cout=cout, y_cs=cout, relu=relu, res=res, res_cs=cout if res is not None else 0,
                           stats=stats[0] if stats else None, stats_pivot=stats[1] if stats else None)
        return
    with E._span("conv_igemm_f32", flops=2.0 * n * Ho * Wo * cout * k * k * cin):
        L.conv2d_nhwc(x, w_ohwi, None, bias, y, N=n, H=H, W=W, Cin=cin, x_cs=cin, Cout=cout, y_cs=cout, KH=k, KW=k,
                      stride=stride, pad=pad, relu=relu, res=res, res_cs=cout if res is not None else 0)


def conv_raw(x, w_ohwi, bias, N, H, W, cin, cout, k, stride, pad, relu=False, bn_pivot=None):
    """y = conv(x) (+bias) (+ReLU).  bn_pivot [cout] (training forward in front of a BatchNorm, Winograd layers only): the
    conv epilogue also leaves the BatchNorm partial sums; returns (y, Ho, Wo, partials or None) then, partials =
    (part [G][cout][2], G, pivot) for bn_train_forward."""
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    per = _image_chunk(N, H * W * cin, Ho * Wo * cout)
    y = _new(N * Ho * Wo * cout, x.device)
    fuse = bn_pivot is not None and not relu and _wino_ok(cin, k, stride, pad)
    part, rows = None, 0
    if fuse:
        rows_of = lambda n: _lib().bevf_wino_stat_rows(n, H, W)
        G = sum(rows_of(min(per, N - i0)) for i0 in range(0, N, per))
        part = _new(G * cout * 2, x.device)
    for i0 in range(0, N, per):
        n = min(per, N - i0)
        st = None
        if fuse:
            st = (part[rows * cout * 2:], bn_pivot)
            rows += rows_of(n)
        _conv_launch(x[i0 * H * W * cin:], w_ohwi, bias, y[i0 * Ho * Wo * cout:], n, H, W, cin, cout, k, stride, pad, relu, stats=st)
    if bn_pivot is not None:
        return y, Ho, Wo, ((part, rows, bn_pivot) if fuse else None)
    return y, Ho, Wo


# 3x3 / stride 1 / pad 1 weight gradients with channels in multiples of 64 run in the Winograd domain (csrc/conv_wino_wgrad.hip:
# 2.25x fewer MFMA FLOPs, deterministic); False = the pixel-GEMM with atomics everywhere (the round-1 path)
WINO_WGRAD = True


def conv_wgrad(x, dy, N, H, W, cin, cout, k, stride, pad, dw=None) -> torch.Tensor:
    """Returns dW in OHWI layout [cout][k][k][cin] (accumulates into `dw` when given)."""
    fresh = dw is None
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    per = _image_chunk(N, H * W * cin, Ho * Wo * cout, Ho * Wo * k * k)          # (the tap table has the same limit)
    if per < N:
        if fresh:
            dw = _zeros(cout * k * k * cin, x.device)
        for i0 in range(0, N, per):
            n = min(per, N - i0)
            conv_wgrad(x[i0 * H * W * cin:], dy[i0 * Ho * Wo * cout:], n, H, W, cin, cout, k, stride, pad, dw=dw)
        return dw[:cout * k * k * cin].view(cout, k, k, cin)
    flops = 2.0 * N * Ho * Wo * cout * k * k * cin
    ws = (_lib().bevf_wino_wgrad_workspace_floats(N, H, W, cin, cout)
          if WINO_WGRAD and E.conv_mode() in ("wino", "wino_x3") and (k, stride, pad) == (3, 1, 1) else 0)
    if ws:
        if fresh:
            dw = _new(cout * 9 * cin, x.device)
        d = L.WgradDesc(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _wino_wgrad_table(N, H, W, cin, cout, x.device).data_ptr(),
                        N, H, W, cin, cin, cout, cout, 3, 3, 1, 1)
        work = _new(ws, x.device)
        with E._span("conv_wgrad_wino_f32", flops=flops):
            _ck(_lib().bevf_conv3x3_wgrad_wino_f32(C.byref(d), work.data_ptr(), 0 if fresh else 1, _st()),
                "bevf_conv3x3_wgrad_wino_f32")
        return dw[:cout * 9 * cin].view(cout, 3, 3, cin)
    if fresh:
        dw = _zeros(cout * k * k * cin, x.device)
    d = L.WgradDesc(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _pixtab(N, H, W, k, stride, pad, cin, x.device).data_ptr(),
                    N, H, W, cin, cin, cout, cout, k, k, stride, pad)
    with E._span("conv_wgrad_f32", flops=flops):
        _ck(_lib().bevf_conv2d_wgrad_f32(C.byref(d), _st()), "bevf_conv2d_wgrad_f32")
    return dw[:cout * k * k * cin].view(cout, k, k, cin)


def _dgrad_conv(src, filt_oihw_sub, N, Hs, Ws, cin, cout, kh, kw):
    """One stride-1, pad-0 correlation of `src` [N][Hs][Ws][cout] with taps filt[t][u] -> [N][Hs-kh+1][Ws-kw+1][cin]."""
    wt = filt_oihw_sub.permute(1, 2, 3, 0).contiguous().view(-1)                         # [cin][kh][kw][cout]
    ho, wo = Hs - kh + 1, Ws - kw + 1
    out = _new(N * ho * wo * cin, src.device)
    L.conv2d_nhwc(src, wt, None, None, out, N=N, H=Hs, W=Ws, Cin=cout, x_cs=cout, Cout=cin, y_cs=cin, KH=kh, KW=kw,
                  stride=1, pad=0, relu=False)
    return out, ho, wo


def dgrad_can_fuse_bn(N, H, W, cin, cout, k, stride, pad) -> bool:
    """True when conv_dgrad runs as ONE fused-Winograd launch (so its epilogue can do the next BatchNorm's first backward pass)."""
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    return (stride == 1 and _wino_ok(cout, k, 1, k - 1 - pad) and (Ho, Wo) == (H, W)
            and _image_chunk(N, H * W * max(cin, cout), (Ho + 1) * (Wo + 1) * cout) >= N)


def conv_dgrad(dy, weight_oihw, N, H, W, cin, cout, k, stride, pad, add=None, bnb=None):
    """dX [N*H*W*cin] = conv_transpose(dy, W).  Stride 1: the forward kernel on dy with the flipped filter.  Stride 2
    (3x3 pad 1, or 1x1 pad 0 -- the ResNet shapes): the four input-parity classes (ih&1, iw&1) each see a fixed subset
    of the taps, so each is a small stride-1 conv over dy (1x1 / 1x2 / 2x1 / 2x2 taps) and `interleave2x2` assembles dX:
    exactly the forward's MFMA work instead of 4x on a zero-stuffed grid.  Other strides: zero stuffing.
    `add` [N*H*W*cin]: a gradient to sum into dX (the skip connection's), fused into the conv epilogue when stride 1.
    `bnb` (only with dgrad_can_fuse_bn): dX is the gradient reaching a train-mode BatchNorm(+ReLU) layer described by bnb = dict(x,
    y|None, mean, invstd, gamma, beta); the epilogue applies that layer's ReLU mask and leaves its backward partial sums:
    returns (dX_masked, (part, G)) then."""
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    per = _image_chunk(N, H * W * max(cin, cout), (Ho + 1) * (Wo + 1) * cout)
    if per < N or (add is not None and stride != 1):
        dx = torch.cat([conv_dgrad(dy[i0 * Ho * Wo * cout:(i0 + min(per, N - i0)) * Ho * Wo * cout], weight_oihw,
                                   min(per, N - i0), H, W, cin, cout, k, stride, pad)[:min(per, N - i0) * H * W * cin]
                        for i0 in range(0, N, per)]) if per < N else conv_dgrad(dy, weight_oihw, N, H, W, cin, cout, k, stride, pad)
        if add is not None:
            add_(dx, add, min(dx.numel(), add.numel()))
        return dx
    flops = 2.0 * N * Ho * Wo * cout * k * k * cin                                       # algorithmic
    w = weight_oihw.detach()
    if stride == 2 and (k, pad) in ((3, 1), (1, 0)):
        dx = _new(N * H * W * cin, dy.device)
        with E._span("conv_dgrad_f32", flops=flops):
            if k == 1:
                c00, h0, w0 = _dgrad_conv(dy, w, N, Ho, Wo, cin, cout, 1, 1)
                cls, hq, wq = [c00, None, None, None], [h0, 0, 0, 0], [w0, 0, 0, 0]
            else:
                src = _new(N * (Ho + 1) * (Wo + 1) * cout, dy.device)                      # dy with a zero row / column appended
                _ck(_lib().bevf_zero_stuff_nhwc_f32(dy.data_ptr(), src.data_ptr(), N, Ho, Wo, cout, Ho + 1, Wo + 1, 1, _st()),
                    "bevf_zero_stuff_nhwc_f32")
                taps = ([1], [2, 0])                                                       # parity 0: kh=1 reads dy[a]; parity 1: kh=2 reads dy[a], kh=0 reads dy[a+1]
                cls, hq, wq = [], [], []
                for ph in (0, 1):
                    for pw in (0, 1):
                        sub = w[:, :, taps[ph]][:, :, :, taps[pw]]
                        c, hc, wc = _dgrad_conv(src, sub, N, Ho + 1, Wo + 1, cin, cout, len(taps[ph]), len(taps[pw]))
                        cls.append(c); hq.append(hc); wq.append(wc)
            ptrs = (C.c_void_p * 4)(*[None if c is None else c.data_ptr() for c in cls])
            _ck(_lib().bevf_interleave2x2_nhwc_f32(ptrs, (C.c_int32 * 4)(*hq), (C.c_int32 * 4)(*wq), dx.data_ptr(), N, H, W, cin,
                                                   _st()), "bevf_interleave2x2_nhwc_f32")
        return dx
    wt = w.flip(2, 3).permute(1, 2, 3, 0).contiguous().view(-1)                          # [cin][k][k][cout]
    src, sh, sw = dy, Ho, Wo
    if stride != 1:
        src = _new(N * H * W * cout, dy.device)
        _ck(_lib().bevf_zero_stuff_nhwc_f32(dy.data_ptr(), src.data_ptr(), N, Ho, Wo, cout, H, W, stride, _st()),
            "bevf_zero_stuff_nhwc_f32")
        sh, sw = H, W
    else:
        assert (Ho, Wo) == (H, W), "stride-1 convs on this path keep the spatial size"
    dx = _new(N * H * W * cin, dy.device)
    wino = stride == 1 and _wino_ok(cout, k, 1, k - 1 - pad)
    with E._span("conv_dgrad_wino_f32" if wino else "conv_dgrad_f32", flops=flops):     # (own span name: bench.py prices the 16/36)
        if wino:
            part = None
            if bnb is not None:
                G = _lib().bevf_wino_stat_rows(N, sh, sw)
                part = _new(G * cin * 2, dy.device)
            L.conv3x3_wino(src, L.wino_filter_transform(wt, cin, cout), None, None, dx, N=N, H=sh, W=sw, Cin=cout, x_cs=cout,
                           Cout=cin, y_cs=cin, relu=False, res=add, res_cs=cin if add is not None else 0, stats=part, bnb=bnb)
            if bnb is not None:
                return dx, (part, G)
        else:
            L.conv2d_nhwc(src, wt, None, None, dx, N=N, H=sh, W=sw, Cin=cout, x_cs=cout, Cout=cin, y_cs=cin, KH=k, KW=k,
                          stride=1, pad=k - 1 - pad, relu=False, res=add if stride == 1 else None,
                          res_cs=cin if (add is not None and stride == 1) else 0)
    if add is not None and stride != 1:
        add_(dx, add, min(dx.numel(), add.numel()))
    return dx


class _BNState:
    __slots__ = ("mean", "invstd", "xraw", "y", "M", "C", "has_res", "frozen")


# Test instrumentation (tests/test_gpu_training.py): when a list, every train-mode forward site that applies a ReLU appends its
# post-activation matrix (tensor [M*C], M, C) -- the decisions the hand-written backward will take -- so that the fp64 oracle can be
# made to take the SAME decisions and whole-network gradients compared to 1e-4 instead of "up to a few ReLU flips".  None = off.
RELU_TRACE: Optional[list] = None


def _trace_relu(y, M: int, Cc: int) -> None:
    if RELU_TRACE is not None and y is not None:
        RELU_TRACE.append((y, M, Cc))


FUSE_BN_STATS = False        # BatchNorm batch statistics from partial sums the Winograd conv epilogue leaves (no stats pass over the
                             # activation).  Built and tested, but off: time-neutral on MI355X (the epilogue is exposed time), and the
                             # sums are shifted by the RUNNING mean, so their accuracy depends on how far that is from the batch mean
                             # (two otherwise identical steps differed by 2e-4 in a gradient when only the running mean differed)


def bn_pivot_of(bn) -> Optional[torch.Tensor]:
    """Shift for the BatchNorm partial sums a conv epilogue produces (FUSE_BN_STATS): any value near the channel mean avoids
    cancellation in E[(x-p)^2] - E[x-p]^2; the running mean is at hand (None: the sums stay a separate, self-shifted pass)."""
    if not FUSE_BN_STATS:
        return None
    rm = getattr(bn, "running_mean", None)
    return rm.detach() if (rm is not None and rm.dtype == torch.float32 and rm.is_cuda) else None


def bn_is_frozen(bn) -> bool:
    """An eval-mode BatchNorm with running statistics inside a module that is being trained normalises with its buffers (torch's rule)."""
    return (not bn.training) and bn.track_running_stats and bn.running_mean is not None


def bn_train_forward(xraw, bn: nn.BatchNorm2d, M: int, Cc: int, res=None, relu=True, apply=True, partials=None):
    """Batch statistics (+ running-buffer update) and, unless apply=False, the normalised activation.
    partials = (part, G, pivot) from a conv epilogue: the statistics are merged from them, xraw is not re-read."""
    dev = xraw.device
    if bn_is_frozen(bn):
        # eval-mode BatchNorm inside a module that trains (mixed mode): the running buffers are the statistics, nothing is updated, and
        # the backward treats them as constants (bevf_bn_backward_f32, relu | 4).  Channel-sized torch arithmetic only.
        mean = bn.running_mean.detach().float().contiguous()
        invstd = torch.rsqrt(bn.running_var.detach().float() + bn.eps).contiguous()
        y = None
        if apply:
            y = _new(M * Cc, dev)
            g = bn.weight.data_ptr() if bn.weight is not None else None
            b = bn.bias.data_ptr() if bn.bias is not None else None
            _ck(_lib().bevf_bn_apply_f32(xraw.data_ptr(), mean.data_ptr(), invstd.data_ptr(), g, b,
                                         res.data_ptr() if res is not None else None, y.data_ptr(), M, Cc, Cc, int(relu), _st()),
                "bevf_bn_apply_f32")
            if relu:
                _trace_relu(y, M, Cc)
        s = _BNState()
        s.mean, s.invstd, s.xraw, s.y, s.M, s.C, s.has_res, s.frozen = mean, invstd, xraw, y, M, Cc, res is not None, True
        return y, s
    mean, var, invstd = _new(Cc, dev), _new(Cc, dev), _new(Cc, dev)
    if partials is not None:
        part, G, pivot = partials
        pivot = pivot.clone()                    # the running mean is updated below, in place
        _ck(_lib().bevf_bn_stats_from_partials_f32(part.data_ptr(), G, pivot.data_ptr(), mean.data_ptr(), var.data_ptr(),
                                                   invstd.data_ptr(), M, Cc, float(bn.eps), _st()), "bevf_bn_stats_from_partials_f32")
    else:
        work = _new(_lib().bevf_bn_work_floats(Cc), dev)
        _ck(_lib().bevf_bn_stats_f32(xraw.data_ptr(), work.data_ptr(), mean.data_ptr(), var.data_ptr(), invstd.data_ptr(),
                                     M, Cc, Cc, float(bn.eps), _st()), "bevf_bn_stats_f32")
    y = None
    if apply:
        y = _new(M * Cc, dev)
        g = bn.weight.data_ptr() if bn.weight is not None else None
        b = bn.bias.data_ptr() if bn.bias is not None else None
        _ck(_lib().bevf_bn_apply_f32(xraw.data_ptr(), mean.data_ptr(), invstd.data_ptr(), g, b,
                                     res.data_ptr() if res is not None else None, y.data_ptr(), M, Cc, Cc, int(relu), _st()),
            "bevf_bn_apply_f32")
    if bn.track_running_stats and bn.running_mean is not None:           # torch: momentum 0.1, unbiased running var
        nbt = bn.num_batches_tracked
        if bn.momentum is None:
            # torch's cumulative moving average: factor 1 / (num_batches_tracked after this batch); the count lives on
            # the device, so this rare setting costs one host read per layer and step
            mom = 1.0 / (int(nbt.item()) + 1) if nbt is not None else 0.0
        else:
            mom = bn.momentum
        _ck(_lib().bevf_bn_update_running_f32(mean.data_ptr(), var.data_ptr(), bn.running_mean.data_ptr(),
                                              bn.running_var.data_ptr(), nbt.data_ptr() if nbt is not None else None, Cc, M,
                                              float(mom), _st()), "bevf_bn_update_running_f32")
        for t in (bn.running_mean, bn.running_var, nbt):                 # written through raw pointers: bump the versions
            if t is not None:                                             # (the engines' repack signature looks at them)
                torch.autograd.graph.increment_version(t)
    if relu:
        _trace_relu(y, M, Cc)
    s = _BNState()
    s.mean, s.invstd, s.xraw, s.y, s.M, s.C, s.has_res, s.frozen = mean, invstd, xraw, y, M, Cc, res is not None, False
    return y, s


def bn_train_backward(dy, s: _BNState, bn, relu=True, need_dx=True):
    """Returns (dxraw or None, dgamma, dbeta).  Layers with a skip connection: dy <- dy*(y>0) in place (it is the skip input's
    gradient); layers without one: dy is left untouched (the mask is recomputed from the raw input in both passes)."""
    dev = dy.device
    work = _new(_lib().bevf_bn_work_floats(s.C), dev)
    dgamma, dbeta = _new(s.C, dev), _new(s.C, dev)
    dx = _new(s.M * s.C, dev) if need_dx else None
    g = bn.weight.data_ptr() if bn.weight is not None else None
    b = bn.bias.data_ptr() if bn.bias is not None else None
    # without a residual the ReLU mask is recomputed from the raw input (same fma as the forward): y is not re-read
    ymask = s.y.data_ptr() if (relu and s.has_res) else None
    _ck(_lib().bevf_bn_backward_f32(dy.data_ptr(), ymask, s.xraw.data_ptr(), s.mean.data_ptr(),
                                    s.invstd.data_ptr(), g, b, work.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                    dx.data_ptr() if dx is not None else None, s.M, s.C, s.C,
                                    ((1 if s.has_res else 2) if relu else 0) | (4 if s.frozen else 0), _st()),
        "bevf_bn_backward_f32")
    return dx, dgamma[:s.C], dbeta[:s.C]


def group_max_with_index(a, G: int, P: int, Cc: int):
    """max over the P rows of each of G groups of a [G*P][Cc] activation, first maximum wins -> (gmax [G*Cc], idx int32 [G*Cc])."""
    dev = a.device
    g = _new(G * Cc, dev)
    idx = torch.empty(G * Cc, dtype=torch.int32, device=dev)
    work = torch.empty(_lib().bevf_group_max_idx_work_bytes(G, P, Cc), dtype=torch.uint8, device=dev)
    _ck(_lib().bevf_group_max_idx_f32(a.data_ptr(), g.data_ptr(), idx.data_ptr(), work.data_ptr(), G, P, Cc, _st()),
        "bevf_group_max_idx_f32")
    return g, idx


def group_max_scatter(dg, idx, G: int, P: int, Cc: int):
    """Backward of group_max_with_index: a zero [G*P][Cc] gradient with dg at the winning rows."""
    d = _zeros(G * P * Cc, dg.device)
    _ck(_lib().bevf_group_max_bwd_f32(dg.data_ptr(), idx.data_ptr(), d.data_ptr(), G, P, Cc, _st()), "bevf_group_max_bwd_f32")
    return d


def _bn_or_none(m):
    """The reference builds `nn.BatchNorm1d(w) if use_bn else nn.Identity()` (ref src/encoders.py:258-269, 520-529)."""
    return m if isinstance(m, nn.modules.batchnorm._BatchNorm) else None


class PointFirstLayer:
    """conv1 of a shared point MLP (Conv1d k=1 over <= 16 input channels) -> train-mode BatchNorm (or none: use_bn=False) -> ReLU."""

    def __init__(self, conv, bn):
        self.conv, self.bn = conv, _bn_or_none(bn)

    def forward(self, rows, M: int, Cc: int):
        self.rows, self.M, self.Cc = rows, M, Cc
        w0 = self.conv.weight.detach().reshape(self.conv.weight.shape[0], Cc).contiguous()
        self.c0 = c0 = w0.shape[0]
        bias = self.conv.bias.detach() if self.conv.bias is not None else None
        out = _new(M * c0, rows.device)
        if self.bn is None:                                                # conv + bias + ReLU in one pass
            L.pointwise_smallk(rows, w0, None, bias, out, M, Cc, c0, True)
            _trace_relu(out, M, c0)
            self.y = out
            return out
        L.pointwise_smallk(rows, w0, None, bias, out, M, Cc, c0, False)
        a, self.bns = bn_train_forward(out, self.bn, M, c0, relu=True)
        return a

    def backward(self, d, sink) -> None:
        M, Cc, c0 = self.M, self.Cc, self.c0
        if self.bn is None:
            n4 = (M * c0 + 3) // 4 * 4
            _ck(_lib().bevf_relu_mask_f32(d.data_ptr(), self.y.data_ptr(), n4, _st()), "bevf_relu_mask_f32")
            draw = d
        else:
            draw, dgamma, dbeta = bn_train_backward(d, self.bns, self.bn, relu=True)
            sink.add(self.bn.weight, dgamma)
            sink.add(self.bn.bias, dbeta)
        if self.conv.bias is not None:
            sink.add(self.conv.bias, colsum(draw, M, c0))
        dw = _zeros(c0 * Cc, d.device)
        _ck(_lib().bevf_smallk_wgrad_f32(draw.data_ptr(), self.rows.data_ptr(), dw.data_ptr(), M, Cc, c0, _st()),
            "bevf_smallk_wgrad_f32")
        sink.add(self.conv.weight, dw[:c0 * Cc])


LOWRANK_GMAX_BACKWARD = True  # PointNet's last layer (conv -> BatchNorm -> ReLU -> max over points): weight / data gradients through
                              # a K x K Gram matrix instead of the dense M x C gradient (half the GEMM FLOPs, no 1.15 GB tensor)
FUSE_POOL_BN_BACKWARD = True  # stem: BatchNorm + ReLU evaluated inside the max-pool (forward) and the max-pool backward gathered inside
                              # the BatchNorm backward passes: neither the normalised map nor its gradient (1.1 GB each) is ever written
FUSE_BN_BACKWARD = False     # the producing data-gradient conv does the next BatchNorm's first backward pass in its epilogue: correct
                             # (tests run both settings) but +0.4 ms per step on MI355X, because the Winograd epilogue is exposed time


def bn_backward_from_partials(dy, s: _BNState, bn, pre):
    """BatchNorm backward when the producer of dy already masked it and left the sums as partials: merge + apply."""
    part, G = pre
    dev = dy.device
    dgamma, dbeta = _new(s.C, dev), _new(s.C, dev)
    dx = _new(s.M * s.C, dev)
    g = bn.weight.data_ptr() if bn.weight is not None else None
    _ck(_lib().bevf_bn_backward_from_partials_f32(dy.data_ptr(), s.xraw.data_ptr(), s.mean.data_ptr(), s.invstd.data_ptr(), g,
                                                  part.data_ptr(), G, dgamma.data_ptr(), dbeta.data_ptr(), dx.data_ptr(),
                                                  s.M, s.C, s.C, _st()), "bevf_bn_backward_from_partials_f32")
    return dx, dgamma[:s.C], dbeta[:s.C]


def colsum(dy, M, Cc):
    """sum over rows (bias gradients)."""
    work = _new(_lib().bevf_bn_work_floats(Cc), dy.device)
    out = _new(Cc, dy.device)
    _ck(_lib().bevf_bn_backward_f32(dy.data_ptr(), None, None, None, None, None, None, work.data_ptr(), None, out.data_ptr(),
                                    None, M, Cc, Cc, 0, _st()), "bevf_bn_backward_f32(colsum)")
    return out[:Cc]


def add_(y, x, n):
    n4 = (n + 3) // 4 * 4
    _ck(_lib().bevf_add_inplace_f32(y.data_ptr(), x.data_ptr(), n4, _st()), "bevf_add_inplace_f32")


class GradSink:
    """Collects parameter gradients by parameter identity (summing repeated contributions)."""

    def __init__(self, reducer=None):
        self.g: Dict[int, torch.Tensor] = {}
        self.reducer = reducer               # replicas.GradReducer: averages gradients over the ranks while backward runs
        self._sent = set()

    def add(self, p: Optional[torch.Tensor], g: torch.Tensor):
        if p is None or not p.requires_grad:
            return
        g = g.reshape(p.shape)
        k = id(p)
        assert k not in self._sent, "gradient contribution after the parameter was handed to the all-reduce"
        self.g[k] = g if k not in self.g else self.g[k] + g

    def ready(self):
        """Every gradient collected so far is final: start its all-reduce now, under the rest of the backward."""
        if self.reducer is None:
            return
        keys = [k for k in self.g if k not in self._sent]
        if keys:
            self.reducer.submit(self, keys)
            self._sent.update(keys)

    def finish(self):
        if self.reducer is not None:
            self.ready()
            self.reducer.finish(self)

    def get(self, p):
        return self.g.get(id(p))


# ---- layer records --------------------------------------------------------------------------------------------------------

class ConvBNLayer:
    """conv (any bias) -> train-mode BN -> (+residual) -> (ReLU).  bn None: conv(+bias)(+ReLU) only (head)."""

    def __init__(self, conv, bn, relu=True):
        self.conv, self.bn, self.relu = conv, bn, relu
        w = conv.weight
        self.k = w.shape[2] if w.dim() == 4 else 1
        self.cin, self.cout = w.shape[1], w.shape[0]
        self.stride = conv.stride[0]
        self.pad = conv.padding[0]

    def forward(self, x, N, H, W, res=None):
        w4 = self.conv.weight.detach()
        if w4.dim() == 3:
            w4 = w4.unsqueeze(-1)
        w_ohwi = w4.permute(0, 2, 3, 1).contiguous().view(-1)
        bias = self.conv.bias.detach() if self.conv.bias is not None else None
        self.x, self.N, self.H, self.W = x, N, H, W
        if self.bn is None:
            y, Ho, Wo = conv_raw(x, w_ohwi, bias, N, H, W, self.cin, self.cout, self.k, self.stride, self.pad, relu=self.relu)
            self.y, self.M = y, N * Ho * Wo
            if self.relu:
                _trace_relu(y, self.M, self.cout)
            return y, Ho, Wo
        pivot, partials = (None if bn_is_frozen(self.bn) else bn_pivot_of(self.bn)), None
        if pivot is not None:
            xraw, Ho, Wo, partials = conv_raw(x, w_ohwi, bias, N, H, W, self.cin, self.cout, self.k, self.stride, self.pad, bn_pivot=pivot)
        else:
            xraw, Ho, Wo = conv_raw(x, w_ohwi, bias, N, H, W, self.cin, self.cout, self.k, self.stride, self.pad)
        self.M = N * Ho * Wo
        y, self.bns = bn_train_forward(xraw, self.bn, self.M, self.cout, res=res, relu=self.relu, partials=partials)
        self.has_res = res is not None
        return y, Ho, Wo

    def forward_groupmax(self, x, B: int, P: int):
        """conv (1x1 over B*P rows) -> train-mode BN -> ReLU -> max over the P rows of each group, without writing the
        activation: the max / argmax kernel evaluates relu(bn(.)) from the raw rows with bn_apply's own fma.  Returns
        (gmax [B*cout], idx int32 [B*cout]); pair with backward_from_groupmax."""
        assert self.bn is not None and self.relu and self.k == 1
        w4 = self.conv.weight.detach()
        if w4.dim() == 3:
            w4 = w4.unsqueeze(-1)
        w_ohwi = w4.permute(0, 2, 3, 1).contiguous().view(-1)
        bias = self.conv.bias.detach() if self.conv.bias is not None else None
        M = B * P
        self.x, self.N, self.H, self.W, self.M, self.has_res = x, M, 1, 1, M, False
        xraw, _, _ = conv_raw(x, w_ohwi, bias, M, 1, 1, self.cin, self.cout, 1, 1, 0)
        _, self.bns = bn_train_forward(xraw, self.bn, M, self.cout, relu=True, apply=False)
        dev = x.device
        if RELU_TRACE is not None:                               # (tests only: the activation this path never writes)
            yt = _new(M * self.cout, dev)
            _ck(_lib().bevf_bn_apply_f32(xraw.data_ptr(), self.bns.mean.data_ptr(), self.bns.invstd.data_ptr(),
                                         self.bn.weight.data_ptr() if self.bn.weight is not None else None,
                                         self.bn.bias.data_ptr() if self.bn.bias is not None else None, None, yt.data_ptr(), M,
                                         self.cout, self.cout, 1, _st()), "bevf_bn_apply_f32")
            _trace_relu(yt, M, self.cout)
        g = _new(B * self.cout, dev)
        idx = torch.empty(B * self.cout, dtype=torch.int32, device=dev)
        work = torch.empty(_lib().bevf_group_max_idx_work_bytes(B, P, self.cout), dtype=torch.uint8, device=dev)
        gam = self.bn.weight.data_ptr() if self.bn.weight is not None else None
        bet = self.bn.bias.data_ptr() if self.bn.bias is not None else None
        _ck(_lib().bevf_bn_relu_group_max_idx_f32(xraw.data_ptr(), self.bns.mean.data_ptr(), self.bns.invstd.data_ptr(), gam, bet,
                                                  g.data_ptr(), idx.data_ptr(), work.data_ptr(), B, P, self.cout, _st()),
            "bevf_bn_relu_group_max_idx_f32")
        return g, idx

    def bnb_request(self):
        """What a producing data-gradient conv needs to do this layer's first BatchNorm-backward pass in its epilogue."""
        st = self.bns
        return dict(x=st.xraw, y=st.y if (self.relu and st.has_res) else None, mean=st.mean, invstd=st.invstd,
                    gamma=self.bn.weight.detach() if self.bn.weight is not None else None,
                    beta=self.bn.bias.detach() if self.bn.bias is not None else None)

    def can_take_fused_dy(self) -> bool:
        return self.bn is not None and self.relu and self.cout % 4 == 0 and not bn_is_frozen(self.bn)

    def backward(self, dy, sink: GradSink, need_dx=True, add=None, fuse_next=None, pre=None):
        """dy: gradient of the layer output (modified in place).  Returns (dx or None, d_res or None); `add` is summed
        into dx (skip-connection gradient, fused into the data-gradient conv's epilogue when it can be).
        pre = (part, G): dy arrives with this layer's ReLU mask applied and its BatchNorm-backward sums as partials (the
        producing conv's epilogue did that pass).  fuse_next: the ConvBNLayer that consumes dx -- when this layer's data
        gradient is one fused-Winograd launch, its epilogue does THAT layer's pass; the return is then (dx, d_res, pre_next)."""
        d_res = None
        if self.bn is None:
            if self.relu:
                n4 = (self.M * self.cout + 3) // 4 * 4
                _ck(_lib().bevf_relu_mask_f32(dy.data_ptr(), self.y.data_ptr(), n4, _st()), "bevf_relu_mask_f32")
            dxraw = dy
        else:
            if pre is not None:
                dxraw, dgamma, dbeta = bn_backward_from_partials(dy, self.bns, self.bn, pre)
            else:
                dxraw, dgamma, dbeta = bn_train_backward(dy, self.bns, self.bn, relu=self.relu)
            sink.add(self.bn.weight, dgamma)
            sink.add(self.bn.bias, dbeta)
            if self.has_res:
                d_res = dy                                    # masked by the ReLU in place: gradient of the skip input
        if fuse_next is not None:
            dx, pre_next = self._conv_backward(dxraw, sink, need_dx, add, fuse_next)
            return dx, d_res, pre_next
        return self._conv_backward(dxraw, sink, need_dx, add), d_res

    def backward_from_groupmax(self, dg, gmax, idx, B: int, P: int, sink: GradSink):
        """Backward when this layer's output went straight into a max over the P rows of each of B groups (PointNet's
        last layer): dg / gmax / idx [B][cout].  The gradient is non-zero in one row per (group, channel), so BatchNorm's
        sums are gathered from those entries and no dense dY is ever built (bevf_gmax_bn_backward_f32)."""
        assert self.bn is not None and self.relu and not self.has_res and B * P == self.M
        st, dev = self.bns, dg.device
        dgm, dgamma, dbeta = _new(B * self.cout, dev), _new(self.cout, dev), _new(self.cout, dev)
        if LOWRANK_GMAX_BACKWARD and self.k == 1 and self.cin % 4 == 0 and self.cout % 4 == 0:
            return self._backward_from_groupmax_lowrank(dg, gmax, idx, B, P, sink, dgm, dgamma, dbeta)
        dxraw = _new(self.M * self.cout, dev)
        g = self.bn.weight.data_ptr() if self.bn.weight is not None else None
        _ck(_lib().bevf_gmax_bn_backward_f32(dg.data_ptr(), gmax.data_ptr(), idx.data_ptr(), st.xraw.data_ptr(), st.mean.data_ptr(),
                                             st.invstd.data_ptr(), g, dgm.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                             dxraw.data_ptr(), B, P, self.cout, self.cout, _st()), "bevf_gmax_bn_backward_f32")
        sink.add(self.bn.weight, dgamma[:self.cout])
        sink.add(self.bn.bias, dbeta[:self.cout])
        return self._conv_backward(dxraw, sink, True, None)

    def _backward_from_groupmax_lowrank(self, dg, gmax, idx, B, P, sink, dgm, dgamma, dbeta):
        """The same gradients without the dense dX [M][cout] (1.15 GB for PointNet's conv5) and with half the GEMM work.  With
        A [M][K] the layer input, W [C][K] its weight, x = A W^T + b its raw output, BatchNorm's backward is
            dX = S + 1 (beta')^T + (A W^T) diag(kappa),   kappa = -gamma invstd^2 dgamma / M,
                                                          beta' = -gamma invstd dbeta / M + kappa (b - mean),
        S = the B x C entries gamma invstd dg (one row per frame and channel, at the argmax).  Hence
            dW = S^T A + beta' colsum(A)^T + diag(kappa) W (A^T A)          -- one K x K Gram matrix instead of a C x K wgrad GEMM,
            dA = S W  + 1 (beta'^T W)      + A (W^T diag(kappa) W)          -- one M x K x K GEMM instead of M x C x K,
            db = colsum(S) + M beta' + kappa (W colsum(A)).
        (C = 1024, K = 512: 294 instead of 586 GFLOP, and no 1.15 GB tensor written and read twice.)"""
        st, dev, M, K, Cc = self.bns, dg.device, self.M, self.cin, self.cout
        _ck(_lib().bevf_gmax_bn_sums_f32(dg.data_ptr(), gmax.data_ptr(), idx.data_ptr(), st.xraw.data_ptr(), st.mean.data_ptr(),
                                         st.invstd.data_ptr(), dgm.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), B, P, Cc, Cc, _st()),
            "bevf_gmax_bn_sums_f32")
        dgamma, dbeta = dgamma[:Cc], dbeta[:Cc]
        sink.add(self.bn.weight, dgamma)
        sink.add(self.bn.bias, dbeta)
        w = self.conv.weight
        W2 = w.detach().reshape(Cc, K)
        invstd, mean = st.invstd[:Cc], st.mean[:Cc]
        gi = invstd if self.bn.weight is None else self.bn.weight.detach() * invstd
        kap = -(gi * invstd) * dgamma / M
        bias = self.conv.bias.detach() if self.conv.bias is not None else torch.zeros(Cc, device=dev)
        beta_p = -(gi * dbeta) / M + kap * (bias - mean)
        S = (gi.unsqueeze(0) * dgm[:B * Cc].view(B, Cc)).contiguous()                      # [B][C] (parameter-sized torch arithmetic)
        cs_a = colsum(self.x, M, K)                                                       # 1^T A
        if self.conv.bias is not None:
            sink.add(self.conv.bias, S.sum(0) + M * beta_p + kap * (W2 * cs_a.unsqueeze(0)).sum(1))
        # weight gradient: K x K Gram matrix on the pixel-GEMM kernel, the rest is C x K sized
        gram = conv_wgrad(self.x, self.x, M, 1, 1, K, K, 1, 1, 0).reshape(K, K)
        wg, _, _ = conv_raw(W2.contiguous().view(-1), gram.contiguous().view(-1), None, Cc, 1, 1, K, K, 1, 1, 0)
        sa = _new(Cc * K, dev)                                                            # S^T A over the B*C argmax rows, frames in order
        _ck(_lib().bevf_sparse_rows_wgrad_f32(S.data_ptr(), idx.data_ptr(), self.x.data_ptr(), sa.data_ptr(), B, P, Cc, K, _st()),
            "bevf_sparse_rows_wgrad_f32")
        dW = sa[:Cc * K].view(Cc, K) + beta_p.unsqueeze(1) * cs_a.unsqueeze(0) + kap.unsqueeze(1) * wg[:Cc * K].view(Cc, K)
        sink.add(w, dW.reshape(w.shape))
        # data gradient: A (W^T diag(kappa) W) + the row-constant term in ONE 1x1-conv launch, then the B*C sparse rows
        kw = (kap.unsqueeze(1) * W2).contiguous()
        gw = conv_wgrad(W2.contiguous().view(-1), kw.view(-1), Cc, 1, 1, K, K, 1, 1, 0).reshape(K, K)
        v = (beta_p.unsqueeze(1) * W2).sum(0).contiguous()
        dA, _, _ = conv_raw(self.x, gw.contiguous().view(-1), v, M, 1, 1, K, K, 1, 1, 0)
        # + S W at the argmax rows: channels sharing a row are added by the row's first channel, in order (no atomics: reproducible)
        _ck(_lib().bevf_sparse_rows_scatter_add_f32(S.data_ptr(), idx.data_ptr(), W2.contiguous().data_ptr(), dA.data_ptr(), B, P, Cc, K,
                                                    _st()), "bevf_sparse_rows_scatter_add_f32")
        return dA

    def _conv_backward(self, dxraw, sink: GradSink, need_dx=True, add=None, fuse_next=None):
        if self.conv.bias is not None:
            sink.add(self.conv.bias, colsum(dxraw, self.M, self.cout))
        dw = conv_wgrad(self.x, dxraw, self.N, self.H, self.W, self.cin, self.cout, self.k, self.stride, self.pad)
        w = self.conv.weight
        sink.add(w, dw.permute(0, 3, 1, 2).reshape(w.shape))
        dx, pre_next = None, None
        if need_dx:
            w4 = w if w.dim() == 4 else w.unsqueeze(-1)
            geom = (self.N, self.H, self.W, self.cin, self.cout, self.k, self.stride, self.pad)
            if (fuse_next is not None and FUSE_BN_BACKWARD and fuse_next.can_take_fused_dy() and fuse_next.cout == self.cin
                    and fuse_next.M == self.N * self.H * self.W and dgrad_can_fuse_bn(*geom)):
                dx, pre_next = conv_dgrad(dxraw, w4, *geom, add=add, bnb=fuse_next.bnb_request())
            else:
                dx = conv_dgrad(dxraw, w4, *geom, add=add)
        if fuse_next is not None:
            return dx, pre_next
        return dx


class LinearLayer:
    def __init__(self, lin: nn.Linear, relu: bool, perm: Tuple[int, int] = (0, 0)):
        self.lin, self.relu, self.perm = lin, relu, perm

    def forward(self, x, B):
        w = self.lin.weight.detach().contiguous()
        O, K = w.shape
        y = _new(B * O, x.device)
        L.linear(x, w, self.lin.bias.detach() if self.lin.bias is not None else None, y, B, K, O, self.relu, *self.perm)
        self.x, self.y, self.B = x, y, B
        if self.relu and self.perm == (0, 0):
            _trace_relu(y, B, O)
        return y

    def backward(self, dy, sink: GradSink, need_dx=True):
        w = self.lin.weight.detach().contiguous()
        O, K = w.shape
        dev = dy.device
        if self.relu:
            n4 = (self.B * O + 3) // 4 * 4
            _ck(_lib().bevf_relu_mask_f32(dy.data_ptr(), self.y.data_ptr(), n4, _st()), "bevf_relu_mask_f32")
        dx = _new(self.B * K, dev) if need_dx else None
        dw, db = _new(O * K, dev), _new(O, dev)
        work = _new(_lib().bevf_linear_bwd_work_floats(self.B, K, O), dev)
        _ck(_lib().bevf_linear_bwd_f32(dy.data_ptr(), self.x.data_ptr(), w.data_ptr(),
                                       dx.data_ptr() if dx is not None else None, dw.data_ptr(), db.data_ptr(),
                                       work.data_ptr(), self.B, K, O, self.perm[0], self.perm[1], _st()), "bevf_linear_bwd_f32")
        sink.add(self.lin.weight, dw[:O * K])
        sink.add(self.lin.bias, db[:O])
        return dx


class Bilinear:
    def forward(self, x, B, Hi, Wi, Cc, Ho, Wo, y=None, y_cs=None):
        self.geom = (B, Hi, Wi, Cc, Ho, Wo, y_cs or Cc)
        if y is None:
            y = _new(B * Ho * Wo * Cc, x.device)
        L.bilinear_nhwc(x, y, B, Hi, Wi, Cc, Cc, Ho, Wo, y_cs or Cc)
        return y

    def backward(self, dy):
        B, Hi, Wi, Cc, Ho, Wo, y_cs = self.geom
        dx = _zeros(B * Hi * Wi * Cc, dy.device)
        _ck(_lib().bevf_bilinear_bwd_nhwc_f32(dy.data_ptr(), dx.data_ptr(), B, Hi, Wi, Cc, Cc, Ho, Wo, y_cs, _st()),
            "bevf_bilinear_bwd_nhwc_f32")
        return dx


# ---- the detector graph --------------------------------------------------------------------------------------------------------

class DetectorTape:
    """Train-mode forward of FlexibleMultiModal3DDetector (bev + centernet) with everything backward needs."""

    def __init__(self, model):
        self.m = model

    # -- camera encoder ---------------------------------------------------------------------------------------------------------
    def _camera_forward(self, imgs):
        enc = self.m.camera_encoder
        if imgs.dim() == 5:
            B, n = imgs.shape[:2]
            x = imgs.reshape(B * n, *imgs.shape[2:]).contiguous().float()
        else:
            B, n = imgs.shape[0], 1
            x = imgs.contiguous().float()
        N, _, H, W = x.shape
        dev = x.device
        self.cam_geom_in = (N, H, W)
        self.imgs = x
        H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        w = enc.conv1.weight.detach()
        packed = torch.zeros(148, 64, device=dev)
        packed[:147] = w.reshape(64, 147).t()
        one, zero = torch.ones(64, device=dev), torch.zeros(64, device=dev)
        raw = _new(N * H1 * W1 * 64, dev)
        L.stem_conv7x7(x, packed.view(-1), one, zero, raw, N, H, W, relu=False)
        H2, W2 = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
        pooled = _new(N * H2 * W2 * 64, dev)
        self.pool_idx = torch.empty(N * H2 * W2 * 64, dtype=torch.uint8, device=dev)
        self.stem_fused = FUSE_POOL_BN_BACKWARD and not bn_is_frozen(enc.bn1)
        if self.stem_fused:
            # statistics only; BatchNorm + ReLU are evaluated inside the max-pool: the normalised stem map (1.1 GB) is never written --
            # nothing downstream reads it (the backward recomputes the ReLU mask from the raw conv output)
            _, self.stem_bn = bn_train_forward(raw, enc.bn1, N * H1 * W1, 64, relu=True, apply=False)
            s = self.stem_bn
            g = enc.bn1.weight.data_ptr() if enc.bn1.weight is not None else None
            b = enc.bn1.bias.data_ptr() if enc.bn1.bias is not None else None
            _ck(_lib().bevf_bn_relu_maxpool3x3s2_idx_f32(raw.data_ptr(), s.mean.data_ptr(), s.invstd.data_ptr(), g, b, pooled.data_ptr(),
                                                         self.pool_idx.data_ptr(), N, H1, W1, 64, _st()),
                "bevf_bn_relu_maxpool3x3s2_idx_f32")
        else:
            y, self.stem_bn = bn_train_forward(raw, enc.bn1, N * H1 * W1, 64, relu=True)
            _ck(_lib().bevf_maxpool3x3s2_idx_f32(y.data_ptr(), pooled.data_ptr(), self.pool_idx.data_ptr(), N, H1, W1, 64, _st()),
                "bevf_maxpool3x3s2_idx_f32")
        self.pool_geom = (N, H1, W1)
        cur, h, wd = pooled, H2, W2
        self.blocks = []
        for layer in (enc.layer1, enc.layer2, enc.layer3):
            for blk in layer:
                c1, c2 = ConvBNLayer(blk.conv1, blk.bn1, True), ConvBNLayer(blk.conv2, blk.bn2, True)
                down = ConvBNLayer(blk.downsample[0], blk.downsample[1], False) if blk.downsample is not None else None
                t, ho, wo = c1.forward(cur, N, h, wd)
                idt = cur
                if down is not None:
                    idt, _, _ = down.forward(cur, N, h, wd)
                out, _, _ = c2.forward(t, N, ho, wo, res=idt)
                self.blocks.append((c1, c2, down))
                cur, h, wd = out, ho, wo
        self.proj = ConvBNLayer(enc.channel_proj[0], enc.channel_proj[1], True)
        feat, _, _ = self.proj.forward(cur, N, h, wd)
        return feat, (B, n, h, wd)

    def _camera_backward(self, dfeat, sink):
        enc = self.m.camera_encoder
        d, _ = self.proj.backward(dfeat, sink)
        rev = list(reversed(self.blocks))
        pre = None                                             # BatchNorm-backward partials that arrive WITH d (or None)
        for i, (c1, c2, down) in enumerate(rev):
            # c2's data-gradient conv does the first BatchNorm-backward pass of c1 (mask + sums) in its epilogue
            dt, d_res, pre1 = c2.backward(d, sink, fuse_next=c1, pre=pre)      # d_res: gradient reaching the skip connection
            pre = None
            if down is not None:
                dx, _ = c1.backward(dt, sink, pre=pre1)
                dd, _ = down.backward(d_res, sink)
                add_(dx, dd, dx.numel())
            elif i + 1 < len(rev):                             # identity skip: summed in the dgrad conv's epilogue, which then
                dx, _, pre = c1.backward(dt, sink, add=d_res, fuse_next=rev[i + 1][1], pre=pre1)   # serves the previous block's c2
            else:
                dx, _ = c1.backward(dt, sink, add=d_res, pre=pre1)
            d = dx
            if i % 2 == 1:
                sink.ready()                                   # one ResNet stage done: its gradients can travel
        N, H1, W1 = self.pool_geom
        if self.stem_fused:
            # max-pool backward + BatchNorm/ReLU backward in one pair of passes: the dense dY of the stem map (1.1 GB at 48 images of
            # 448x800) is gathered from the pooled gradient on the fly, never written (bit-identical to the two-kernel chain below)
            s = self.stem_bn
            work = _new(_lib().bevf_bn_work_floats(64), d.device)
            dgamma, dbeta, draw = _new(64, d.device), _new(64, d.device), _new(N * H1 * W1 * 64, d.device)
            g = enc.bn1.weight.data_ptr() if enc.bn1.weight is not None else None
            b = enc.bn1.bias.data_ptr() if enc.bn1.bias is not None else None
            _ck(_lib().bevf_pool_bn_backward_f32(d.data_ptr(), self.pool_idx.data_ptr(), s.xraw.data_ptr(), s.mean.data_ptr(),
                                                 s.invstd.data_ptr(), g, b, work.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                 draw.data_ptr(), N, H1, W1, 64, _st()), "bevf_pool_bn_backward_f32")
        else:
            dpool_in = _new(N * H1 * W1 * 64, d.device)
            _ck(_lib().bevf_maxpool3x3s2_bwd_f32(d.data_ptr(), self.pool_idx.data_ptr(), dpool_in.data_ptr(), N, H1, W1, 64, _st()),
                "bevf_maxpool3x3s2_bwd_f32")
            draw, dgamma, dbeta = bn_train_backward(dpool_in, self.stem_bn, enc.bn1, relu=True)
        sink.add(enc.bn1.weight, dgamma)
        sink.add(enc.bn1.bias, dbeta)
        # stem weight gradient: direct MFMA kernel on the image patches (no im2col matrix), dW as [64][160]
        Ni, H, W = self.cam_geom_in
        dwbuf = _zeros(64 * 160, d.device)
        with E._span("conv_wgrad_f32", flops=2.0 * N * H1 * W1 * 64 * 147):
            _ck(_lib().bevf_stem_wgrad_f32(self.imgs.data_ptr(), draw.data_ptr(), dwbuf.data_ptr(), Ni, H, W, _st()),
                "bevf_stem_wgrad_f32")
        sink.add(enc.conv1.weight, dwbuf[:64 * 160].view(64, 160)[:, :147].reshape(64, 3, 7, 7))

    # -- PointNet ----------------------------------------------------------------------------------------------------------------
    def _lidar_forward(self, pts):
        enc = self.m.lidar_encoder
        rows = enc._rows(pts)
        B, Np, Cc = rows.shape
        M = B * Np
        self.pn_geom = (B, Np, Cc)
        self.pn_first = PointFirstLayer(enc.conv1, enc.bn1)          # bn*: BatchNorm1d, or Identity under use_bn=False
        a = self.pn_first.forward(rows, M, Cc)
        self.pn_layers = []
        for i in range(2, 5):
            lyr = ConvBNLayer(getattr(enc, f"conv{i}"), _bn_or_none(getattr(enc, f"bn{i}")), True)
            a, _, _ = lyr.forward(a, M, 1, 1)
            self.pn_layers.append(lyr)
        lyr = ConvBNLayer(enc.conv5, _bn_or_none(enc.bn5), True)
        self.pn_layers.append(lyr)
        self.pn_fused_max = lyr.bn is not None and not bn_is_frozen(lyr.bn)
        if self.pn_fused_max:                                        # last layer: BN + ReLU + max over points, fused
            g, self.pn_idx = lyr.forward_groupmax(a, B, Np)
        else:                                                        # use_bn=False / frozen statistics: the activation is written, then max + argmax
            a, _, _ = lyr.forward(a, M, 1, 1)
            g, self.pn_idx = group_max_with_index(a, B, Np, lyr.cout)
        self.pn_g = g
        return g

    def _lidar_backward(self, dg, sink):
        B, Np, Cc = self.pn_geom
        last = self.pn_layers[-1]
        if self.pn_fused_max:
            d = last.backward_from_groupmax(dg.contiguous(), self.pn_g, self.pn_idx, B, Np, sink)
        else:
            d = group_max_scatter(dg.contiguous(), self.pn_idx, B, Np, last.cout)
            d, _ = last.backward(d, sink)
        for lyr in reversed(self.pn_layers[:-1]):
            d, _ = lyr.backward(d, sink)
        self.pn_first.backward(d, sink)

    # -- radar: shared per-sweep MLP + max, concat -> Linear (ref src/encoders.py:628-661) --------------------------------------
    def _radar_forward(self, radars):
        renc = self.m.radar_encoder
        if renc.fusion_method not in ("concat", "max", "mean"):
            raise ValueError(f"Unknown fusion method: {renc.fusion_method}")
        enc = renc.radar_encoder
        self.rad_sweeps = []
        feats = []
        for pts in radars:
            rows = enc._rows(pts)
            B, Np, Cc = rows.shape
            M = B * Np
            first = PointFirstLayer(enc.conv1, enc.bn1)               # bn*: BatchNorm1d, or Identity under use_bn=False
            a = first.forward(rows, M, Cc)
            layers = []
            for i in range(2, 5):
                lyr = ConvBNLayer(getattr(enc, f"conv{i}"), _bn_or_none(getattr(enc, f"bn{i}")), True)
                a, _, _ = lyr.forward(a, M, 1, 1)
                layers.append(lyr)
            feat = layers[-1].cout
            g, idx = group_max_with_index(a, B, Np, feat)
            feats.append(g[:B * feat].view(B, feat))
            self.rad_sweeps.append((first, layers, idx, (B, Np, feat)))
        per = torch.stack(feats, dim=1).contiguous()                      # (B, R, feat) -- layout copy only
        B, R, feat = per.shape
        self.rad_geom = (B, R, feat)
        if renc.fusion_method == "max":                                   # ref src/encoders.py:654-655
            out = _new(B * feat, per.device)
            self.rad_fuse_idx = torch.empty(B * feat, dtype=torch.int32, device=per.device)
            gwork = torch.empty(_lib().bevf_group_max_idx_work_bytes(B, R, feat), dtype=torch.uint8, device=per.device)
            _ck(_lib().bevf_group_max_idx_f32(per.data_ptr(), out.data_ptr(), self.rad_fuse_idx.data_ptr(), gwork.data_ptr(),
                                              B, R, feat, _st()), "bevf_group_max_idx_f32")
            return out, B
        if renc.fusion_method == "mean":                                  # ref src/encoders.py:656-657
            out = _new(B * feat, per.device)
            L.cam_mean(per.view(-1), out, B, R, 1, feat)
            return out, B
        if R * feat != renc.fusion_fc.weight.shape[1]:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({B}x{R * feat} and "
                               f"{renc.fusion_fc.weight.shape[1]}x{renc.fusion_fc.weight.shape[0]})")
        self.rad_fc = LinearLayer(renc.fusion_fc, False)
        return self.rad_fc.forward(per.view(-1), B), B

    def _radar_backward(self, dfeat, sink):
        renc = self.m.radar_encoder
        enc = renc.radar_encoder
        B, R, feat = self.rad_geom
        if renc.fusion_method == "max":            # the gradient goes to the sweep that held the maximum (first one on ties)
            dper = _zeros(B * R * feat, dfeat.device)
            _ck(_lib().bevf_group_max_bwd_f32(dfeat.data_ptr(), self.rad_fuse_idx.data_ptr(), dper.data_ptr(), B, R, feat, _st()),
                "bevf_group_max_bwd_f32")
        elif renc.fusion_method == "mean":         # every sweep receives dfeat / R
            dper = _new(B * R * feat, dfeat.device)
            _ck(_lib().bevf_cam_mean_bwd_f32(dfeat.data_ptr(), dper.data_ptr(), B, R, 1, feat, _st()), "bevf_cam_mean_bwd_f32")
        else:
            dper = self.rad_fc.backward(dfeat, sink)                        # [B][R][feat]
        for r in reversed(range(R)):
            first, layers, idx, (B, Np, feat) = self.rad_sweeps[r]
            dg = dper[:B * R * feat].view(B, R, feat)[:, r].contiguous().view(-1)
            d = group_max_scatter(dg, idx, B, Np, feat)
            for lyr in reversed(layers):
                d, _ = lyr.backward(d, sink)
            first.backward(d, sink)

    # -- fusion + head -----------------------------------------------------------------------------------------------------------
    def forward(self, imgs, pts, radars):
        m = self.m
        fus = m.fusion
        Sh, Sw, bc = fus.bev_h, fus.bev_w, fus.bev_channels
        P = Sh * Sw
        self.has_cam = m.use_camera and imgs is not None
        self.has_lid = m.use_lidar and pts is not None
        self.has_rad = m.use_radar and radars is not None
        nmod = int(self.has_cam) + int(self.has_lid) + int(self.has_rad)
        if nmod == 0:
            raise ValueError("No modality features provided")
        dev = imgs.device if self.has_cam else (pts.device if self.has_lid else radars[0].device)
        cam_feat = lid_feat = rad_feat = None
        if self.has_rad:
            rad_feat, B = self._radar_forward(radars)
        if self.has_cam:
            cam_feat, (B, ncam, Hc, Wc) = self._camera_forward(imgs)
        if self.has_lid:
            lid_feat = self._lidar_forward(pts)
            B = self.pn_geom[0]
        fused = self._fusion_forward(cam_feat, (B, ncam, Hc, Wc) if self.has_cam else None, lid_feat, rad_feat, B, dev)
        return self._head_forward(fused, B, Sh, Sw)

    def _fusion_forward(self, cam_feat, cam_geom, lid_feat, rad_feat, B, dev):
        """FlexibleBEVFusion under train-mode BatchNorm (ref src/fusion.py:209-297): NHWC camera features [B*ncam*Hc*Wc*C] with
        cam_geom = (B, ncam, Hc, Wc), LiDAR (B, C_l) and radar (B, C_r) vectors -> fused NHWC map [B*Sh*Sw*bev_channels]."""
        fus = self.m.fusion
        Sh, Sw, bc = fus.bev_h, fus.bev_w, fus.bev_channels
        P = Sh * Sw
        self.has_cam, self.has_lid, self.has_rad = cam_feat is not None, lid_feat is not None, rad_feat is not None
        nmod = int(self.has_cam) + int(self.has_lid) + int(self.has_rad)
        if nmod == 0:
            raise ValueError("No modality features provided")
        ccs = bc * nmod
        if ccs != fus.bev_fusion[0].weight.shape[1]:
            raise RuntimeError(f"expected input to have {fus.bev_fusion[0].weight.shape[1]} channels, but got {ccs} channels instead")
        concat = _new(B * P * ccs, dev)
        slot = 0
        self.B, self.ccs = B, ccs
        self.S = (Sh, Sw)
        if self.has_cam:
            _, ncam, Hc, Wc = cam_geom
            Cc = fus.camera_proj[0].weight.shape[1]
            self.cam_pool_geom = (B, ncam, Hc * Wc, Cc)
            pooled = cam_feat
            if ncam > 1:
                pooled = _new(B * Hc * Wc * Cc, dev)
                L.cam_mean(cam_feat, pooled, B, ncam, Hc * Wc, Cc)
            self.cp1 = ConvBNLayer(fus.camera_proj[0], fus.camera_proj[1], True)
            self.cp2 = ConvBNLayer(fus.camera_proj[3], fus.camera_proj[4], True)
            t1, _, _ = self.cp1.forward(pooled, B, Hc, Wc)
            t2, _, _ = self.cp2.forward(t1, B, Hc, Wc)
            self.cam_resize = Bilinear()
            self.cam_resize.forward(t2, B, Hc, Wc, bc, Sh, Sw, y=concat[slot * bc:], y_cs=ccs)
            self.cam_slot = slot
            slot += 1
        if self.has_lid:
            s0 = fus.lidar_start_size
            self.li0 = LinearLayer(fus.lidar_init[0], True)
            O = fus.lidar_init[2].weight.shape[0]
            ch = O // (s0 * s0)
            self.li2 = LinearLayer(fus.lidar_init[2], False, (s0 * s0, ch))
            hid = self.li0.forward(lid_feat, B)
            grid0 = self.li2.forward(hid, B)
            self.lu1 = ConvBNLayer(fus.lidar_upsample[0], fus.lidar_upsample[1], True)
            self.lu2 = ConvBNLayer(fus.lidar_upsample[4], fus.lidar_upsample[5], True)
            g1, _, _ = self.lu1.forward(grid0, B, s0, s0)
            self.lid_up = Bilinear()
            s1 = 2 * s0
            g2 = self.lid_up.forward(g1, B, s0, s0, self.lu1.cout, s1, s1)
            g3, _, _ = self.lu2.forward(g2, B, s1, s1)
            self.lid_s1 = s1
            self.lid_resize = None
            if (s1, s1) == (Sh, Sw):
                # copy into the concat slice (bilinear at identical size is the identity map)
                self.lid_resize = Bilinear()
                self.lid_resize.forward(g3, B, s1, s1, bc, Sh, Sw, y=concat[slot * bc:], y_cs=ccs)
            else:
                self.lid_resize = Bilinear()
                self.lid_resize.forward(g3, B, s1, s1, bc, Sh, Sw, y=concat[slot * bc:], y_cs=ccs)
            self.lid_slot = slot
            slot += 1
        if self.has_rad:
            self.rp = LinearLayer(fus.radar_proj[0], True)
            rv = self.rp.forward(rad_feat, B)
            r0 = _new(B * P * bc, dev)
            L.broadcast_nhwc(rv, r0, B, P, bc, bc)
            self.rr1 = ConvBNLayer(fus.radar_refine[0], fus.radar_refine[1], True)
            self.rr2 = ConvBNLayer(fus.radar_refine[3], fus.radar_refine[4], True)
            r1, _, _ = self.rr1.forward(r0, B, Sh, Sw)
            r2, _, _ = self.rr2.forward(r1, B, Sh, Sw)
            concat[:B * P * ccs].view(B * P, ccs)[:, slot * bc:(slot + 1) * bc] = r2[:B * P * bc].view(B * P, bc)   # slice copy
            self.rad_slot = slot
            slot += 1
        self.f1 = ConvBNLayer(fus.bev_fusion[0], fus.bev_fusion[1], True)
        self.f2 = ConvBNLayer(fus.bev_fusion[3], fus.bev_fusion[4], True)
        a1, _, _ = self.f1.forward(concat, B, Sh, Sw)
        fused, _, _ = self.f2.forward(a1, B, Sh, Sw)
        return fused

    def _head_forward(self, fused, B, Sh, Sw):
        """CenterNetHead (ref src/fusion.py:869-884): the five 3x3 branches as one conv (weights concatenated along Cout), then
        the tail kernel."""
        head = self.m.det_head
        dev = fused.device
        P = Sh * Sw
        self.B = B
        convs3 = [getattr(head, f"{n}_head")[0] for n in E.HEAD_BRANCHES]
        convs1 = [getattr(head, f"{n}_head")[2] for n in E.HEAD_BRANCHES]
        self.hc = convs3[0].weight.shape[0]
        w3 = torch.cat([c.weight.detach() for c in convs3], 0)
        b3 = torch.cat([c.bias.detach() for c in convs3], 0).contiguous()
        self.head_w3 = w3
        hid, _, _ = conv_raw(fused, w3.permute(0, 2, 3, 1).contiguous().view(-1), b3, B, Sh, Sw, w3.shape[1], w3.shape[0],
                             3, 1, 1, relu=True)
        self.head_in, self.head_hid = fused, hid
        _trace_relu(hid, B * P, w3.shape[0])
        self.cs = [c.weight.shape[0] for c in convs1]
        self.w1 = torch.cat([c.weight.detach().reshape(c.weight.shape[0], self.hc) for c in convs1], 0).contiguous()
        b1 = torch.cat([c.bias.detach() for c in convs1], 0).contiguous()
        outs = [torch.empty(B, c, Sh, Sw, device=dev) for c in self.cs]
        L.head_tail(hid, self.w1, b1, outs, B, P, self.hc, self.cs, self.cs[0])
        self.outs = outs
        self.S = (Sh, Sw)
        return outs

    def backward(self, douts: List[torch.Tensor], reducer=None) -> GradSink:
        sink = GradSink(reducer)
        dfused, pre_f2 = self._head_backward(douts, sink, self.f2)

        def camera(dfeat):
            sink.ready()              # head, fusion, radar, LiDAR (the 164 MB dense layer): reduce under the camera trunk
            self._camera_backward(dfeat, sink)

        self._fusion_backward(dfused, pre_f2, sink, on_radar=lambda d: self._radar_backward(d, sink),
                              on_lidar=lambda d: self._lidar_backward(d, sink), on_camera=camera)
        sink.finish()
        return sink

    def _head_backward(self, douts: List[torch.Tensor], sink: GradSink, f2=None):
        """-> (gradient of the fused NHWC map, BatchNorm-backward partials for `f2` or None)."""
        m = self.m
        B, (Sh, Sw) = self.B, self.S
        P = Sh * Sw
        dev = self.outs[0].device
        head = m.det_head
        convs3 = [getattr(head, f"{n}_head")[0] for n in E.HEAD_BRANCHES]
        convs1 = [getattr(head, f"{n}_head")[2] for n in E.HEAD_BRANCHES]
        ctot = sum(self.cs)
        dhid = _new(B * P * 5 * self.hc, dev)
        dw1, db1 = _zeros(ctot * self.hc, dev), _zeros(ctot, dev)
        d = L.HeadBwdDesc()
        d.hid, d.w, d.out0 = self.head_hid.data_ptr(), self.w1.data_ptr(), self.outs[0].data_ptr()
        keep = []
        for k in range(5):
            g = douts[k]
            g = torch.zeros_like(self.outs[k]) if g is None else g.contiguous().float()
            keep.append(g)
            d.dout[k], d.c[k] = g.data_ptr(), self.cs[k]
        d.dhid, d.dw, d.db, d.B, d.P, d.hc, d.n_sigmoid = dhid.data_ptr(), dw1.data_ptr(), db1.data_ptr(), B, P, self.hc, self.cs[0]
        _ck(_lib().bevf_head_tail_bwd_f32(C.byref(d), _st()), "bevf_head_tail_bwd_f32")
        o = 0
        for k, c1 in enumerate(convs1):
            n = self.cs[k]
            sink.add(c1.weight, dw1[o * self.hc:(o + n) * self.hc])
            sink.add(c1.bias, db1[o:o + n])
            o += n
        # fused 3x3 head conv: ReLU mask, bias / weight / data gradients, split back per branch
        c5 = 5 * self.hc
        n4 = (B * P * c5 + 3) // 4 * 4
        _ck(_lib().bevf_relu_mask_f32(dhid.data_ptr(), self.head_hid.data_ptr(), n4, _st()), "bevf_relu_mask_f32")
        db3 = colsum(dhid, B * P, c5)
        cin = self.head_w3.shape[1]
        dw3 = conv_wgrad(self.head_in, dhid, B, Sh, Sw, cin, c5, 3, 1, 1).permute(0, 3, 1, 2)
        for k, c3 in enumerate(convs3):
            sink.add(c3.weight, dw3[k * self.hc:(k + 1) * self.hc])
            sink.add(c3.bias, db3[k * self.hc:(k + 1) * self.hc])
        pre_f2 = None
        if FUSE_BN_BACKWARD and f2 is not None and f2.can_take_fused_dy() and dgrad_can_fuse_bn(B, Sh, Sw, cin, c5, 3, 1, 1):
            dfused, pre_f2 = conv_dgrad(dhid, self.head_w3, B, Sh, Sw, cin, c5, 3, 1, 1, bnb=f2.bnb_request())
        else:
            dfused = conv_dgrad(dhid, self.head_w3, B, Sh, Sw, cin, c5, 3, 1, 1)
        return dfused, pre_f2

    def _fusion_backward(self, dfused, pre_f2, sink: GradSink, on_radar=None, on_lidar=None, on_camera=None):
        """Backward of `_fusion_forward`.  Each modality's input gradient (radar (B*C_r), LiDAR (B*C_l), camera NHWC) goes to its
        callback as soon as it exists -- the detector continues into that encoder there -- and is returned as well."""
        m = self.m
        B, (Sh, Sw) = self.B, self.S
        P = Sh * Sw
        dev = dfused.device
        drad = dlid = dfeat = None
        da1, _, pre_f1 = self.f2.backward(dfused, sink, fuse_next=self.f1, pre=pre_f2)
        dconcat, _ = self.f1.backward(da1, sink, pre=pre_f1)
        bc = m.fusion.bev_channels
        if self.has_rad:
            ccs = self.ccs
            dr2 = dconcat[:B * P * ccs].view(B * P, ccs)[:, self.rad_slot * bc:(self.rad_slot + 1) * bc].contiguous().view(-1)
            dr1, _ = self.rr2.backward(dr2, sink)
            dr0, _ = self.rr1.backward(dr1, sink)
            drv = torch.cat([colsum(dr0[b * P * bc:], P, bc) for b in range(B)])      # d(broadcast) = sum over cells
            drad = self.rp.backward(drv.contiguous(), sink)
            if on_radar is not None:
                on_radar(drad)
        if self.has_lid:
            dg3 = self.lid_resize.backward(dconcat[self.lid_slot * bc:])
            dg2, _ = self.lu2.backward(dg3, sink)
            dg1 = self.lid_up.backward(dg2)
            dgrid0, _ = self.lu1.backward(dg1, sink)
            dhid_l = self.li2.backward(dgrid0, sink)
            dlid = self.li0.backward(dhid_l, sink)
            if on_lidar is not None:
                on_lidar(dlid)
        if self.has_cam:
            dt2 = self.cam_resize.backward(dconcat[self.cam_slot * bc:])
            dt1, _ = self.cp2.backward(dt2, sink)
            dpooled, _ = self.cp1.backward(dt1, sink)
            Bc, ncam, Pc, Cc = self.cam_pool_geom
            dfeat = dpooled
            if ncam > 1:
                dfeat = _new(Bc * ncam * Pc * Cc, dev)
                _ck(_lib().bevf_cam_mean_bwd_f32(dpooled.data_ptr(), dfeat.data_ptr(), Bc, ncam, Pc, Cc, _st()),
                    "bevf_cam_mean_bwd_f32")
            if on_camera is not None:
                on_camera(dfeat)
        return drad, dlid, dfeat


_GRAD_REDUCER = None


def set_grad_reducer(reducer) -> None:
    """Data-parallel training: install a replicas.GradReducer and the detector's backward averages its gradients over
    the ranks itself, bucket by bucket as they become final, overlapped with the rest of the backward (None: off)."""
    global _GRAD_REDUCER
    _GRAD_REDUCER = reducer


class _DetectorTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, imgs, pts, radars, *params):
        tape = DetectorTape(model)
        with torch.no_grad():
            outs = tape.forward(imgs, pts, radars)
        ctx.tape, ctx.params = tape, params
        # The tape keeps its own tensors; autograd gets fresh aliases.  Returning the tape's objects would close a
        # reference cycle through C++ (output -> grad_fn -> ctx.tape -> output) that the garbage collector cannot
        # see: every step's activations (~11 GiB at config 4) would stay allocated for ever.
        return tuple(o.detach() for o in outs)

    @staticmethod
    def backward(ctx, *douts):
        if ctx.tape is None:
            raise RuntimeError("Trying to backward through the detector a second time: its saved activations were freed")
        global _ZPOOL
        with torch.no_grad():
            pool = _ZPOOL = _ZeroPool(next(g for g in douts if g is not None).device)
            try:
                sink = ctx.tape.backward(list(douts), _GRAD_REDUCER)
            finally:
                _ZPOOL = None
        ctx.tape = None                                   # activations are dead now: hand them back to the allocator
        return (None, None, None, None, *_param_grads(sink, ctx.params, pool))


def _owned(g, pool):
    """Never hand out a view of the shared zero pool."""
    if g is not None and g.untyped_storage().data_ptr() == pool.buf.untyped_storage().data_ptr():
        return g.clone()
    return g


def _param_grads(sink: GradSink, params, pool) -> list:
    grads = []
    for p in params:
        g = sink.get(p)
        if g is not None:
            g = _owned(g.reshape(p.shape).contiguous(), pool)
        grads.append(g)
    return grads


def any_bn_training(module: nn.Module) -> bool:
    return any(isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training for m in module.modules())


def wants_train_path(module: nn.Module) -> bool:
    """A module in train mode takes the tape when BatchNorm runs on batch statistics somewhere in it (then even under no_grad: the running
    buffers move, as in torch) or when autograd is recording -- use_bn=False encoders (ref src/encoders.py:258-269, 520-529) and modules
    whose BatchNorm layers were all put in eval mode still train their weights; the eval engines keep no gradient path."""
    return any_bn_training(module) or torch.is_grad_enabled()


def detector_train_forward(model, imgs, pts, radars) -> Dict[str, torch.Tensor]:
    params = [p for p in model.parameters() if p.requires_grad]
    outs = _DetectorTrainFn.apply(model, imgs, pts, radars, *params)
    return dict(zip(E.HEAD_BRANCHES, outs))


# ---- stand-alone modules under train-mode BatchNorm (ref src/encoders.py:792-846 calls freshly built encoders, i.e. in train mode) ----

class _ModuleTrainFn(torch.autograd.Function):
    """One encoder / fusion / head module used outside the detector: the matching part of DetectorTape.
    `run_fwd()` -> (tape, outputs); `run_bwd(tape, douts, sink)` -> one gradient per tensor input (None where there is none)."""

    @staticmethod
    def forward(ctx, run_fwd, run_bwd, n_in, *tensors):
        with torch.no_grad():
            tape, outs = run_fwd()
        ctx.tape, ctx.run_bwd, ctx.n_in, ctx.params = tape, run_bwd, n_in, tensors[n_in:]
        return tuple(o.detach() for o in outs)                    # fresh aliases: see _DetectorTrainFn.forward

    @staticmethod
    def backward(ctx, *douts):
        if ctx.tape is None:
            raise RuntimeError("Trying to backward through the module a second time: its saved activations were freed")
        global _ZPOOL
        with torch.no_grad():
            pool = _ZPOOL = _ZeroPool(next(g for g in douts if g is not None).device)
            try:
                sink = GradSink(None)
                dins = ctx.run_bwd(ctx.tape, list(douts), sink)
                sink.finish()
            finally:
                _ZPOOL = None
        ctx.tape = None
        need = ctx.needs_input_grad[3:3 + ctx.n_in]
        dins = [_owned(d, pool) if (n and d is not None) else None for d, n in zip(dins, need)]
        return (None, None, None, *dins, *_param_grads(sink, ctx.params, pool))


def _no_input_grad(x, what: str) -> None:
    if isinstance(x, torch.Tensor) and x.requires_grad:
        raise L.BevfError(f"training: {what} has no gradient path on the device (the reference's training loop never asks for "
                          "one, ref src/train_detect.py:401-434); detach the input")


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    """(N,C,H,W) -> flat NHWC fp32 buffer."""
    return E.to_nhwc(x.float().contiguous()).float().reshape(-1)


def camera_encoder_train_forward(enc, x: torch.Tensor) -> torch.Tensor:
    """ResNetCameraEncoder.forward under train-mode BatchNorm (ref src/encoders.py:133-172): batch statistics, running buffers
    updated, gradients for every trainable parameter."""
    _no_input_grad(x, "the camera images")
    five_d = x.dim() == 5
    geom = {}

    def fwd():
        tape = DetectorTape(SimpleNamespace(camera_encoder=enc))
        feat, (B, n, h, w) = tape._camera_forward(x)
        geom["g"] = (B, n, h, w)
        return tape, (E.to_nchw(feat, B * n, tape.proj.cout, h, w),)

    def bwd(tape, douts, sink):
        B, n, h, w = geom["g"]
        tape._camera_backward(_nhwc(douts[0].reshape(B * n, tape.proj.cout, h, w)), sink)
        return [None]

    (out,) = _ModuleTrainFn.apply(fwd, bwd, 1, x, *[p for p in enc.parameters() if p.requires_grad])
    B, n, h, w = geom["g"]
    return out.view(B, n, out.shape[1], h, w) if five_d else out


def pointnet_train_forward(enc, x: torch.Tensor) -> torch.Tensor:
    """PointNetLiDAREncoder.forward under train-mode BatchNorm (ref src/encoders.py:271-306) -> (B, feat_dim)."""
    _no_input_grad(x, "the LiDAR points")
    if getattr(enc, "return_point_features", False):
        raise L.BevfError("training: PointNetLiDAREncoder(return_point_features=True) has no train-mode path on the device "
                          "(the detector uses the global feature, ref src/fusion.py:1105-1108); call .eval() for per-point features")

    def fwd():
        tape = DetectorTape(SimpleNamespace(lidar_encoder=enc))
        g = tape._lidar_forward(x)
        B, feat = tape.pn_geom[0], tape.pn_layers[-1].cout
        return tape, (g.reshape(-1)[:B * feat].view(B, feat),)

    def bwd(tape, douts, sink):
        tape._lidar_backward(douts[0].contiguous().float().reshape(-1), sink)
        return [None]

    (out,) = _ModuleTrainFn.apply(fwd, bwd, 1, x, *[p for p in enc.parameters() if p.requires_grad])
    return out


def vfe_train_forward(layer, x: torch.Tensor) -> torch.Tensor:
    """VFELayer.forward under train-mode BatchNorm (ref src/encoders.py:431-455): Linear -> BatchNorm1d over all B*Nv*P rows (padding
    rows included, as the reference) -> ReLU -> max over the P points of a voxel -> (B, Nv, out_channels)."""
    _no_input_grad(x, "the voxel points")
    B, Nv, P, Cc = x.shape                                        # a 3-D input raises ValueError exactly like the reference
    if Cc > 16:
        raise L.BevfError(f"training: VFELayer(in_channels={Cc}) has a train-mode path for point features of <= 16 channels only")
    G, M = B * Nv, B * Nv * P
    lin = SimpleNamespace(weight=layer.linear.weight, bias=layer.linear.bias)      # same Parameters: gradients land on them

    def fwd():
        first = PointFirstLayer(lin, layer.bn)
        a = first.forward(x.detach().float().contiguous().view(M, Cc), M, Cc)
        g, idx = group_max_with_index(a, G, P, first.c0)
        return (first, idx), (g[:G * first.c0].view(B, Nv, first.c0),)

    def bwd(tape, douts, sink):
        first, idx = tape
        first.backward(group_max_scatter(douts[0].contiguous().float().reshape(-1), idx, G, P, first.c0), sink)
        return [None]

    (out,) = _ModuleTrainFn.apply(fwd, bwd, 1, x, *[p for p in layer.parameters() if p.requires_grad])
    return out


def radar_train_forward(enc, radar_list) -> torch.Tensor:
    """MultiRadarEncoder.forward (ref src/encoders.py:619-661), or one RadarEncoder (ref :527-557, `radar_list` a single tensor),
    under train-mode BatchNorm -> (B, feat_dim)."""
    single = isinstance(radar_list, torch.Tensor)
    sweeps = [radar_list] if single else list(radar_list)
    for r in sweeps:
        _no_input_grad(r, "the radar points")
    # one encoder alone = the shared encoder over one sweep, "max" over that single sweep being the identity
    renc = SimpleNamespace(radar_encoder=enc, fusion_method="max") if single else enc

    def fwd():
        tape = DetectorTape(SimpleNamespace(radar_encoder=renc))
        out, B = tape._radar_forward(sweeps)
        feat = tape.rad_geom[2] if renc.fusion_method != "concat" else renc.fusion_fc.weight.shape[0]
        return tape, (out.reshape(-1)[:B * feat].view(B, feat),)

    def bwd(tape, douts, sink):
        tape._radar_backward(douts[0].contiguous().float().reshape(-1), sink)
        return [None] * len(sweeps)

    (out,) = _ModuleTrainFn.apply(fwd, bwd, len(sweeps), *sweeps, *[p for p in enc.parameters() if p.requires_grad])
    return out


def fusion_train_forward(fus, camera_features=None, lidar_features=None, radar_features=None) -> torch.Tensor:
    """FlexibleBEVFusion.forward under train-mode BatchNorm (ref src/fusion.py:209-297) -> (B, bev_channels, bev_h, bev_w), with
    gradients for the parameters AND for the three feature inputs."""
    cam = camera_features if fus.use_camera else None
    lid = lidar_features if fus.use_lidar else None
    rad = radar_features if fus.use_radar else None
    first = next((t for t in (cam, lid, rad) if t is not None), None)
    if first is None:
        raise ValueError("No modality features provided")
    B, dev = first.shape[0], first.device
    geom = None
    if cam is not None:
        geom = (B, cam.shape[1], cam.shape[3], cam.shape[4]) if cam.dim() == 5 else (B, 1, cam.shape[2], cam.shape[3])

    def fwd():
        tape = DetectorTape(SimpleNamespace(fusion=fus))
        cam_nhwc = None
        if cam is not None:
            _, n, h, w = geom
            cam_nhwc = _nhwc(cam.detach().reshape(B * n, -1, h, w))
        fused = tape._fusion_forward(cam_nhwc, geom, None if lid is None else lid.detach().float().contiguous(),
                                     None if rad is None else rad.detach().float().contiguous(), B, dev)
        return tape, (E.to_nchw(fused, B, fus.bev_channels, fus.bev_h, fus.bev_w),)

    def bwd(tape, douts, sink):
        drad, dlid, dcam = tape._fusion_backward(_nhwc(douts[0]), None, sink)
        if dcam is not None:
            _, n, h, w = geom
            Cc = tape.cam_pool_geom[3]
            dcam = E.to_nchw(dcam, B * n, Cc, h, w).view(cam.shape)
        if dlid is not None:
            dlid = dlid.reshape(-1)[:lid.numel()].view(lid.shape)
        if drad is not None:
            drad = drad.reshape(-1)[:rad.numel()].view(rad.shape)
        return [dcam, dlid, drad]

    (out,) = _ModuleTrainFn.apply(fwd, bwd, 3, cam, lid, rad, *[p for p in fus.parameters() if p.requires_grad])
    return out


def head_train_forward(head, x: torch.Tensor) -> Dict[str, torch.Tensor]:
    """CenterNetHead.forward with a gradient path (ref src/fusion.py:869-884; the head has no BatchNorm, so train and eval mode
    compute the same values): gradients for its parameters and for the BEV map `x` (B, C, H, W)."""
    B, _, Sh, Sw = x.shape

    def fwd():
        tape = DetectorTape(SimpleNamespace(det_head=head))
        tape.S = (Sh, Sw)
        return tape, tuple(tape._head_forward(_nhwc(x.detach()), B, Sh, Sw))

    def bwd(tape, douts, sink):
        dfused, _ = tape._head_backward(douts, sink, None)
        return [E.to_nchw(dfused, B, x.shape[1], Sh, Sw)]

    outs = _ModuleTrainFn.apply(fwd, bwd, 1, x, *[p for p in head.parameters() if p.requires_grad])
    return dict(zip(E.HEAD_BRANCHES, outs))


# ---- loss with gradient ------------------------------------------------------------------------------------------------------------

class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, tgt_keys, *tensors):
        preds = dict(zip(E.HEAD_BRANCHES, tensors[:5]))
        tgt = dict(zip(tgt_keys, tensors[5:]))
        vals = L.centernet_loss(preds, tgt, weights)
        ctx.preds, ctx.tgt, ctx.weights = preds, tgt, weights
        return tuple(vals[i].clone() for i in range(6))

    @staticmethod
    def backward(ctx, g_total, *g_rest):
        for g in g_rest:
            if g is not None and bool((g != 0).any()):
                raise NotImplementedError("backward through the individual loss terms is not built; use total_loss")
        preds, tgt, w = ctx.preds, ctx.tgt, ctx.weights
        heat = preds["heatmap"]
        B, Cn, H, W = heat.shape
        K = tgt["ind"].shape[1]
        dev = heat.device
        d = L.LossDesc()
        keep = []

        def f32(t):
            t = t.detach().float().contiguous()
            keep.append(t)
            return t.data_ptr()
        d.pred_heatmap, d.tgt_heatmap = f32(heat), f32(tgt["heatmap"])
        for q, name in enumerate(("offset", "size", "rot", "vel")):
            d.pred_reg[q], d.tgt_reg[q] = f32(preds[name]), f32(tgt["target_" + name])
        ind = tgt["ind"].to(torch.int64).contiguous()
        rm = tgt["reg_mask"].to(torch.uint8).contiguous()
        d.ind, d.reg_mask = ind.data_ptr(), rm.data_ptr()
        d.B, d.C, d.H, d.W, d.K = B, Cn, H, W, K
        for i in range(5):
            d.weights[i] = float(w[i])
        dp = [torch.zeros_like(preds[n], dtype=torch.float32) for n in E.HEAD_BRANCHES]
        arr = (C.c_void_p * 5)(*[t.data_ptr() for t in dp])
        scratch = torch.empty(4, device=dev)
        _ck(_lib().bevf_centernet_loss_bwd_f32(C.byref(d), arr, scratch.data_ptr(), _st()), "bevf_centernet_loss_bwd_f32")
        gt = g_total if g_total is not None else torch.zeros((), device=dev)
        return (None, None, *[t * gt for t in dp], *([None] * len(ctx.tgt)))


def loss_with_grad(predictions: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor], weights) -> Dict[str, torch.Tensor]:
    keys = tuple(targets.keys())
    vals = _LossFn.apply(tuple(weights), keys, *[predictions[n] for n in E.HEAD_BRANCHES], *[targets[k] for k in keys])
    names = ("total_loss", "heatmap_loss", "offset_loss", "size_loss", "rot_loss", "vel_loss")
    return dict(zip(names, vals))


# ---- optimiser pieces on device ------------------------------------------------------------------------------------------------------

def clip_grad_norm_(parameters, max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ (L2) with the norm and the scaling on device (ref src/train_detect.py:431)."""
    params = [p for p in parameters if p.grad is not None]
    dev = params[0].grad.device
    flat = torch.cat([p.grad.detach().reshape(-1).float() for p in params])
    work = torch.empty(512, dtype=torch.float64, device=dev)
    out = torch.empty(2, device=dev)
    _ck(_lib().bevf_grad_norm_f32(flat.data_ptr(), flat.numel(), work.data_ptr(), float(max_norm), out.data_ptr(), _st()),
        "bevf_grad_norm_f32")
    for p in params:
        p.grad.mul_(out[1])
    return out[0]


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (ref src/train_detect.py:725-741: lr 1e-4, weight_decay 0.01) on bevf_adamw_step_f32.

    The parameters that receive gradients are moved into ONE flat fp32 arena on the first step (each Parameter becomes
    a view of it, so state_dict / load_state_dict keep working); a step is then one gradient gather, optionally the
    global-norm clip of ref src/train_detect.py:431 (`max_grad_norm`, folded into the update as a device-side
    factor), and one AdamW launch over the whole arena instead of one per tensor."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm: Optional[torch.Tensor] = None
        self._arenas = None

    def _build(self, chosen=None):
        """Arena per group over the parameters that receive gradients (or `chosen`: per-group lists, when a saved
        state is loaded before the first step)."""
        self._arenas = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None] if chosen is None else chosen[gi]
            if not ps:
                self._arenas.append(None)
                continue
            dev = ps[0].device
            if dev.type != "cuda" or any(p.dtype != torch.float32 or p.device != dev for p in ps):
                raise RuntimeError("FusedAdamW: parameters must be fp32 tensors on one cuda device")
            n = sum(p.numel() for p in ps)
            flat = torch.empty(n, device=dev)
            off = 0
            for p in ps:
                k = p.numel()
                flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + k].view(p.shape)
                off += k
            self._arenas.append(dict(params=ps, flat=flat, m=torch.zeros(n, device=dev), v=torch.zeros(n, device=dev),
                                     work=torch.empty(512, dtype=torch.float64, device=dev),
                                     clip=torch.empty(2, device=dev), step=0))

    @torch.no_grad()
    def step(self, closure=None):
        if self._arenas is None:
            self._build()
        for group, ar in zip(self.param_groups, self._arenas):
            with_grad = [p for p in group["params"] if p.grad is not None]
            if ar is None:
                if with_grad:
                    raise RuntimeError("FusedAdamW: the set of parameters with gradients changed; build a new optimiser")
                continue
            if len(with_grad) != len(ar["params"]) or any(a is not b for a, b in zip(with_grad, ar["params"])):
                raise RuntimeError("FusedAdamW: the set of parameters with gradients changed; build a new optimiser")
            g = torch.cat([p.grad.reshape(-1) for p in with_grad])
            clip = None
            if self.max_grad_norm is not None:
                _ck(_lib().bevf_grad_norm_f32(g.data_ptr(), g.numel(), ar["work"].data_ptr(), float(self.max_grad_norm),
                                              ar["clip"].data_ptr(), _st()), "bevf_grad_norm_f32")
                clip = ar["clip"].data_ptr()
                self.last_grad_norm = ar["clip"][0]
            ar["step"] += 1
            b1, b2 = group["betas"]
            _ck(_lib().bevf_adamw_step_f32(ar["flat"].data_ptr(), g.data_ptr(), ar["m"].data_ptr(), ar["v"].data_ptr(), clip,
                                           g.numel(), group["lr"], b1, b2, group["eps"], group["weight_decay"], ar["step"],
                                           _st()), "bevf_adamw_step_f32")
            for p in with_grad:                      # the kernel wrote through raw pointers: tell torch (and the
                torch.autograd.graph.increment_version(p)   # engines' repack signature) that the values changed

    # ---- torch.optim.AdamW's state layout, both ways (checkpoint compatibility) ---------------------------
    def state_dict(self):
        """Same structure as torch.optim.AdamW.state_dict(): per-parameter `step` / `exp_avg` / `exp_avg_sq`, indexed by
        the parameter's position; the file loads into either optimiser."""
        index, groups, k = {}, [], 0
        for group in self.param_groups:
            ids = []
            for p in group["params"]:
                index[id(p)] = k
                ids.append(k)
                k += 1
            g = {key: val for key, val in group.items() if key != "params"}
            g.update(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     decoupled_weight_decay=True)
            g["params"] = ids
            groups.append(g)
        state = {}
        for ar in self._arenas or []:
            if ar is None:
                continue
            off = 0
            for p in ar["params"]:
                n = p.numel()
                state[index[id(p)]] = {"step": torch.tensor(float(ar["step"])),
                                       "exp_avg": ar["m"][off:off + n].view(p.shape).clone(),
                                       "exp_avg_sq": ar["v"][off:off + n].view(p.shape).clone()}
                off += n
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        if len(groups) != len(self.param_groups) or any(len(a["params"]) != len(b["params"])
                                                       for a, b in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict does not match the optimiser's parameter groups")
        chosen, entries = [], []
        for saved, group in zip(groups, self.param_groups):
            for key in ("lr", "betas", "eps", "weight_decay"):
                if key in saved:
                    group[key] = tuple(saved[key]) if key == "betas" else saved[key]
            have = [(p, state_dict["state"][i]) for i, p in zip(saved["params"], group["params"]) if i in state_dict["state"]]
            chosen.append([p for p, _ in have])
            entries.append([e for _, e in have])
        with torch.no_grad():
            self._build(chosen)
            for ar, ent in zip(self._arenas, entries):
                if ar is None:
                    continue
                off, steps = 0, set()
                for p, e in zip(ar["params"], ent):
                    n = p.numel()
                    ar["m"][off:off + n].copy_(e["exp_avg"].reshape(-1))
                    ar["v"][off:off + n].copy_(e["exp_avg_sq"].reshape(-1))
                    steps.add(int(float(e["step"])))
                    off += n
                if len(steps) > 1:
                    raise ValueError(f"FusedAdamW keeps one step count per group; the loaded state has {sorted(steps)}")
                ar["step"] = steps.pop() if steps else 0
