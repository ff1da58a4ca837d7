"""Eval-mode executors for the two backbones: same parameters, same arithmetic, but laid out for
MI355X (everything channel-last, no layout copies) and with the obvious fusions the module graphs
miss.  They read the parameters of the standard modules (`transformers.SegformerModel`, the
ResNet / decoder modules of `deeplab.py`), so state_dicts and training are untouched.

What changes relative to the module graphs (profiles/r01_bench_step_kernels_v1.csv):
  * eval BatchNorm is folded into the preceding convolution's weights (cached per parameter
    version) and `+shift [+identity] -> ReLU` is ONE in-place HIP pass (awseg_bias_act_nhwc)
    instead of BN, add and clamp kernels;
  * fp32 depthwise 3x3 convolutions (MiT Mix-FFN, smp SeparableConv2d) run on
    awseg_dwconv3x3_nhwc with bias + GELU fused, instead of MIOpen's naive reference kernel;
  * MiT tokens stay [B,H,W,C]: the token<->NCHW transposes of the HF forward disappear
    (a channels_last NCHW view of the same memory feeds the strided convolutions).
"""
from __future__ import annotations

import contextlib
import os
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _native as N
from .. import ops

CL = torch.channels_last


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# --------------------------------------------------------------------------- parameter caches
def _versions(*tensors):
    return tuple((t.data_ptr(), t._version) for t in tensors if t is not None)


def folded_conv_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    """(weight * bn_scale in channels_last, shift) for eval-mode Conv -> BN; cached on the modules
    and recomputed whenever a parameter / running stat changes (in-place version counters)."""
    key = _versions(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    cache = getattr(conv, "_awseg_fold", None)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2]
    inv = torch.rsqrt(bn.running_var + bn.eps)
    scale = bn.weight * inv
    shift = bn.bias - bn.running_mean * scale
    if conv.bias is not None:
        shift = shift + conv.bias * scale
    w = (conv.weight * scale.view(-1, 1, 1, 1)).contiguous(memory_format=CL)
    conv._awseg_fold = (key, w, shift.contiguous())
    # everything derived from the folded weights dies with them: the split-operand image is keyed on the folded tensor's
    # (address, version), and a re-folded tensor can land on the same address with the same version
    for dep in ("_awseg_wsplit", "_awseg_wbf16"):
        if hasattr(conv, dep):
            delattr(conv, dep)
    return w, shift.contiguous()


def cached(module: nn.Module, name: str, tensors, fn):
    key = _versions(*tensors)
    cache = getattr(module, "_awseg_" + name, None)
    if cache is not None and cache[0] == key:
        return cache[1]
    val = fn()
    setattr(module, "_awseg_" + name, (key, val))
    if name not in ("wsplit", "wbf16"):
        for dep in ("_awseg_wsplit", "_awseg_wbf16"):
            if hasattr(module, dep):
                delattr(module, dep)              # operand images are keyed on the tensor just replaced
    return val


def split_weights(owner: nn.Module, w2: torch.Tensor, m: int):
    """The GEMM's prepared weight image cached on `owner`: gemm_bf16_weights(w2) in bf16 mode, gemm_split_weights(w2) for
    the split-operand kernel, None when the problem stays on the library GEMM."""
    if ops.gemm_wants_bf16(m, w2.shape[0], w2.shape[1]):
        def build():
            wb = ops.gemm_bf16_weights(w2)
            wb._awseg_bf16 = True
            return wb
        return cached(owner, "wbf16", (w2,), build)
    if not ops.gemm_wants_split(m, w2.shape[0], w2.shape[1]):
        return None
    return cached(owner, "wsplit", (w2,), lambda: ops.gemm_split_weights(w2))


def dw_taps(conv: nn.Conv2d) -> torch.Tensor:
    """[C,1,3,3] depthwise weight -> [9, C] tap-major."""
    return cached(conv, "w9", [conv.weight], lambda: conv.weight.view(conv.weight.shape[0], 9).t().contiguous())


def nhwc_view(t: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] channels_last tensor -> contiguous [B,H,W,C] view of the same memory."""
    if not t.is_contiguous(memory_format=CL):
        t = t.contiguous(memory_format=CL)
    v = t.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


WINOGRAD_MIN_CIN = 64      # 64->64 @1/4 (ResNet layer1): 0.585 ms vs 0.64-0.69 ms for MIOpen's direct kernel (tools/kernel_bench.py)


def _is_winograd(conv: nn.Conv2d) -> bool:
    return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.groups == 1 and conv.padding == conv.dilation
            and conv.dilation[0] == conv.dilation[1] and conv.in_channels % 16 == 0 and conv.out_channels % 64 == 0
            and conv.in_channels >= WINOGRAD_MIN_CIN)


def winograd_conv_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    """(U [16,Cin,Cout] = G (w * bn_scale) G^T, shift) for eval-mode Conv3x3 -> BN; cached like folded_conv_bn."""
    def build():
        inv = torch.rsqrt(bn.running_var + bn.eps)
        scale = bn.weight * inv
        shift = bn.bias - bn.running_mean * scale
        if conv.bias is not None:
            shift = shift + conv.bias * scale
        return ops.winograd_weights(conv.weight, scale), shift.contiguous()
    return cached(conv, "wino", [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var], build)


def winograd_split_conv_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    """(split-operand image of U = G (w * bn_scale) G^T, shift) for the f16-MFMA Winograd kernel; cached like winograd_conv_bn."""
    def build():
        inv = torch.rsqrt(bn.running_var + bn.eps)
        scale = bn.weight * inv
        shift = bn.bias - bn.running_mean * scale
        if conv.bias is not None:
            shift = shift + conv.bias * scale
        return ops.winograd_split_weights(conv.weight, scale), shift.contiguous()
    return cached(conv, "wino_split", [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var], build)


def conv3x3_winograd_bn(x_nhwc: torch.Tensor, conv: nn.Conv2d, bn: nn.BatchNorm2d, act: int = 0, residual: torch.Tensor = None,
                        w2: torch.Tensor = None, b2: torch.Tensor = None) -> torch.Tensor:
    """act(bn(conv3x3(x)) [+ residual]) (or the fused 1x1 + sigmoid head) on a contiguous NHWC tensor: the split-operand
    f16-MFMA Winograd kernel, or the float32-input MFMA one (ops.WINO_SPLIT)."""
    if ops.PRECISION == "bf16":
        def build():
            inv = torch.rsqrt(bn.running_var + bn.eps)
            scale = bn.weight * inv
            shift = bn.bias - bn.running_mean * scale
            if conv.bias is not None:
                shift = shift + conv.bias * scale
            return ops.winograd_bf16_weights(conv.weight, scale), shift.contiguous()
        ub, shift = cached(conv, "wino_bf16", [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var], build)
        return ops.conv3x3_winograd_bf16(x_nhwc, ub, conv.out_channels, shift, act=act, dilation=conv.dilation[0], residual=residual, w2=w2, b2=b2)
    if ops.WINO_SPLIT:
        us, shift = winograd_split_conv_bn(conv, bn)
        return ops.conv3x3_winograd_split(x_nhwc, us, conv.out_channels, shift, act=act, dilation=conv.dilation[0], residual=residual, w2=w2, b2=b2)
    u, shift = winograd_conv_bn(conv, bn)
    return ops.conv3x3_winograd(x_nhwc, u, shift, act=act, dilation=conv.dilation[0], residual=residual, w2=w2, b2=b2)


def _is_patch_gemm(conv: nn.Conv2d) -> bool:
    """Strided / patch convolutions that run as im2col + ONE GEMM (deterministic summation order; MIOpen's default for
    these shapes is a split-K kernel that accumulates with atomics): stride-2 3x3 (ResNet layer2/3 first blocks, MiT
    patch embeddings 2-4) and kernel == stride (MiT sequence reduction).  The two 7x7 stems on 3 input channels stay on
    MIOpen (C % 4 != 0; their solver is a plain implicit GEMM)."""
    return (conv.groups == 1 and conv.in_channels % 4 == 0 and conv.stride[0] == conv.stride[1] and conv.stride[0] > 1
            and conv.kernel_size[0] > 1 and conv.dilation == (1, 1) and conv.padding[0] == conv.padding[1])


def patch_weights(conv: nn.Conv2d, scale: torch.Tensor = None) -> torch.Tensor:
    """[N,C,kh,kw] (times an optional per-N scale) -> [N, round8(kh*kw*C)] in im2col column order (ky, kx, c)."""
    w = conv.weight if scale is None else conv.weight * scale.view(-1, 1, 1, 1)
    n = w.shape[0]
    w2 = w.permute(0, 2, 3, 1).reshape(n, -1)
    k = w2.shape[1]
    kp = (k + 7) // 8 * 8
    if kp != k:
        w2 = torch.nn.functional.pad(w2, (0, kp - k))
    return w2.contiguous()


def conv_gemm_nhwc(x_nhwc: torch.Tensor, conv: nn.Conv2d, w2: torch.Tensor, bias: torch.Tensor, act: int, owner=None) -> torch.Tensor:
    """conv(x) as im2col + GEMM on a contiguous [B,H,W,C] tensor -> [B,Ho,Wo,N]; w2 = patch_weights(conv[, scale])."""
    kh, kw, st, pd = conv.kernel_size[0], conv.kernel_size[1], conv.stride[0], conv.padding[0]
    b, h, w, c = x_nhwc.shape
    if ops.CONV_GATHER and c % 32 == 0 and w2.shape[1] == kh * kw * c and x_nhwc.dtype == torch.float32:
        # the A operand gathered inside the GEMM: no im2col matrix (written once, read once: 0.9 ms per step at 8 x 1024 x 2048)
        m = b * ((h + 2 * pd - kh) // st + 1) * ((w + 2 * pd - kw) // st + 1)
        own = owner if owner is not None else conv
        ws = split_weights(own, w2, m)
        if ws is None and ops.SMALL_CONV_SPLIT and ops.PRECISION != "bf16" and ops.GEMM_SPLIT and m >= 128 and w2.shape[0] >= 8:
            # MiT's sequence-reduction convolutions at stages 1 / 2 (16 384 output tokens x 32 / 64 channels): too few rows for the
            # dispatcher's rule, but the gathered form still beats an im2col pass + a library GEMM (two launches, the matrix written
            # and read back)
            ws = cached(own, "wsplit_small", (w2,), lambda: ops.gemm_split_weights(w2))
        if ws is not None and not getattr(ws, "_awseg_bf16", False):
            return ops.conv_gemm_split(x_nhwc, ws, bias, act, kh, kw, st, pd)
    cols, ho, wo = ops.im2col_nhwc(x_nhwc, kh, kw, st, pd, 1, w2.shape[1])
    y = ops.gemm_bias_act(cols, w2, bias, act, w_split=split_weights(owner if owner is not None else conv, w2, cols.shape[0]))
    return y.view(x_nhwc.shape[0], ho, wo, w2.shape[0])


def _is_pointwise(conv: nn.Conv2d) -> bool:
    # (dilation is irrelevant for a 1x1 kernel: smp's make_dilated sets it on every conv of the stage)
    return conv.kernel_size == (1, 1) and conv.stride in ((1, 1), (2, 2)) and conv.padding == (0, 0) and conv.groups == 1


def conv_bn_act(x: torch.Tensor, conv: nn.Conv2d, bn: nn.BatchNorm2d, act: int, residual: torch.Tensor = None,
                residual_is_scratch: bool = False) -> torch.Tensor:
    """act(bn(conv(x)) [+ residual]) with x / result logically NCHW in channels_last memory.

    A 1x1 stride-1 convolution of an NHWC tensor IS a row-major GEMM [B*H*W, Cin] x [Cin, Cout], so
    it is ONE hipBLASLt call (awseg_gemm_bias_act) whose epilogue does `+shift`, the residual (the
    GEMM's beta*C operand, written over the residual's buffer when the caller says it is dead
    afterwards: `residual_is_scratch`) and the ReLU.  3x3 stride-1 convolutions go to the Winograd /
    MFMA kernel; the rest (7x7, strided) stays on MIOpen with the one-pass HIP epilogue."""
    w, shift = folded_conv_bn(conv, bn)
    if _is_pointwise(conv) and x.is_contiguous(memory_format=CL):
        if conv.stride == (2, 2):
            # a strided 1x1 (the ResNet downsample branches) reads every other pixel: gathered inside the GEMM when the split
            # kernel takes the shape, else one small copy (a quarter of x) and it is the same GEMM
            Bq, Cq, Hq, Wq = x.shape
            mq = Bq * ((Hq - 1) // 2 + 1) * ((Wq - 1) // 2 + 1)
            wq = w.view(w.shape[0], Cq)
            wsq = split_weights(conv, wq, mq) if (ops.CONV_GATHER and Cq % 32 == 0 and residual is None and act in (N.ACT_RELU, N.ACT_NONE)) else None
            if wsq is not None and not getattr(wsq, "_awseg_bf16", False):
                return ops.conv_gemm_split(nhwc_view(x), wsq, shift, act, 1, 1, 2, 0).permute(0, 3, 1, 2)
            x = x[:, :, ::2, ::2].contiguous(memory_format=CL)
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        x2 = x.permute(0, 2, 3, 1).reshape(B * H * W, Cin)
        w2 = w.view(Cout, Cin)
        if act in (N.ACT_RELU, N.ACT_NONE):
            r2 = None if residual is None else nhwc_view(residual).reshape(B * H * W, Cout)
            y2 = ops.gemm_bias_act(x2, w2, shift, act, residual=r2, out=r2 if (r2 is not None and residual_is_scratch) else None,
                                   w_split=split_weights(conv, w2, B * H * W))
        else:
            y2 = torch.addmm(shift, x2, w2.t()) if residual is None else torch.addmm(nhwc_view(residual).reshape(B * H * W, Cout), x2, w2.t())
            ops.bias_act_nhwc_(y2, torch.zeros_like(shift) if residual is None else shift, None, act)
        return y2.view(B, H, W, Cout).permute(0, 3, 1, 2)
    if _is_patch_gemm(conv) and residual is None and act in (N.ACT_NONE, N.ACT_RELU):
        def build():
            inv = torch.rsqrt(bn.running_var + bn.eps)
            scale = bn.weight * inv
            sh = bn.bias - bn.running_mean * scale
            if conv.bias is not None:
                sh = sh + conv.bias * scale
            return patch_weights(conv, scale), sh.contiguous()
        w2, sh = cached(conv, "patch", [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var], build)
        return conv_gemm_nhwc(nhwc_view(x), conv, w2, sh, act).permute(0, 3, 1, 2)
    if _is_winograd(conv) and act in (N.ACT_NONE, N.ACT_RELU):
        # 3x3 stride-1 "same" convolution: Winograd F(2x2,3x3) on the fp32 matrix cores, epilogue fused
        y = conv3x3_winograd_bn(nhwc_view(x), conv, bn, act, None if residual is None else nhwc_view(residual))
        return y.permute(0, 3, 1, 2)
    y = F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    if not y.is_contiguous(memory_format=CL):
        y = y.contiguous(memory_format=CL)
    ops.bias_act_nhwc_(y.permute(0, 2, 3, 1), shift, None if residual is None else nhwc_view(residual), act)
    return y


STEM_ROWS = os.environ.get("AWSEG_STEM_ROWS", "1") != "0"
_stem_pad = {}
_stem_scope = None


@contextlib.contextmanager
def stem_scope():
    """Inside one ensemble forward both members read the same input: the zero-padded 4-channel image of `_stem_rows` is built by
    the first stem and reused by the second (keyed on the input's storage; the scope ends with the forward, so a buffer that is
    refilled in place for the next batch is padded again)."""
    global _stem_scope
    prev, _stem_scope = _stem_scope, {}
    try:
        yield
    finally:
        _stem_scope = prev


def stem_image_ok(x: torch.Tensor) -> bool:
    return STEM_ROWS and ops.GEMM_SPLIT and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] <= 4 and ops.PRECISION != "bf16"


def stem_image(x: torch.Tensor) -> torch.Tensor:
    """The zero-padded 4-channel NHWC copy [B, H, W + pads, 4] of the frames that both 7x7 stems read (3 zero columns on the left, enough on
    the right for the last run of 8 pixels at either stride); built once per forward inside a stem_scope, the buffer kept across forwards."""
    B, C, H, W = x.shape
    wp_any = max(W + 3, ((W + 6 - 7) // 2) * 2 + 8, ((W + 6 - 7) // 4) * 4 + 8)   # one image for both strides
    skey = (x.data_ptr(), tuple(x.shape), tuple(x.stride()))
    xp = _stem_scope.get(skey) if _stem_scope is not None else None
    if xp is None:
        # W and C are part of the key: wp_any is the same for W = 2k-1 and 2k, and a buffer filled for the wider (or deeper) input
        # would keep that input's last column (4th channel) where this one's zero padding belongs
        key = (str(x.device), B, H, W, C, wp_any)
        xp = _stem_pad.get(key)
        if xp is None:
            _stem_pad.clear()                                  # one shape at a time: 270 MB at 8 x 1024 x 2048
            xp = _stem_pad[key] = torch.zeros(B, H, wp_any, 4, dtype=torch.float32, device=x.device)
        if x.stride(3) == 1 and x.stride(2) >= W and min(x.stride()) >= 0:
            ops.stem_image_fill(x, xp)                     # the padding columns / channel stay zero
        else:
            xp[:, :, 3:3 + W, :C] = x.permute(0, 2, 3, 1)
        if _stem_scope is not None:
            _stem_scope[skey] = xp
    return xp


def _stem_rows(x: torch.Tensor, conv: nn.Conv2d, w: torch.Tensor, bias: torch.Tensor = None):
    """A 7x7 stem on <= 4 input channels (ResNet conv1: stride 2; MiT's first patch embedding: stride 4, 32 channels) as one
    split-operand GEMM whose A operand is gathered row by row from a zero-padded [B, H, W + pads, 4] copy of the image
    (ops.conv_rows_gemm_split) instead of MIOpen's float32 implicit GEMM (+ its separate bias kernel).  Returns the NHWC result
    [B, Ho, Wo, N], or None when the shape is not a stem's / the kernel does not take it."""
    st = conv.stride[0]
    if not (STEM_ROWS and ops.GEMM_SPLIT and x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (7, 7) and conv.stride in ((2, 2), (4, 4))
            and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels <= 4
            and (conv.out_channels % 64 == 0 or 8 <= conv.out_channels < 64) and ops.PRECISION != "bf16"):
        return None
    B, C, H, W = x.shape
    wo = (W + 6 - 7) // st + 1
    wp = max(W + 3, (wo - 1) * st + 8)                     # 3 zero columns on the left; the last run of 8 pixels ends inside the row
    xp = stem_image(x)
    assert wp <= xp.shape[2]
    ws = cached(conv, "stemrows", [w], lambda: ops.gemm_split_weights(ops.stem_rows_weights(w)))
    return ops.conv_rows_gemm_split(xp, ws, bias, N.ACT_NONE, 7, st, 3, wo)


# --------------------------------------------------------------------------- ResNet encoder
def _bottleneck_tail_dual(out: torch.Tensor, y: torch.Tensor, blk):
    """relu(bn3(conv3(out)) + bn_d(downsample(y))) for a bottleneck whose two branches end in 1x1 convolutions, as one split-operand
    GEMM over [out | y] (ops.gemm_split_dual); None where that form does not apply (bf16 mode, other layouts, shapes the kernel
    declines) — the caller then runs the branches one after the other."""
    c3, cd, bn3, bnd = blk.conv3, blk.downsample[0], blk.bn3, blk.downsample[1]
    if not (ops.DUAL_TAIL and ops.GEMM_SPLIT and ops.PRECISION != "bf16" and out.is_cuda and out.dtype == torch.float32
            and c3.kernel_size == (1, 1) and cd.kernel_size == (1, 1) and c3.stride == (1, 1) and c3.groups == 1 and cd.groups == 1
            and cd.stride[0] == cd.stride[1] and cd.padding == (0, 0) and c3.padding == (0, 0)
            and c3.in_channels % 32 == 0 and cd.in_channels % 32 == 0 and isinstance(bnd, nn.BatchNorm2d)):
        return None
    ol, yl = nhwc_view(out), nhwc_view(y)
    B, Ho, Wo, k1 = ol.shape
    _, H, W, k2 = yl.shape
    st = cd.stride[0]
    if (H - 1) // st + 1 != Ho or (W - 1) // st + 1 != Wo:
        return None
    n = c3.out_channels

    def build():
        w3, b3 = folded_conv_bn(c3, bn3)
        wd, bd = folded_conv_bn(cd, bnd)
        return ops.gemm_split_weights(torch.cat([w3.reshape(n, k1), wd.reshape(n, k2)], dim=1).contiguous()), (b3 + bd).contiguous()
    ws, bias = cached(blk, "dualtail", [c3.weight, bn3.weight, bn3.bias, bn3.running_mean, bn3.running_var,
                                        cd.weight, bnd.weight, bnd.bias, bnd.running_mean, bnd.running_var], build)
    y2 = ops.gemm_split_dual(ol.view(B * Ho * Wo, k1), yl if st > 1 else yl.view(B * H * W, k2), ws, bias, N.ACT_RELU, stride=st if st > 1 else 0)
    return None if y2 is None else y2.view(B, Ho, Wo, n).permute(0, 3, 1, 2)


@torch.no_grad()
def resnet_features(enc, x: torch.Tensor, stem_feature: bool = False) -> List[torch.Tensor]:
    """smp feature list [x, stem, layer1..layer4] of `deeplab.ResNetEncoder`, fused eval execution.
    The DeepLabV3+ decoder never reads the stride-2 stem feature, so by default it is not produced (None in the
    list) and the stem's `+shift -> ReLU` runs AFTER the max-pool, on a quarter of the pixels: both are monotone
    per channel, so maxpool(relu(x + b)) == relu(maxpool(x) + b) bit for bit."""
    feats = [x]
    if stem_feature:
        x = x.contiguous(memory_format=CL)
        feats = [x]
        y = conv_bn_act(x, enc.conv1, enc.bn1, N.ACT_RELU)
        feats.append(y)
        y = enc.maxpool(y)
    else:
        w, shift = folded_conv_bn(enc.conv1, enc.bn1)
        # the stem reads the input as it comes (NCHW from the weather kernels): ONE strided copy into the zero-padded image both
        # members share — no channels_last copy of the frames per member (two of them plus a second padded fill were 0.23 ms a step)
        y = _stem_rows(x, enc.conv1, w)
        y = y.permute(0, 3, 1, 2) if y is not None else F.conv2d(x.contiguous(memory_format=CL), w, None, enc.conv1.stride, enc.conv1.padding)
        mp = enc.maxpool
        if (y.is_cuda and y.dtype == torch.float32 and y.is_contiguous(memory_format=CL) and y.shape[1] % 4 == 0
                and _pair(mp.kernel_size) == (3, 3) and _pair(mp.stride) == (2, 2) and _pair(mp.padding) == (1, 1)
                and _pair(mp.dilation) == (1, 1) and not mp.ceil_mode):
            # HIP: no int64 index tensor (torch writes 537 MB of them); the stem's shift + ReLU ride in the pooling kernel's store
            y = ops.maxpool3x3s2_nhwc(nhwc_view(y), shift=shift).permute(0, 3, 1, 2)
        else:
            y = mp(y)
            if not y.is_contiguous(memory_format=CL):
                y = y.contiguous(memory_format=CL)
            ops.bias_act_nhwc_(y.permute(0, 2, 3, 1), shift, None, N.ACT_RELU)
        feats.append(None)
    for layer in (enc.layer1, enc.layer2, enc.layer3, enc.layer4):
        for blk in layer:
            out = conv_bn_act(y, blk.conv1, blk.bn1, N.ACT_RELU)
            out = conv_bn_act(out, blk.conv2, blk.bn2, N.ACT_RELU)
            if blk.downsample is not None:
                # first block of a stage: relu(bn3(conv3(out)) + bn_d(downsample(y))) is ONE product over [out | y at the stride] — the
                # downsample branch's map (as wide as the block's output) is neither written nor read back as a residual
                y2 = _bottleneck_tail_dual(out, y, blk)
                if y2 is not None:
                    y = y2
                    continue
            idn = y if blk.downsample is None else conv_bn_act(y, blk.downsample[0], blk.downsample[1], N.ACT_NONE)
            # the identity is dead after this block unless it is a tensor handed out in `feats`
            scratch = not any(idn is f for f in feats)
            y = conv_bn_act(out, blk.conv3, blk.bn3, N.ACT_RELU, residual=idn, residual_is_scratch=scratch)
        feats.append(y)
    return feats


@torch.no_grad()
def separable_bn_relu(x: torch.Tensor, sep, bn: nn.BatchNorm2d) -> torch.Tensor:
    """smp SeparableConv2d(3x3 depthwise, 1x1 pointwise) -> BN -> ReLU on a channels_last tensor."""
    dwc, pwc = sep[0], sep[1]
    xl = nhwc_view(x)
    B, H, W, C = xl.shape
    d = ops.dwconv3x3_nhwc(xl, dw_taps(dwc), None, N.ACT_NONE, dilation=dwc.dilation[0])
    w, shift = folded_conv_bn(pwc, bn)                                   # [Cout,Cin,1,1]
    w2 = w.view(w.shape[0], C)
    y = ops.gemm_bias_act(d.view(B * H * W, C), w2, shift, N.ACT_RELU, w_split=split_weights(pwc, w2, B * H * W))
    return y.view(B, H, W, -1).permute(0, 3, 1, 2)                       # NCHW view, channels_last memory


# --------------------------------------------------------------------------- MiT encoder
def _ln(t, ln: nn.LayerNorm):
    c = t.shape[-1]
    if c % 4 == 0 and c <= 1024 and ln.weight is not None and ln.bias is not None:
        return ops.layernorm_rows(t, ln.weight, ln.bias, ln.eps)           # sub-wave rows (HIP)
    return F.layer_norm(t, (c,), ln.weight, ln.bias, ln.eps)


LINEAR_SPLIT = os.environ.get("AWSEG_LINEAR_SPLIT", "1") != "0"


def _linear(x: torch.Tensor, lin: nn.Linear) -> torch.Tensor:
    """lin(x) on the last dimension: this repo's GEMM kernels where they take the shape (split-operand f16 MFMA; bf16 MFMA in bf16
    mode), torch's float32 GEMM otherwise.  AWSEG_LINEAR_SPLIT=0: the q / k / v / fc1 projections stay on torch's GEMM as in round 2."""
    k = x.shape[-1]
    m = x.numel() // k
    if lin.bias is not None and x.is_contiguous() and (ops.gemm_wants_bf16(m, lin.out_features, k) or
                                                       (LINEAR_SPLIT and ops.gemm_wants_split(m, lin.out_features, k))):
        y = ops.gemm_bias_act(x.view(m, k), lin.weight, lin.bias, N.ACT_NONE, w_split=split_weights(lin, lin.weight, m))
        return y.view(*x.shape[:-1], lin.out_features)
    return F.linear(x, lin.weight, lin.bias)


def _linear_residual(x2: torch.Tensor, lin: nn.Linear, tok: torch.Tensor) -> torch.Tensor:
    """tok + lin(x2) for NHWC tokens [B,H,W,C], written over tok (which the caller owns and drops)."""
    if lin.bias is None or not tok.is_contiguous():
        return tok + F.linear(x2, lin.weight, lin.bias).view_as(tok)
    t2 = tok.view(-1, tok.shape[-1])
    ops.gemm_bias_act(x2, lin.weight, lin.bias, N.ACT_NONE, residual=t2, out=t2, w_split=split_weights(lin, lin.weight, x2.shape[0]))
    return tok


def _kv_packed(kv: torch.Tensor, a) -> torch.Tensor:
    """[key | value] projections of the (reduced) tokens kv [...,C] -> [...,2C]: one GEMM with k_proj's and v_proj's weights stacked.
    The handful of reduced tokens (16 384 rows at the bench shape) is below the sizes gemm_wants_split() sends to this repo's kernel
    on its own; it goes there all the same — one launch of ~10 us either way, and the step keeps to one GEMM implementation."""
    kp, vp = a.k_proj, a.v_proj
    wkv, bkv = cached(a, "kvpack", [kp.weight, kp.bias, vp.weight, vp.bias],
                      lambda: (torch.cat([kp.weight, vp.weight]).contiguous(), torch.cat([kp.bias, vp.bias]).contiguous()))
    c = kv.shape[-1]
    m = kv.numel() // c
    if ops.gemm_wants_bf16(m, 2 * c, c) or ops.gemm_wants_split(m, 2 * c, c):
        y = ops.gemm_bias_act(kv.view(m, c), wkv, bkv, N.ACT_NONE, w_split=split_weights(a, wkv, m))
    else:
        ws = cached(a, "kvsplit", (wkv,), lambda: ops.gemm_split_weights(wkv))
        y = ops.gemm_bias_act(kv.view(m, c), wkv, bkv, N.ACT_NONE, w_split=ws, split=True)
    return y.view(*kv.shape[:-1], 2 * c)


@torch.no_grad()
def mit_features_nhwc(seg, x: torch.Tensor) -> torch.Tensor:
    """Last-stage hidden state of transformers.SegformerModel as [B,h,w,C] (NHWC).  Same arithmetic
    as the HF forward (modeling_segformer.py SegformerStage / Layer / Attention / MixMLP)."""
    if not hasattr(seg, "stages"):
        # other transformers versions name their sub-modules differently: use their own forward
        return seg(x).last_hidden_state.permute(0, 2, 3, 1).contiguous()
    t = x                                                            # (the first stage's stem takes any layout; later stages: channels_last views)
    tok = None
    for st in seg.stages:
        pe = st.patch_embeddings
        if _is_patch_gemm(pe.proj) and pe.proj.bias is not None:
            w2 = cached(pe.proj, "patch", [pe.proj.weight], lambda: patch_weights(pe.proj))
            y = conv_gemm_nhwc(nhwc_view(t), pe.proj, w2, pe.proj.bias, N.ACT_NONE)              # [B,H,W,C]
        else:
            y = _stem_rows(t, pe.proj, pe.proj.weight, pe.proj.bias) if pe.proj.bias is not None else None
            if y is None:
                w = cached(pe.proj, "wcl", [pe.proj.weight], lambda: pe.proj.weight.contiguous(memory_format=CL))
                y = nhwc_view(F.conv2d(t.contiguous(memory_format=CL), w, pe.proj.bias, pe.proj.stride, pe.proj.padding))
        tok = _ln(y, pe.layer_norm)                                          # [B,H,W,C]
        B, H, W, C = tok.shape
        for blk in st.blocks:
            a = blk.attention
            hcur = _ln(tok, blk.layernorm_before)
            q = _linear(hcur, a.q_proj)
            if a.sequence_reduction_ratio > 1:
                sr = a.sequence_reduction
                src = sr.sequence_reduction
                if _is_patch_gemm(src) and src.bias is not None:
                    w2 = cached(src, "patch", [src.weight], lambda: patch_weights(src))
                    kv = conv_gemm_nhwc(hcur, src, w2, src.bias, N.ACT_NONE)
                else:
                    wsr = cached(src, "wcl", [src.weight], lambda: src.weight.contiguous(memory_format=CL))
                    kv = nhwc_view(F.conv2d(hcur.permute(0, 3, 1, 2), wsr, src.bias, src.stride))
                kv = _ln(kv, sr.layer_norm)
            else:
                kv = hcur
            nh, d = a.num_attention_heads, a.head_dim
            nkv = kv.shape[1] * kv.shape[2]
            if (d == 32 and nkv % 32 == 0 and ops.KV_PACKED and kv.is_cuda and kv.dtype == torch.float32 and C % 8 == 0
                    and a.k_proj.bias is not None and a.v_proj.bias is not None and kv.is_contiguous()):
                # the key and the value projection as ONE GEMM over the stacked weights; the attention kernel (exact-fp32-grade flash
                # attention on the matrix cores, token-major in and out) reads a token's key and value from the two halves of its row
                o = ops.attention_d32_packed_kv(q.view(B, H * W, C), _kv_packed(kv, a).view(B, nkv, 2 * C), nh, a.scaling).view(B * H * W, C)
            elif d == 32 and nkv % 32 == 0:
                k, v = _linear(kv, a.k_proj), _linear(kv, a.v_proj)
                o = ops.attention_d32(q.view(B, H * W, C), k.reshape(B, nkv, C), v.reshape(B, nkv, C), nh, a.scaling).view(B * H * W, C)
            else:
                # other head widths (MiT-B1..B5: 64): torch's fused attention; in bf16 mode on bf16 operands (bf16 MFMA path)
                k, v = _linear(kv, a.k_proj), _linear(kv, a.v_proj)
                qq, kk, vv = (t_.view(B, -1, nh, d).transpose(1, 2) for t_ in (q.view(B, H * W, C), k.reshape(B, -1, C), v.reshape(B, -1, C)))
                if ops.PRECISION == "bf16":
                    o = F.scaled_dot_product_attention(qq.bfloat16(), kk.bfloat16(), vv.bfloat16(), scale=a.scaling).float()
                else:
                    o = F.scaled_dot_product_attention(qq, kk, vv, scale=a.scaling)
                o = o.transpose(1, 2).reshape(B * H * W, C)
            # tok + o_proj(o): the residual is the GEMM's beta*C operand, accumulated over tok's buffer
            tok = _linear_residual(o, a.o_proj, tok)
            m = blk.mlp
            ln2 = blk.layernorm_after
            # (also in bf16 mode — MiT-B5's stage 1 is 64 wide: the fused kernel computes in float32 grade, above what that mode asks for)
            if (ops.MIXFFN_FUSED and tok.is_cuda and tok.dtype == torch.float32 and C in (32, 64)
                    and getattr(seg.config, "hidden_act", "gelu") == "gelu" and m.fc1.bias is not None and m.fc2.bias is not None
                    and m.dwconv.dwconv.bias is not None and ln2.weight is not None and ln2.bias is not None):
                # the whole Mix-FFN as one tile kernel: the 4x-wide hidden map never reaches HBM (four launches and four passes otherwise)
                prep = cached(m, "mixffn", [ln2.weight, ln2.bias, m.fc1.weight, m.fc2.weight],
                              lambda: (ops.mixffn_split_weights(m.fc1.weight), ops.mixffn_split_weights(m.fc2.weight))
                              if ops.mixffn_operands_ok(ln2.weight, ln2.bias, m.fc1.weight, m.fc2.weight) else None)
                fusedtok = None if prep is None else ops.mixffn_fused(
                    tok, ln2.weight, ln2.bias, ln2.eps, m.fc1.weight, m.fc1.bias, dw_taps(m.dwconv.dwconv), m.dwconv.dwconv.bias,
                    m.fc2.weight, m.fc2.bias, w1_split=prep[0], w2_split=prep[1], checked=True)
                if fusedtok is not None:
                    tok = fusedtok
                    continue
            hcur = _linear(_ln(tok, blk.layernorm_after), m.fc1)
            if getattr(seg.config, "hidden_act", "gelu") == "gelu":
                hcur = ops.dwconv3x3_nhwc(hcur, dw_taps(m.dwconv.dwconv), m.dwconv.dwconv.bias, N.ACT_GELU)   # dwconv + GELU fused
            else:
                hcur = m.activation_fn(ops.dwconv3x3_nhwc(hcur, dw_taps(m.dwconv.dwconv), m.dwconv.dwconv.bias, N.ACT_NONE))
            tok = _linear_residual(hcur.reshape(B * H * W, -1), m.fc2, tok)
        tok = _ln(tok, st.layer_norm)
        t = tok.permute(0, 3, 1, 2)                                          # NCHW view (channels_last) for the next stage
    return tok
