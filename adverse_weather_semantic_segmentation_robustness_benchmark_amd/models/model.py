"""Model classes of the hot path, same names / constructor arguments / output dict keys /
state_dict prefixes as the reference's PKG/models/model.py, re-designed for MI355X:

* backbones (MiT encoder, ResNet) run on PyTorch-ROCm (MIOpen / hipBLASLt),
* the SegFormer head never materialises the x32-upsampled feature map (HIP, A8),
* the ASPP's three atrous depthwise convolutions are one HIP pass (A9),
* ensemble combine / temperature / argmax / confusion is one HIP pass (A11-A13),
* the fog-density-aware loss is a HIP forward + backward pair (A15).

Eval-mode forwards on CUDA tensors take the HIP kernels; training-mode forwards keep the
reference's op graph on torch-ROCm so autograd works.  Nothing here runs the HIP parts on the
CPU: CPU tensors raise (see _native.ptr).
"""
from __future__ import annotations

import logging
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _native as N
from .. import ops
from . import fused
from .deeplab import Conv2d, DeepLabV3Plus, _bn_fold

logger = logging.getLogger(__name__)

# MiT-B0..B5 encoder shapes (SegFormer paper, table 6); B0 is what the reference ends up with
# offline (PKG/models/model.py:120-130).
MIT_CONFIGS = {
    "b0": dict(hidden_sizes=[32, 64, 160, 256], depths=[2, 2, 2, 2]),
    "b1": dict(hidden_sizes=[64, 128, 320, 512], depths=[2, 2, 2, 2]),
    "b2": dict(hidden_sizes=[64, 128, 320, 512], depths=[3, 4, 6, 3]),
    "b3": dict(hidden_sizes=[64, 128, 320, 512], depths=[3, 4, 18, 3]),
    "b4": dict(hidden_sizes=[64, 128, 320, 512], depths=[3, 8, 27, 3]),
    "b5": dict(hidden_sizes=[64, 128, 320, 512], depths=[3, 6, 40, 3]),
}


def _use_hip(module: nn.Module, x: torch.Tensor) -> bool:
    """Eval-mode forwards ALWAYS take the HIP kernels (and therefore raise on CPU tensors or a
    missing library — there is no silent fallback).  Training-mode forwards keep the reference's op
    graph on torch-ROCm so autograd works; set `module.fused_eval = False` to force that graph in
    eval mode too (e.g. to differentiate through an eval-mode model)."""
    return (not module.training) and getattr(module, "fused_eval", True)


def _mit_variant(model_name: str) -> str:
    name = model_name.lower()
    for v in ("b5", "b4", "b3", "b2", "b1", "b0"):
        if f"-{v}" in name or name.endswith(v) or f"_{v}" in name:
            return v
    return "b0"


def _build_mit(model_name: str, pretrained: bool):
    """transformers.SegformerModel with the reference's fall-back config (model.py:120-130), SDPA
    attention.  Hub access does not exist offline; a locally cached checkpoint is used if present."""
    from transformers import SegformerConfig, SegformerModel
    if pretrained:
        try:
            m = SegformerModel.from_pretrained(model_name, local_files_only=True, attn_implementation="sdpa")
            return m
        except Exception as e:  # noqa: BLE001 - same broad fall-back as the reference (:132-146)
            logger.warning("Could not load pretrained SegFormer %s offline (%s); using random init", model_name, type(e).__name__)
    v = MIT_CONFIGS[_mit_variant(model_name)]
    cfg = SegformerConfig(num_channels=3, num_encoder_blocks=4, depths=v["depths"], sr_ratios=[8, 4, 2, 1],
                          hidden_sizes=v["hidden_sizes"], patch_sizes=[7, 3, 3, 3], strides=[4, 2, 2, 2],
                          num_attention_heads=[1, 2, 5, 8], mlp_ratios=[4, 4, 4, 4])
    try:
        cfg._attn_implementation = "sdpa"
    except Exception:  # noqa: BLE001
        pass
    return SegformerModel(cfg)


def _mit_dwconv_forward(self, hidden_states, height, width):
    """transformers' SegformerDepthWiseConv.forward with the TRAINING pass of the depthwise 3x3 on this repo's kernels: the tokens
    [B, N, C] ARE the NHWC map, so no transposes either way (the library forward hands MIOpen a channels-last view, whose grouped
    convolution kernels cost 37 ms per weight gradient at 1024x2048: a quarter of the training step).  Eval and CPU: as written."""
    conv = self.dwconv
    if (torch.is_grad_enabled() and hidden_states.dim() == 3 and hidden_states.is_contiguous() and ops.depthwise_conv3x3_train_ok(conv, hidden_states)):
        b, n, c = hidden_states.shape
        return ops.depthwise_conv3x3_nhwc_train(hidden_states.view(b, height, width, c), conv).view(b, n, c)
    b, n, c = hidden_states.shape
    x = hidden_states.transpose(1, 2).view(b, c, height, width)
    return conv(x).flatten(2).transpose(1, 2)


def _patch_mit_dwconv(seg) -> None:
    import types
    for mod in seg.modules():
        if type(mod).__name__ == "SegformerDepthWiseConv" and isinstance(getattr(mod, "dwconv", None), nn.Conv2d):
            mod.forward = types.MethodType(_mit_dwconv_forward, mod)


_SIDE_STREAMS: Dict[str, "torch.cuda.Stream"] = {}


class DepthEstimationHead(nn.Module):
    """PKG/models/model.py:16-78 — same Sequential layout (keys depth_head.{0,1,4,5,7})."""

    def __init__(self, in_channels: int, hidden_channels: int = 256, out_channels: int = 1, dropout: float = 0.1) -> None:
        super().__init__()
        self.depth_head = nn.Sequential(
            nn.Conv2d(in_channels, hidden_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(hidden_channels),
            nn.ReLU(inplace=True),
            nn.Dropout2d(dropout),
            nn.Conv2d(hidden_channels, hidden_channels // 2, kernel_size=3, padding=1),
            nn.BatchNorm2d(hidden_channels // 2),
            nn.ReLU(inplace=True),
            Conv2d(hidden_channels // 2, out_channels, kernel_size=1),
            nn.Sigmoid())
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        return self.depth_head(features)

    @torch.no_grad()
    def forward_from_lowres(self, feats: torch.Tensor, height: int, width: int) -> torch.Tensor:
        """depth_head(F.interpolate(feats, (H,W))) for eval without the upsampled tensor in front of
        the first 3x3: that conv goes through the same linearity trick as the seg head (HIP), the
        rest (3x3 on the hidden map, 1x1, sigmoid) stays on MIOpen."""
        h = self.depth_head
        Bq, hq, wq, _ = feats.shape
        head64 = fused._is_winograd(h[4]) and h[4].out_channels == 64 and h[7].kernel_size == (1, 1) and h[7].out_channels == 1
        if (ops.DEPTH_FUSED and head64 and height == 32 * hq and width == 32 * wq and h[0].kernel_size == (3, 3) and h[0].padding == (1, 1)
                and (ops.PRECISION == "bf16" or ops.WINO_SPLIT) and feats.is_cuda and wq <= 85):       # (wider: the forms kernel's LDS rows)
            # ONE full-resolution launch: the first 3x3 on the x32 upsampling is a bilinear form per upsampling cell (built at the
            # encoder's resolution), evaluated tile by tile inside the second 3x3's Winograd kernel — no hidden map in HBM
            g9, shift = _head_g9(feats, h[0], h[1])
            forms = ops.upconv_forms(g9, shift)
            bf = ops.PRECISION == "bf16"
            conv, bn = h[4], h[5]
            if bf:
                def build():
                    sc, sh = _fold_conv_bn(conv, bn)
                    return ops.winograd_bf16_weights(conv.weight, sc), sh
                us, sh2 = fused.cached(conv, "wino_bf16", [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var], build)
            else:
                us, sh2 = fused.winograd_split_conv_bn(conv, bn)
            d = ops.depth_head_fused(forms, hq, wq, h[0].out_channels, us, sh2, h[7].weight.view(-1), h[7].bias, bf16=bf)
            return d.unsqueeze(1)
        mid = upconv3x3_bn_relu(feats, h[0], h[1], height, width)            # [B,hidden,H,W], channels_last memory
        if head64:
            # Conv3x3 -> BN -> ReLU -> Conv1x1 -> Sigmoid in one Winograd/MFMA launch (no 64-channel map in HBM)
            d = fused.conv3x3_winograd_bn(fused.nhwc_view(mid), h[4], h[5], w2=h[7].weight.view(-1), b2=h[7].bias)
            return d.unsqueeze(1)
        y = fused.conv_bn_act(mid, h[4], h[5], N.ACT_RELU)
        return torch.sigmoid(h[7](y)).contiguous()

    @torch.no_grad()
    def forward_fused(self, feats_cl: torch.Tensor) -> torch.Tensor:
        """Eval path on a stride-16 feature map (DeepLab branch): BN folded, bias+ReLU one HIP pass."""
        h = self.depth_head
        y = fused.conv_bn_act(feats_cl, h[0], h[1], N.ACT_RELU)
        y = fused.conv_bn_act(y, h[4], h[5], N.ACT_RELU)
        if h[7].kernel_size == (1, 1):
            yl = fused.nhwc_view(y)
            Bq, hh, ww, Cc = yl.shape
            if h[7].out_channels == 1 and Cc % 4 == 0 and yl.is_cuda and yl.dtype == torch.float32:
                d = ops.rowdot_sigmoid(yl.reshape(Bq * hh * ww, Cc), h[7].weight.view(Cc), h[7].bias)     # 1x1 to one channel + Sigmoid
                return d.view(Bq, 1, hh, ww)
            d = torch.addmm(h[7].bias, yl.reshape(Bq * hh * ww, Cc), h[7].weight.view(h[7].out_channels, Cc).t())
            return torch.sigmoid(d).view(Bq, hh, ww, -1).permute(0, 3, 1, 2).contiguous()
        return torch.sigmoid(h[7](y))


def _head_g9(tok: torch.Tensor, conv: nn.Conv2d, bn: nn.BatchNorm2d):
    """The nine per-tap 1x1 products (W_tap * bn_scale) . f at the encoder's resolution,
    [B,h,w,9,Cmid], plus the folded shift.  `tok` is the encoder output as NHWC tokens [B,h,w,Cin].
    BatchNorm's scale is folded into the weights so the kernel's epilogue is `relu(acc + shift)`."""
    B, h, w, Cin = tok.shape
    cmid = conv.weight.shape[0]

    def build():
        scale, shift = _fold_conv_bn(conv, bn)
        w1r = (conv.weight * scale.view(-1, 1, 1, 1)).permute(1, 2, 3, 0).reshape(Cin, 9 * cmid).contiguous()
        # the same matrix as GEMM weights [N = (tap, cout)][K = Cin] with a zero bias: this repo's float32-grade split-operand GEMM takes
        # the product where it accepts the shape (the library's float32 kernel ran it at 67 TFLOP/s: 0.45 ms a step for the two heads)
        return w1r, shift, w1r.t().contiguous(), torch.zeros(9 * cmid, dtype=w1r.dtype, device=w1r.device)

    w1r, shift, wnk, zero_bias = fused.cached(conv, "w1r", [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var], build)
    tok2 = tok.reshape(B * h * w, Cin)
    if tok2.is_cuda and tok2.dtype == torch.float32 and ops.PRECISION != "bf16" and ops.gemm_wants_split(B * h * w, 9 * cmid, Cin):
        ws = fused.cached(conv, "wsplit", (wnk,), lambda: ops.gemm_split_weights(wnk))
        g9 = ops.gemm_bias_act(tok2, wnk, zero_bias, N.ACT_NONE, w_split=ws, split=True)
        return g9.view(B, h, w, 9, cmid), shift
    return (tok2 @ w1r).view(B, h, w, 9, cmid), shift


def _fold_conv_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    s, b = _bn_fold(bn)
    bias = conv.bias if conv.bias is not None else torch.zeros_like(s)
    return s.contiguous(), (bias * s + b).contiguous()


@torch.no_grad()
def upconv3x3_bn_relu(feats, conv, bn, height, width):
    """relu(bn(conv3x3(interpolate(feats)))) at full resolution (HIP, MFMA): the 256-channel
    upsampled tensor in front of the conv is never materialised."""
    g9, shift = _head_g9(feats, conv, bn)
    return ops.upconv3x3_bn_relu(g9, None, shift, height, width, channels_last=True)


class SegFormerModel(nn.Module):
    """PKG/models/model.py:81-223."""

    def __init__(self, model_name: str = "nvidia/segformer-b0-finetuned-ade-512-512", num_classes: int = 19,
                 include_depth: bool = True, pretrained: bool = True, *, compute_dtype: Optional[str] = None) -> None:
        super().__init__()
        self.compute_dtype = compute_dtype          # None / "f32": float32-grade; "bf16": bf16 MFMA contractions (eval path)
        self.num_classes = num_classes
        self.include_depth = include_depth
        self.segformer = _build_mit(model_name, pretrained)
        _patch_mit_dwconv(self.segformer)
        self.feature_dim = getattr(self.segformer.config, "hidden_sizes", [256])[-1]
        self.segmentation_head = nn.Sequential(
            nn.Conv2d(self.feature_dim, 256, kernel_size=3, padding=1),
            nn.BatchNorm2d(256),
            nn.ReLU(inplace=True),
            nn.Dropout2d(0.1),
            Conv2d(256, num_classes, kernel_size=1))
        if self.include_depth:
            self.depth_head = DepthEstimationHead(in_channels=self.feature_dim, hidden_channels=128, out_channels=1)
        for m in self.segmentation_head.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        logger.info("Initialized SegFormer model with %d classes", num_classes)

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        feats = self.segformer(x).last_hidden_state
        if feats.dim() == 3:                                   # model.py:203-207
            B, Nn, Cc = feats.shape
            hh = ww = int(Nn ** 0.5)
            feats = feats.transpose(1, 2).reshape(B, Cc, hh, ww)
        return feats

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        H, W = x.shape[2], x.shape[3]
        if _use_hip(self, x):
            if not x.is_cuda:
                raise N.AwsegError("eval-mode forward runs HIP kernels: it needs CUDA (HIP) tensors; no CPU fallback exists")
            with torch.no_grad(), ops.precision(self.compute_dtype):
                return self._forward_hip(fused.mit_features_nhwc(self.segformer, x), H, W)
        return self.heads_forward(self.encode(x), H, W)

    def heads_forward(self, feats: torch.Tensor, H: int, W: int) -> Dict[str, torch.Tensor]:
        """The two heads on the encoder output [B,C,h,w] in the module's current mode, with autograd (model.py:209-221)."""
        hh, ww = feats.shape[2], feats.shape[3]
        on_hip = feats.is_cuda and getattr(self, "fused_train", True)

        def hip_ok(conv: nn.Conv2d) -> bool:
            # each head is checked with ITS OWN first convolution (a depth head built with other hidden_channels falls back alone)
            return (on_hip and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.stride == (1, 1) and
                    ops.upconv3x3_train_supported(conv.in_channels, conv.out_channels, hh, ww, H, W))

        # on the GPU: the first convolution of each head — conv3x3(interpolate(f)), model.py:209-214 / :219-221 — as one small
        # GEMM at the encoder's resolution + a HIP kernel, forward and backward (ops._UpConv3x3); BatchNorm (batch
        # statistics in training), ReLU, Dropout2d and the remaining layers are the reference's modules on the result
        head = self.segmentation_head
        dh = self.depth_head.depth_head if self.include_depth else None
        seg_hip, dep_hip = hip_ok(head[0]), dh is not None and hip_ok(dh[0])
        tok = feats.permute(0, 2, 3, 1) if (seg_hip or dep_hip) else None
        up = None
        if not seg_hip or (dh is not None and not dep_hip):
            up = F.interpolate(feats, size=(H, W), mode="bilinear", align_corners=False)     # model.py:211
        def tail(seq, z):
            # the layers behind the first convolution: every BatchNorm2d -> ReLU [-> Dropout2d] run is ONE fused pass forward and two
            # backward (ops._BNReLUDropout2d: same batch statistics, same Dropout2d draw); everything else is the reference's module
            mods = list(seq)[1:]
            i = 0
            while i < len(mods):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                drop = mods[i + 2] if (i + 2 < len(mods) and isinstance(mods[i + 2], nn.Dropout2d)) else None
                if isinstance(mods[i], nn.BatchNorm2d) and nxt is not None and ops.bn_relu_dropout2d_train_ok(z, mods[i], nxt, drop):
                    # (i == 0: the producer is ops._UpConv3x3, whose adjoint kernel reads its gradient as NHWC)
                    z = ops.bn_relu_dropout2d_train(z, mods[i], drop, dx_channels_last=(i == 0))
                    i += 3 if drop is not None else 2
                else:
                    z = mods[i](z)
                    i += 1
            return z
        results = {"segmentation": tail(head, ops.upconv3x3_train(tok, head[0].weight, head[0].bias, H, W)) if seg_hip else head(up)}
        if dh is not None:
            results["depth"] = tail(dh, ops.upconv3x3_train(tok, dh[0].weight, dh[0].bias, H, W)) if dep_hip else self.depth_head(up)
        return results

    @torch.no_grad()
    def _forward_hip(self, feats, H, W):
        """feats: encoder output as NHWC tokens [B,h,w,C]."""
        head = self.segmentation_head
        g9, shift = _head_g9(feats, head[0], head[1])
        w2 = head[4].weight.view(self.num_classes, -1)
        results = {"segmentation": ops.segformer_head_fused(g9, None, shift, w2, head[4].bias, H, W)}
        if self.include_depth:
            results["depth"] = self.depth_head.forward_from_lowres(feats, H, W)
        return results


class DeepLabV3PlusModel(nn.Module):
    """PKG/models/model.py:226-374 (the smp branch; the degenerate torchvision fall-back at
    :286-336 is out of scope)."""

    def __init__(self, backbone: str = "resnet50", num_classes: int = 19, include_depth: bool = True,
                 pretrained: bool = True, output_stride: int = 16, *, compute_dtype: Optional[str] = None) -> None:
        super().__init__()
        self.compute_dtype = compute_dtype
        self.num_classes = num_classes
        self.include_depth = include_depth
        if pretrained:
            logger.warning("ImageNet weights are not reachable offline; DeepLabV3+ encoder is randomly initialised")
        # `output_stride` is accepted and ignored exactly as in the reference (model.py:240, 259-265)
        self.model = DeepLabV3Plus(encoder_name=backbone, classes=num_classes)
        self.feature_dim = self.model.encoder.out_channels[-1]
        if self.include_depth:
            self.depth_head = DepthEstimationHead(in_channels=self.feature_dim, hidden_channels=256, out_channels=1)
        logger.info("Initialized DeepLabV3+ model with %s backbone", backbone)

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        if _use_hip(self, x):
            with ops.precision(self.compute_dtype):
                return self._forward_hip(x)
        seg = self.model(x)
        results = {"segmentation": seg}
        if self.include_depth:
            feats = self.model.encoder(x)[-1]                                          # model.py:358
            d = self.depth_head(feats)
            results["depth"] = F.interpolate(d, size=x.shape[2:], mode="bilinear", align_corners=False)
        return results

    @torch.no_grad()
    def _forward_hip(self, x):
        if not x.is_cuda:
            raise N.AwsegError("eval-mode forward runs HIP kernels: it needs CUDA (HIP) tensors; no CPU fallback exists")
        seg, enc = self.model.forward_fused(x, return_features=True)     # (the stem reads x in place; see fused.resnet_features)
        results = {"segmentation": seg.contiguous()}
        if self.include_depth:
            # the reference runs the encoder a second time here (model.py:358); in eval mode the
            # result is identical, so the features of the first pass are reused
            d = self.depth_head.forward_fused(enc)
            if getattr(self, "_defer_depth_upsample", False):
                results["depth_low"] = d.contiguous()        # the ensemble upsamples + combines in one HIP pass
            else:
                results["depth"] = F.interpolate(d, size=x.shape[2:], mode="bilinear", align_corners=False).contiguous()
        return results


_STRATEGY = {"weighted_average": N.COMBINE_WEIGHTED, "max_confidence": N.COMBINE_MAXCONF}


class EnsembleModel(nn.Module):
    """PKG/models/model.py:377-513.  `segformer_name` / `deeplab_backbone` are additive keyword
    extensions (the reference cannot select B5 / R101); defaults match it."""

    def __init__(self, num_classes: int = 19, include_depth: bool = True, ensemble_strategy: str = "weighted_average",
                 temperature_scaling: bool = True, *, segformer_name: Optional[str] = None,
                 deeplab_backbone: str = "resnet50", pretrained: bool = True, compute_dtype: Optional[str] = None) -> None:
        super().__init__()
        self.compute_dtype = compute_dtype
        self.num_classes = num_classes
        self.include_depth = include_depth
        self.ensemble_strategy = ensemble_strategy
        self.temperature_scaling = temperature_scaling
        kw = {} if segformer_name is None else {"model_name": segformer_name}
        self.segformer = SegFormerModel(num_classes=num_classes, include_depth=include_depth, pretrained=pretrained,
                                        compute_dtype=compute_dtype, **kw)
        self.deeplabv3plus = DeepLabV3PlusModel(backbone=deeplab_backbone, num_classes=num_classes,
                                                include_depth=include_depth, pretrained=pretrained, compute_dtype=compute_dtype)
        self.ensemble_weights = nn.Parameter(torch.ones(2) / 2)
        if self.temperature_scaling:
            self.temperature = nn.Parameter(torch.ones(1))
        logger.info("Initialized ensemble model with %s strategy", ensemble_strategy)

    # ---- fused eval entry used by the evaluation harness ----------------------------------
    @torch.no_grad()
    def forward_eval(self, x: torch.Tensor, labels: Optional[torch.Tensor] = None, counts: Optional[torch.Tensor] = None,
                     oob: Optional[torch.Tensor] = None, cond: Optional[torch.Tensor] = None, want_logits: bool = True,
                     want_pred: bool = True, pred_dtype=torch.int64, stats=None) -> Dict[str, torch.Tensor]:
        """members -> ONE pass: combine, /temperature, argmax, confusion (slots: overall + condition).
        stats = (edges, ece_bins, auroc_hist, lo, hi): also accumulate the calibration / disagreement statistics in that pass
        when nothing per-pixel is asked for (`self._stats_fused` tells the caller whether it happened)."""
        # (no channels-last copy of the frames here: both 7x7 stems read the zero-padded 4-channel image that fused._stem_rows builds
        # from the input in whatever layout it has — once per forward, shared through stem_scope; a member whose stem does not take
        # that path converts for itself)
        self.deeplabv3plus._defer_depth_upsample = self.include_depth
        try:
            with fused.stem_scope():                               # the two 7x7 stems share one zero-padded copy of the input
                if ops.TWO_STREAMS and x.is_cuda:
                    # the two members are independent until the combine: DeepLabV3+ runs on a side stream beside SegFormer (whose many
                    # small launches — stages 3 / 4 at 1/16 and 1/32 resolution, LayerNorms, the reduced-token projections — leave
                    # most CUs idle on their own).  One fork, one join; the shared padded image is built in front of the fork.
                    cur = torch.cuda.current_stream(x.device)
                    side = self._side_stream(x.device)
                    if fused.stem_image_ok(x):
                        fused.stem_image(x)
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        o2 = self.deeplabv3plus(x)
                    o1 = self.segformer(x)
                    cur.wait_stream(side)
                    for t in o2.values():
                        t.record_stream(cur)
                else:
                    o1 = self.segformer(x)
                    o2 = self.deeplabv3plus(x)
        finally:
            self.deeplabv3plus._defer_depth_upsample = False
        mode = _STRATEGY.get(self.ensemble_strategy, N.COMBINE_MEAN)
        # (softmax of the two ensemble weights: once per value of the parameter, not once per use — the combine and the depth combine
        # of every eval step read the same two floats)
        w = fused.cached(self, "ens_softmax", [self.ensemble_weights], lambda: F.softmax(self.ensemble_weights, dim=0)) \
            if mode == N.COMBINE_WEIGHTED else None
        T = self.temperature if self.temperature_scaling else None
        if (stats is not None and not want_logits and not want_pred and labels is not None and counts is not None
                and mode in (N.COMBINE_WEIGHTED, N.COMBINE_MEAN) and o1["segmentation"].shape[1] == 19
                and o1["segmentation"][0, 0].numel() % 4 == 0):
            # confusion + ECE bins + disagreement histogram in ONE pass over the member logits (stats = (edges, ece, auroc, lo, hi))
            ops.combine_confusion_stats(o1["segmentation"], o2["segmentation"], mode, w, T, labels, cond, counts, oob, *stats)
            logits, pred = None, None
            self._stats_fused = True
        else:
            self._stats_fused = False
            logits, pred = ops.combine_argmax_confusion(o1["segmentation"], o2["segmentation"], mode, w, T,
                                                        want_logits=want_logits, want_pred=want_pred, pred_dtype=pred_dtype,
                                                        label=labels, counts=counts, oob=oob, cond=cond)
        res = {"segformer_seg": o1["segmentation"], "deeplabv3plus_seg": o2["segmentation"]}
        if logits is not None:
            res["segmentation"] = logits
        if pred is not None:
            res["prediction"] = pred
        if self.include_depth:
            if "depth_low" in o2:
                wd = fused.cached(self, "ens_softmax", [self.ensemble_weights], lambda: F.softmax(self.ensemble_weights, dim=0)) \
                    if self.ensemble_strategy == "weighted_average" else None
                d2, d = ops.depth_upsample_combine(o1["depth"], o2["depth_low"], wd)
                res.update({"depth": d, "segformer_depth": o1["depth"], "deeplabv3plus_depth": d2})
            else:
                res.update(self._combine_depth(o1["depth"], o2["depth"]))
        return res

    def _side_stream(self, device):
        key = str(device)                                           # (module-level: a Stream on the module would break copy.deepcopy(model))
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        return _SIDE_STREAMS[key]

    def _combine_depth(self, d1, d2):
        if self.ensemble_strategy == "weighted_average":                              # model.py:472-475
            w = F.softmax(self.ensemble_weights, dim=0)
            d = w[0] * d1 + w[1] * d2
        else:
            d = (d1 + d2) / 2
        return {"depth": d, "segformer_depth": d1, "deeplabv3plus_depth": d2}

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        if _use_hip(self, x):
            res = self.forward_eval(x, want_logits=True, want_pred=False)
            res.pop("prediction", None)
            return res
        o1 = self.segformer(x)
        o2 = self.deeplabv3plus(x)
        s1, s2 = o1["segmentation"], o2["segmentation"]
        if self.ensemble_strategy == "weighted_average":                              # model.py:443-446
            w = F.softmax(self.ensemble_weights, dim=0)
            seg = w[0] * s1 + w[1] * s2
        elif self.ensemble_strategy == "max_confidence":                              # :447-455
            c1 = F.softmax(s1, dim=1).max(dim=1)[0]
            c2 = F.softmax(s2, dim=1).max(dim=1)[0]
            use = (c1 > c2).float().unsqueeze(1)
            seg = use * s1 + (1 - use) * s2
        else:
            seg = (s1 + s2) / 2
        if self.temperature_scaling:
            seg = seg / self.temperature                                             # :462
        res = {"segmentation": seg, "segformer_seg": s1, "deeplabv3plus_seg": s2}
        if self.include_depth:
            res.update(self._combine_depth(o1["depth"], o2["depth"]))
        return res

    def get_ensemble_disagreement(self, x: torch.Tensor) -> torch.Tensor:
        """PKG/models/model.py:488-513 (note F.kl_div(p.log(), m) = KL(m || p), kept as is)."""
        with torch.no_grad():
            out = self.forward(x)
            p1 = F.softmax(out["segformer_seg"], dim=1)
            p2 = F.softmax(out["deeplabv3plus_seg"], dim=1)
            m = (p1 + p2) / 2
            kl1 = F.kl_div(p1.log(), m, reduction="none").sum(dim=1)
            kl2 = F.kl_div(p2.log(), m, reduction="none").sum(dim=1)
            return (kl1 + kl2) / 2


class _FogCE(torch.autograd.Function):
    """mean_p[ w_p * base(ce_p) ] with HIP forward (awseg_fog_ce_forward) and backward."""

    @staticmethod
    def forward(ctx, logits, label, density, focal, sensitivity, oob):
        mean, _ = ops.fog_ce_forward(logits, label, density, focal, sensitivity, oob)
        ctx.save_for_backward(logits, label, density if density is not None else torch.empty(0, device=logits.device))
        ctx.has_density = density is not None
        ctx.focal, ctx.sens = focal, sensitivity
        return mean[0]

    @staticmethod
    def backward(ctx, g):
        logits, label, dens = ctx.saved_tensors
        grad = ops.fog_ce_backward(logits, label, dens if ctx.has_density else None, ctx.focal, ctx.sens, g)
        return grad, None, None, None, None, None


class FogDensityAwareLoss(nn.Module):
    """PKG/models/model.py:516-677.  Same constructor, same three-key result dict."""

    def __init__(self, base_loss: str = "cross_entropy", depth_weight: float = 0.5, fog_sensitivity: float = 2.0,
                 depth_loss_weight: float = 0.1) -> None:
        super().__init__()
        self.base_loss = base_loss
        self.depth_weight = depth_weight
        self.fog_sensitivity = fog_sensitivity
        self.depth_loss_weight = depth_loss_weight
        self.strict = False        # True: synchronise and raise IndexError on out-of-range labels like torch-CPU does
        self._oob = None

    def label_errors(self) -> int:
        """Number of labels outside [0,C) seen so far (device counter; reading it synchronises)."""
        return 0 if self._oob is None else int(self._oob.item())

    def forward(self, predictions: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor],
                fog_density: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        seg_pred = predictions["segmentation"]
        label = targets["label"]
        if label.dtype not in (torch.uint8, torch.int64):
            label = label.long()                                                      # model.py:578
        if self._oob is None or self._oob.device != seg_pred.device:
            self._oob = torch.zeros(1, dtype=torch.int64, device=seg_pred.device)
        focal = self.base_loss == "focal"
        depth_loss = 0.0
        density = fog_density
        if "depth" in predictions and self.depth_weight > 0:                          # :590
            pred_depth = predictions["depth"].squeeze(1)
            if density is None:
                if pred_depth.requires_grad:
                    density = self._estimate_fog_density_from_depth_torch(pred_depth)  # keeps d(loss)/d(depth)
                else:
                    density = ops.fog_density_from_depth(pred_depth)                  # :595
            if "depth" in targets:                                                    # :601-604
                depth_loss = F.mse_loss(pred_depth, targets["depth"], reduction="none").mean()
        if density is not None and density.requires_grad:
            # differentiable density (rare from-depth branch while training): torch graph
            ce = F.cross_entropy(seg_pred, label.long(), reduction="none")
            if focal:
                ce = (1 - torch.exp(-ce)) ** 2 * ce
            total_seg = (ce * (1.0 + self.fog_sensitivity * density)).mean()
        else:
            dens = None if density is None else density.to(torch.float32)
            total_seg = _FogCE.apply(seg_pred, label, dens, focal, float(self.fog_sensitivity), self._oob)
        if self.strict and self.label_errors():
            raise IndexError("Target out of bounds")
        total = total_seg + self.depth_loss_weight * depth_loss                       # :611
        return {"total_loss": total, "segmentation_loss": total_seg, "depth_loss": depth_loss}

    def _estimate_fog_density_from_depth(self, depth: torch.Tensor) -> torch.Tensor:
        return ops.fog_density_from_depth(depth)

    @staticmethod
    def _estimate_fog_density_from_depth_torch(depth: torch.Tensor) -> torch.Tensor:
        """Autograd-capable restatement of model.py:658-677 for the training-time from-depth branch."""
        dn = (depth - depth.min()) / (depth.max() - depth.min() + 1e-8)
        fog = dn * 0.7
        gx = F.pad(torch.abs(depth[:, :, 1:] - depth[:, :, :-1]), (0, 1, 0, 0), mode="replicate")
        gy = F.pad(torch.abs(depth[:, 1:, :] - depth[:, :-1, :]), (0, 0, 0, 1), mode="replicate")
        mag = torch.sqrt(gx ** 2 + gy ** 2 + 1e-8)
        return torch.clamp(fog - (mag > mag.mean()) * 0.3, 0, 1)
