"""DeepLabV3+ (ResNet encoder, separable ASPP, decoder) with segmentation_models_pytorch's
module tree and state_dict key names, written from scratch for torch-ROCm + HIP.

The reference delegates this whole network to ``smp.DeepLabV3Plus(encoder_name='resnet50',
classes=C, activation=None)`` (PKG/models/model.py:259-265); smp / torchvision are not
available offline, so structure and key names follow the published packages from memory
(parity for this part is against this repo's own as-written path only).

Two forwards over the same parameters:
  * ``forward``            — the as-written module graph (autograd-capable; training path)
  * ``forward_fused``      — eval path: channels-last convs, the three atrous depthwise
                             convolutions of the ASPP in ONE HIP pass (awseg_aspp_depthwise3),
                             BatchNorm folded into the pointwise GEMMs, no 1280-channel concat
"""
from __future__ import annotations

import os

from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters, same state_dict keys) whose TRAINING forward of a 1x1 convolution on the GPU is a GEMM on a
    channels-last view: `F.linear` goes to hipBLASLt with ~50 us of host time per call, forward and backward, where MIOpen's
    immediate mode costs ~7 ms of HOST time per convolution call on this stack (cProfile of the 1024x2048 training step: the host,
    not the GPU, bounds the step) — and 44 of DeepLabV3+-R50's 66 convolutions are 1x1.  Same arithmetic as the
    convolution (one dot product of Cin terms per output); inference under `torch.no_grad()` keeps F.conv2d, so the as-written
    reference graph the tests compare against is unchanged."""

    linear_in_training = True
    # (AWSEG_TRAIN_KEEP_CL=0: copy back to NCHW after every 1x1, as in round 3)  The F.linear result stays a channels-last-strided tensor — BatchNorm / ReLU / the
    # residual add behind it run on that layout and the NEXT 1x1 reads it without a copy; only a spatial convolution copies its input
    # back to NCHW (MIOpen's NHWC picks for those are the slow ones).  Halves the layout copies of a bottleneck.
    keep_channels_last = os.environ.get("AWSEG_TRAIN_KEEP_CL", "1") != "0"
    matmul_nchw = os.environ.get("AWSEG_TRAIN_MATMUL_NCHW", "1") != "0"     # 0: the channels-last F.linear form for every input layout

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (Conv2d.linear_in_training and x.is_cuda and torch.is_grad_enabled() and self.kernel_size == (1, 1) and self.groups == 1
                and self.padding == (0, 0) and x.dim() == 4 and x.shape[2] * x.shape[3] > 1):
            if self.stride != (1, 1):
                x = x[:, :, ::self.stride[0], ::self.stride[1]]
            b, c, h, w = x.shape
            if Conv2d.matmul_nchw and not (x.stride(1) == 1 and c > 1):
                # NCHW (or any non-channels-last) input: out[b] = W [o, c] @ x[b] [c, hw] — a batched GEMM in the tensor's OWN layout,
                # forward and backward, so no layout copy exists anywhere around the layer (round 3's channels-last F.linear form
                # copied every NCHW input and every gradient through torch's transposing copy kernel: 0.7 TB/s on the 17 GB head maps,
                # 0.27 s of a 1.30 s step, profiles/r04_train_step_kernels.csv)
                x3 = x.reshape(b, c, h * w)                              # (a view for contiguous x; the strided downsample input copies a quarter)
                w3 = self.weight.view(1, self.out_channels, c).expand(b, -1, -1)
                y = torch.bmm(w3, x3) if self.bias is None else torch.baddbmm(self.bias.view(1, -1, 1), w3, x3)
                return y.view(b, self.out_channels, h, w)
            y = F.linear(x.permute(0, 2, 3, 1).reshape(b * h * w, c), self.weight.view(self.out_channels, c), self.bias)
            y = y.view(b, h, w, self.out_channels).permute(0, 3, 1, 2)
            return y if Conv2d.keep_channels_last else y.contiguous()
        if Conv2d.keep_channels_last and x.is_cuda and torch.is_grad_enabled() and x.dim() == 4 and not x.is_contiguous():
            x = x.contiguous()
        return super().forward(x)


# --------------------------------------------------------------------------- ResNet encoder
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


_RESNET_LAYERS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}


class ResNetEncoder(nn.Module):
    """torchvision ResNet trunk (no fc), smp's feature list and `make_dilated(output_stride=16)`."""

    out_channels = (3, 64, 256, 512, 1024, 2048)

    def __init__(self, name="resnet50", output_stride=16):
        super().__init__()
        layers = _RESNET_LAYERS.get(name, _RESNET_LAYERS["resnet50"])
        self.inplanes = 64
        self.conv1 = Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        for m in self.modules():                         # torchvision's init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if output_stride == 16:
            self._dilate(self.layer4, 2)
        elif output_stride == 8:
            self._dilate(self.layer3, 2)
            self._dilate(self.layer4, 4)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    @staticmethod
    def _dilate(layer, rate):
        """smp.base.utils.replace_strides_with_dilation: every conv loses its stride, gets the rate."""
        for m in layer.modules():
            if isinstance(m, nn.Conv2d):
                m.stride = (1, 1)
                m.dilation = (rate, rate)
                kh, kw = m.kernel_size
                m.padding = ((kh // 2) * rate, (kw // 2) * rate)

    def forward(self, x) -> List[torch.Tensor]:
        feats = [x]
        x = self.relu(self.bn1(self.conv1(x)))
        feats.append(x)
        x = self.layer1(self.maxpool(x))
        feats.append(x)
        x = self.layer2(x)
        feats.append(x)
        x = self.layer3(x)
        feats.append(x)
        x = self.layer4(x)
        feats.append(x)
        return feats


# --------------------------------------------------------------------------- decoder pieces
class SeparableConv2d(nn.Sequential):
    def __init__(self, in_ch, out_ch, kernel_size, padding=0, dilation=1, bias=True):
        super().__init__(
            Conv2d(in_ch, in_ch, kernel_size, padding=padding, dilation=dilation, groups=in_ch, bias=False),
            Conv2d(in_ch, out_ch, 1, bias=bias))

    def forward(self, x):
        dw, pw = self[0], self[1]
        if torch.is_grad_enabled() and x.dim() == 4 and ops.depthwise_conv3x3_train_ok(dw, x) and Conv2d.linear_in_training and pw.kernel_size == (1, 1):
            # training: depthwise half on this repo's NHWC kernels (forward, input gradient, weight gradient) and the pointwise half as
            # the GEMM it is, on the same channels-last tensor — MIOpen's picks for the grouped convolution cost 37 + 5 + 6 ms per call
            b, c, h, w = x.shape
            y = ops.depthwise_conv3x3_nhwc_train(x.permute(0, 2, 3, 1).contiguous(), dw)
            z = F.linear(y.view(b * h * w, c), pw.weight.view(pw.out_channels, c), pw.bias)
            return z.view(b, h, w, pw.out_channels).permute(0, 3, 1, 2).contiguous()
        return super().forward(x)


class ASPPSeparableConv(nn.Sequential):
    def __init__(self, in_ch, out_ch, dilation):
        super().__init__(SeparableConv2d(in_ch, out_ch, 3, padding=dilation, dilation=dilation, bias=False),
                         nn.BatchNorm2d(out_ch), nn.ReLU())


class ASPPPooling(nn.Sequential):
    def __init__(self, in_ch, out_ch):
        super().__init__(nn.AdaptiveAvgPool2d(1), Conv2d(in_ch, out_ch, 1, bias=False), nn.BatchNorm2d(out_ch), nn.ReLU())

    def forward(self, x):
        size = x.shape[-2:]
        for mod in self:
            x = mod(x)
        return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


class ASPP(nn.Module):
    def __init__(self, in_ch, out_ch, atrous_rates):
        super().__init__()
        self.rates = tuple(atrous_rates)
        mods = [nn.Sequential(Conv2d(in_ch, out_ch, 1, bias=False), nn.BatchNorm2d(out_ch), nn.ReLU())]
        mods += [ASPPSeparableConv(in_ch, out_ch, r) for r in self.rates]
        mods.append(ASPPPooling(in_ch, out_ch))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(Conv2d(5 * out_ch, out_ch, 1, bias=False), nn.BatchNorm2d(out_ch), nn.ReLU(),
                                     nn.Dropout(0.5))

    def forward(self, x):
        return self.project(torch.cat([conv(x) for conv in self.convs], dim=1))


def _bn_fold(bn: nn.BatchNorm2d):
    """eval-mode BatchNorm as y*scale + shift."""
    inv = torch.rsqrt(bn.running_var + bn.eps)
    scale = bn.weight * inv
    return scale, bn.bias - bn.running_mean * scale


class DeepLabV3PlusDecoder(nn.Module):
    def __init__(self, encoder_channels, out_channels=256, atrous_rates=(12, 24, 36), output_stride=16):
        super().__init__()
        self.out_channels = out_channels
        self.aspp = nn.Sequential(ASPP(encoder_channels[-1], out_channels, atrous_rates),
                                  SeparableConv2d(out_channels, out_channels, 3, padding=1, bias=False),
                                  nn.BatchNorm2d(out_channels), nn.ReLU())
        self.up = nn.UpsamplingBilinear2d(scale_factor=2 if output_stride == 8 else 4)   # align_corners=True
        self.block1 = nn.Sequential(Conv2d(encoder_channels[-4], 48, 1, bias=False), nn.BatchNorm2d(48), nn.ReLU())
        self.block2 = nn.Sequential(SeparableConv2d(48 + out_channels, out_channels, 3, padding=1, bias=False),
                                    nn.BatchNorm2d(out_channels), nn.ReLU())

    def forward(self, *features):
        a = self.up(self.aspp(features[-1]))
        hi = self.block1(features[-4])
        return self.block2(torch.cat([a, hi], dim=1))

    # ---- eval fusion -----------------------------------------------------------------
    @torch.no_grad()
    def aspp_fused(self, x: torch.Tensor) -> torch.Tensor:
        """ASPP + projection for eval: x [B,2048,h,w] -> [B,256,h,w].
        One HIP pass produces the three atrous depthwise maps; each branch is then a GEMM with
        BatchNorm folded into its weights, and the 1x1 projection is accumulated branch by
        branch (project(cat(b_i)) = sum_i b_i @ P_i) so the 1280-channel concat never exists."""
        aspp: ASPP = self.aspp[0]
        B, Cin, h, w = x.shape
        xl = x.permute(0, 2, 3, 1).contiguous()                      # NHWC (free when x is channels_last)
        flat = xl.view(B * h * w, Cin)
        Cout = self.out_channels
        P = aspp.project[0].weight.view(Cout, 5 * Cout)              # [out, 5*in]

        def branch(inp, wmat, bn):
            from . import fused
            want = ops.gemm_wants_split(inp.shape[0], wmat.shape[0], wmat.shape[1])

            def fold():
                s, b = _bn_fold(bn)
                wf = (wmat * s[:, None]).contiguous()
                return wf, b.contiguous(), (ops.gemm_split_weights(wf) if want else None)
            wf, b, ws = fused.cached(bn, "aspp_fold%d" % int(want), (wmat, bn.weight, bn.bias, bn.running_mean, bn.running_var), fold)
            return ops.gemm_bias_act(inp, wf, b, 1, w_split=ws)             # GEMM with bias + ReLU in the epilogue

        # projection: project(cat(b_0..b_4)) -> BN -> ReLU = relu(sum_i b_i (ps * P_i)^T + ps * (g P_4^T) + pb).  The pooled
        # branch g is one row per image, so it enters as the initial value of the accumulator; the four pixel branches are
        # GEMMs that accumulate in place (residual = out), the last one with the ReLU in its epilogue.
        pbn = aspp.project[1]

        def fold_proj():
            ps, pb = _bn_fold(pbn)
            Pf = (P * ps[:, None]).contiguous()
            parts = [Pf[:, i * Cout:(i + 1) * Cout].contiguous() for i in range(5)]
            want = ops.gemm_wants_split(B * h * w, Cout, Cout)
            return parts, [ops.gemm_split_weights(pp) if want else None for pp in parts[:4]], torch.zeros(Cout, device=P.device), pb.contiguous()
        from . import fused
        parts, psplit, zero_bias, pb = fused.cached(aspp.project[0], "proj_fold%d" % int(ops.gemm_wants_split(B * h * w, Cout, Cout)),
                                                (aspp.project[0].weight, pbn.weight, pbn.bias, pbn.running_mean, pbn.running_var), fold_proj)
        # the depthwise halves of branches 1-3 (HIP, all three rates in one pass) — the same pass leaves the pooling branch's global
        # average (its rate-0 blocks meet every pixel of their channels once)
        dws = [aspp.convs[1 + r][0][0].weight for r in range(3)]
        wdw = fused.cached(aspp, "dw3taps", dws, lambda: torch.stack([t_.view(Cin, 9).t() for t_ in dws]).contiguous())  # [3,9,C]
        if xl.is_cuda and xl.dtype == torch.float32:
            dw, gmean = ops.aspp_depthwise3_mean(xl, wdw, aspp.rates)
        else:
            dw, gmean = ops.aspp_depthwise3(xl, wdw, aspp.rates), None
        dw = dw.view(3, B * h * w, Cin)
        if gmean is None:
            gmean = xl.mean(dim=(1, 2))
        # pooling branch: global mean -> 1x1 -> BN -> ReLU; bilinear upsample of a 1x1 map is a broadcast
        pool = aspp.convs[4]
        pw_, pbn_ = pool[1].weight.view(Cout, Cin), pool[2]
        if xl.is_cuda and xl.dtype == torch.float32:
            def fold_pool():
                s_, b_ = _bn_fold(pbn_)
                return (pw_ * s_[:, None]).contiguous(), b_.contiguous()
            w1f, b1f = fused.cached(pbn_, "aspp_pool_fold", (pw_, pbn_.weight, pbn_.bias, pbn_.running_mean, pbn_.running_var), fold_pool)
            g2 = ops.aspp_pool_branch(gmean, w1f, b1f, parts[4], pb)                         # [B,256]: conv + BN + ReLU + projection slice
        else:
            g2 = branch(gmean, pw_, pbn_) @ parts[4].t() + pb
        acc = g2[:, None, :].expand(B, h * w, Cout).contiguous().view(B * h * w, Cout)
        ys = [branch(flat, aspp.convs[0][0].weight.view(Cout, Cin), aspp.convs[0][1])]                 # branch 0: 1x1
        for r in range(3):                                                                            # branches 1-3: their pointwise GEMMs
            mod = aspp.convs[1 + r]
            ys.append(branch(dw[r], mod[0][1].weight.view(Cout, Cin), mod[1]))
        done = None
        if ops.ASPP_PIECES and ops.GEMM_SPLIT and ops.PRECISION != "bf16" and acc.is_cuda and Cout % 32 == 0:
            # the projection of the four pixel branches as ONE product over [b0 | b1 | b2 | b3] (the pieces stay where they are)
            wcat = fused.cached(aspp.project[0], "proj_pieces", (parts[0], parts[1], parts[2], parts[3]),
                                lambda: ops.gemm_split_weights(torch.cat(parts[:4], dim=1).contiguous()))
            done = ops.gemm_split_pieces(ys, wcat, zero_bias, 1, residual=acc, out=acc)
        if done is None:
            for i, y in enumerate(ys):
                ops.gemm_bias_act(y, parts[i], zero_bias, 1 if i == 3 else 0, residual=acc, out=acc, w_split=psplit[i])
        out = acc                                                                     # project BN + ReLU done (Dropout: eval)
        return out.view(B, h, w, Cout).permute(0, 3, 1, 2)                            # NCHW view, channels_last memory

    @torch.no_grad()
    def forward_fused(self, *features):
        from . import fused
        from .. import _native as N
        a = self.aspp_fused(features[-1])
        a = fused.separable_bn_relu(a, self.aspp[1], self.aspp[2])          # SeparableConv2d -> BN -> ReLU
        hi = fused.conv_bn_act(features[-4], self.block1[0], self.block1[1], N.ACT_RELU)
        # up x4 (align_corners=True) -> cat -> depthwise 3x3 in one HIP pass (no upsampled map, no concat), then the
        # pointwise half of block2's SeparableConv2d with BN + ReLU in the GEMM epilogue
        dwc, pwc = self.block2[0][0], self.block2[0][1]
        d = ops.dwconv3x3_upcat(fused.nhwc_view(a), fused.nhwc_view(hi), fused.dw_taps(dwc))
        B, H4, W4, Cc = d.shape
        w, shift = fused.folded_conv_bn(pwc, self.block2[1])
        w2 = w.view(w.shape[0], Cc)
        y = ops.gemm_bias_act(d.view(B * H4 * W4, Cc), w2, shift, N.ACT_RELU, w_split=fused.split_weights(pwc, w2, B * H4 * W4))
        return y.view(B, H4, W4, -1).permute(0, 3, 1, 2)


class SegmentationHead(nn.Sequential):
    def __init__(self, in_ch, out_ch, kernel_size=1, upsampling=4):
        super().__init__(Conv2d(in_ch, out_ch, kernel_size, padding=kernel_size // 2),
                         nn.UpsamplingBilinear2d(scale_factor=upsampling) if upsampling > 1 else nn.Identity(),
                         nn.Identity())


class DeepLabV3Plus(nn.Module):
    """smp.DeepLabV3Plus(encoder_name, encoder_weights=None, classes, activation=None) counterpart."""

    def __init__(self, encoder_name="resnet50", classes=19, encoder_output_stride=16, decoder_channels=256,
                 decoder_atrous_rates=(12, 24, 36), upsampling=4):
        super().__init__()
        self.encoder = ResNetEncoder(encoder_name, encoder_output_stride)
        self.decoder = DeepLabV3PlusDecoder(self.encoder.out_channels, decoder_channels, decoder_atrous_rates,
                                            encoder_output_stride)
        self.segmentation_head = SegmentationHead(decoder_channels, classes, 1, upsampling)
        for m in list(self.decoder.modules()) + list(self.segmentation_head.modules()):   # smp initialize_decoder/head
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        nn.init.xavier_uniform_(self.segmentation_head[0].weight)

    def forward(self, x):
        feats = self.encoder(x)
        return self.segmentation_head(self.decoder(*feats))

    @torch.no_grad()
    def forward_fused(self, x, return_features=False):
        from . import fused
        feats = fused.resnet_features(self.encoder, x)
        dec = self.decoder.forward_fused(*feats)
        head = self.segmentation_head[0]
        if head.kernel_size == (1, 1):
            # the classifier as a GEMM on the NHWC rows (deterministic accumulation order; MIOpen's 1x1 path is not promised to be)
            dl = fused.nhwc_view(dec)
            Bq, H4, W4, Cd = dl.shape
            # (the dispatcher's choice: on 10^6 rows the 19-class head is one masked 64-column tile of this repo's split-operand GEMM,
            # bound by reading the decoder map once; smaller problems stay on the library's float32 kernel)
            y2 = ops.gemm_bias_act(dl.reshape(Bq * H4 * W4, Cd), head.weight.view(head.out_channels, Cd), head.bias, 0,
                                   w_split=fused.split_weights(head, head.weight.view(head.out_channels, Cd), Bq * H4 * W4))
            low = y2.view(Bq, H4, W4, -1).permute(0, 3, 1, 2)               # [B,C,H/4,W/4] view of the NHWC rows (ops.upsample_bilinear reads strides)
        else:
            low = head(dec).contiguous()
        up = self.segmentation_head[1]
        if isinstance(up, nn.UpsamplingBilinear2d) and low.is_cuda and low.dtype == torch.float32:
            f = int(up.scale_factor)
            out = ops.upsample_bilinear(low, (low.shape[2] * f, low.shape[3] * f), True)   # x4 bilinear, align_corners=True (HIP)
        else:
            out = up(low)
        return (out, feats[-1]) if return_features else out
