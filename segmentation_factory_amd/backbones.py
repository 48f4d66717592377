"""Backbones behind the reference's plugin contract (models/build_models.py:25-29): a callable name that
returns an nn.Module exposing ``.channels`` and ``forward(x[B,3,H,W]) -> 4 NCHW feature maps``.

Internally every activation is a token-major ``[B*H*W, C]`` tensor in the compute dtype and every op is
a HIP kernel (segmentation_factory_amd.functional).  The NCHW maps handed to the head are zero-copy
permuted views of those token buffers.
"""

import torch
from torch import nn

from . import functional as Fh
from .containers import (BatchNormWeights, ChannelsFirstLayerNormWeights, ConvWeights, LayerNormWeights, LinearWeights,
                         init_mit_style)

# reference models/backbones/mit.py:149-156
mit_settings = {
    'B0': [[32, 64, 160, 256], [2, 2, 2, 2]],
    'B1': [[64, 128, 320, 512], [2, 2, 2, 2]],
    'B2': [[64, 128, 320, 512], [3, 4, 6, 3]],
    'B3': [[64, 128, 320, 512], [3, 4, 18, 3]],
    'B4': [[64, 128, 320, 512], [3, 8, 27, 3]],
    'B5': [[64, 128, 320, 512], [3, 6, 40, 3]],
}


class TokenMap:
    """A feature map as tokens: data [B*H*W, C] (+ geometry).  ``nchw()`` is the plugin-API view."""
    __slots__ = ('data', 'B', 'H', 'W')

    def __init__(self, data, B, H, W):
        self.data, self.B, self.H, self.W = data, B, H, W

    def nchw(self):
        return self.data.view(self.B, self.H, self.W, -1).permute(0, 3, 1, 2)


def tokens_from_nchw(x, dtype):
    """Accept an NCHW tensor from a foreign backbone/head: zero-copy when it is a permuted NHWC buffer."""
    B, C, H, W = x.shape
    t = x.permute(0, 2, 3, 1)
    if not t.is_contiguous() or t.dtype != dtype:
        from . import hip
        src = x.contiguous()
        t = hip.permute021(src, B, C, H * W, dtype)          # NCHW -> NHWC re-layout kernel
        return TokenMap(t.view(B * H * W, C), B, H, W)
    return TokenMap(t.reshape(B * H * W, C), B, H, W)


class Attention(nn.Module):
    """Spatial-reduction attention (mit.py:9-59)."""

    def __init__(self, dim, head, sr_ratio):
        super().__init__()
        self.head, self.sr_ratio, self.dim = head, sr_ratio, dim
        self.q = LinearWeights(dim, dim)
        self.kv = LinearWeights(dim, dim * 2)
        self.proj = LinearWeights(dim, dim)
        if sr_ratio > 1:
            self.sr = ConvWeights(dim, dim, sr_ratio, sr_ratio)
            self.norm = LayerNormWeights(dim)
        self.apply(init_mit_style)

    def tokens(self, h, B, H, W, residual, rscale, col=None):
        N = H * W
        if col is not None:
            # the norm in front already wrote its output as the im2col matrix of the spatial-reduction convolution (layer_norm_res_patch):
            # no im2col / col2im passes, and the two consumers' gradients meet in the LayerNorm backward
            sr = self.sr_ratio
            q = Fh.linear(h, self.q.weight, self.q.bias)
            xr = Fh.conv_from_col(col, self.sr.weight, self.sr.bias, sr)
            xr = Fh.layer_norm(xr, self.norm.weight, self.norm.bias, self.norm.eps)
            Nkv = (H // sr) * (W // sr)
            kv = Fh.linear(xr, self.kv.weight, self.kv.bias)
            o = Fh.attention(q, kv, B, N, Nkv, self.head)
            return Fh.linear(o, self.proj.weight, self.proj.bias, residual=residual, rscale=rscale, rows_per_group=N)
        # two consumers (q and the key / value path): the key / value path's gradient joins inside q's data-gradient product
        q, h = Fh.linear_fork(h, self.q.weight, self.q.bias)
        if self.sr_ratio > 1:
            sr = self.sr_ratio
            xr = Fh.conv_patch(h, self.sr.weight, self.sr.bias, (B, H, W, self.dim, sr, sr, 0))
            xr = Fh.layer_norm(xr, self.norm.weight, self.norm.bias, self.norm.eps)
            Nkv = ((H - sr) // sr + 1) * ((W - sr) // sr + 1)
        else:
            xr, Nkv = h, N
        kv = Fh.linear(xr, self.kv.weight, self.kv.bias)
        o = Fh.attention(q, kv, B, N, Nkv, self.head)
        return Fh.linear(o, self.proj.weight, self.proj.bias, residual=residual, rscale=rscale, rows_per_group=N)


class DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = ConvWeights(dim, dim, 3, 1, 1, groups=dim)


class MLP(nn.Module):
    """fc1 -> depthwise 3x3 -> GELU -> fc2 (mit.py:74-99)."""

    def __init__(self, c1, c2):
        super().__init__()
        self.fc1 = LinearWeights(c1, c2)
        self.dwconv = DWConv(c2)
        self.fc2 = LinearWeights(c2, c1)
        self.apply(init_mit_style)

    def tokens(self, h, B, H, W, residual, rscale):
        f = Fh.linear(h, self.fc1.weight, self.fc1.bias)
        g = Fh.dwconv3x3_gelu(f, self.dwconv.dwconv.weight, self.dwconv.dwconv.bias, B, H, W, True)
        return Fh.linear(g, self.fc2.weight, self.fc2.bias, residual=residual, rscale=rscale, rows_per_group=H * W)


class PatchEmbed(nn.Module):
    """Overlapped patch embedding: strided conv + LayerNorm (mit.py:102-131)."""

    def __init__(self, c1=3, c2=32, patch_size=7, stride=4):
        super().__init__()
        self.c1, self.k, self.stride = c1, patch_size, stride
        self.proj = ConvWeights(c1, c2, patch_size, stride, patch_size // 2)
        self.norm = LayerNormWeights(c2)
        self.apply(init_mit_style)

    def tokens(self, x, B, H, W, image, dtype):
        k, s, p = self.k, self.stride, self.k // 2
        t = Fh.conv_patch(x, self.proj.weight, self.proj.bias, (B, H, W, self.c1, k, s, p), image=image, dtype=dtype)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        return Fh.layer_norm(t, self.norm.weight, self.norm.bias, self.norm.eps), Ho, Wo


class Block(nn.Module):
    """x + DropPath(Attn(LN(x)));  x + DropPath(MLP(LN(x)))   (mit.py:134-146).  The residual add and the
    per-sample DropPath scale are fused into the epilogue of the proj / fc2 GEMMs."""

    def __init__(self, dim, head, sr_ratio=1, dpr=0.):
        super().__init__()
        self.norm1 = LayerNormWeights(dim)
        self.attn = Attention(dim, head, sr_ratio)
        self.drop_prob = float(dpr)
        self.norm2 = LayerNormWeights(dim)
        self.mlp = MLP(dim, int(dim * 4))

    def tokens(self, x, B, H, W, scales):
        s1, s2 = scales
        sr = self.attn.sr_ratio
        if Fh.patch_layout_ok(W, H, sr) and x.is_cuda:
            x, h, col = Fh.layer_norm_res_patch(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, W, sr)
            x = self.attn.tokens(h, B, H, W, x, s1, col=col)
        else:
            x, h = Fh.layer_norm_res(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)     # x passes through for the residual
            x = self.attn.tokens(h, B, H, W, x, s1)
        x, h = Fh.layer_norm_res(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return self.mlp.tokens(h, B, H, W, x, s2)


class MiT(nn.Module):
    """Mix Transformer encoder (mit.py:159-218), variants B0-B5."""

    def __init__(self, model_name: str = 'B0', **kwargs):
        super().__init__()
        assert model_name in mit_settings.keys(), f"MiT model name should be in {list(mit_settings.keys())}"
        embed_dims, depths = mit_settings[model_name]
        drop_path_rate = 0.1
        self.channels = embed_dims
        self.depths = depths
        self.compute_dtype = torch.bfloat16
        self.stochastic_override = None      # tests: {'drop_path': keep[n_draws, B]}

        self.patch_embed1 = PatchEmbed(3, embed_dims[0], 7, 4)
        self.patch_embed2 = PatchEmbed(embed_dims[0], embed_dims[1], 3, 2)
        self.patch_embed3 = PatchEmbed(embed_dims[1], embed_dims[2], 3, 2)
        self.patch_embed4 = PatchEmbed(embed_dims[2], embed_dims[3], 3, 2)

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        heads, srs = [1, 2, 5, 8], [8, 4, 2, 1]
        cur = 0
        for s in range(4):
            blocks = nn.ModuleList([Block(embed_dims[s], heads[s], srs[s], dpr[cur + i]) for i in range(depths[s])])
            setattr(self, f'block{s + 1}', blocks)
            setattr(self, f'norm{s + 1}', LayerNormWeights(embed_dims[s]))
            cur += depths[s]

    def _drop_path_scales(self, B, device):
        """One [n_draws, B] tensor of keep/kp scales per forward (drop_path.py:18-25: x/kp*floor(kp+U)); blocks
        with rate 0 use nn.Identity in the reference and draw nothing."""
        rates = [blk.drop_prob for s in range(4) for blk in getattr(self, f'block{s + 1}')]
        if not self.training or all(r == 0 for r in rates):
            return [(None, None)] * len(rates)
        draws = [r for r in rates if r > 0 for _ in range(2)]
        if self.stochastic_override is not None and 'drop_path' in self.stochastic_override:
            kp = 1.0 - torch.tensor(draws, dtype=torch.float32, device=device)[:, None]
            scale = (self.stochastic_override['drop_path'].to(device=device, dtype=torch.float32) / kp).contiguous()
        else:
            scale = Fh.stochastic_scales(self, tuple(1.0 - r for r in draws), B, device)       # keep / kp, one row per draw
        out, i = [], 0
        for r in rates:
            if r > 0:
                out.append((scale[i], scale[i + 1]))
                i += 2
            else:
                out.append((None, None))
        return out

    def forward_tokens(self, x):
        """x: fp32 NCHW image.  Returns 4 TokenMaps (strides 4/8/16/32)."""
        B, _, H, W = x.shape
        dtype = self.compute_dtype
        scales = self._drop_path_scales(B, x.device)
        outs, bi = [], 0
        cur, image = x, True
        for s in range(4):
            pe = getattr(self, f'patch_embed{s + 1}')
            t, H, W = pe.tokens(cur, B, H, W, image, dtype)
            for blk in getattr(self, f'block{s + 1}'):
                t = blk.tokens(t, B, H, W, scales[bi])
                bi += 1
            nrm = getattr(self, f'norm{s + 1}')
            if s < 3:       # the stage output feeds the decode head AND the next patch embedding: one LayerNorm, two consumers
                t_head, t = Fh.layer_norm_fork(t, nrm.weight, nrm.bias, nrm.eps)
            else:
                t_head = t = Fh.layer_norm(t, nrm.weight, nrm.bias, nrm.eps)
            outs.append(TokenMap(t_head, B, H, W))
            cur, image = t, False
        return outs

    def forward(self, x):
        return tuple(tm.nchw() for tm in self.forward_tokens(x))


# ---- ConvNeXt / ConvNeXtV2 (models/backbones/convnext.py, convnextv2.py) -------------------------------------------------
# reference models/backbones/convnext.py:70-76
convnext_settings = {
    'T': [[3, 3, 9, 3], [96, 192, 384, 768], 0.1],
    'S': [[3, 3, 27, 3], [96, 192, 384, 768], 0.4],
    'B': [[3, 3, 27, 3], [128, 256, 512, 1024], 0.5],
    'L': [[3, 3, 27, 3], [192, 384, 768, 1536], 0.5],
    'XL': [[3, 3, 27, 3], [256, 512, 1024, 2048], 0.5],
}


def _init_convnext(m):
    """convnext.py:103-106 / convnextv2.py:164-167: trunc_normal(.02) weights, zero bias for Conv2d and Linear."""
    from .containers import trunc_normal_
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        trunc_normal_(m.weight, std=.02)
        nn.init.constant_(m.bias, 0)


class GRNWeights(nn.Module):
    """Global Response Normalization parameters (convnextv2.py:68-80): gamma, beta of shape [1, 1, 1, dim]."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1, 1, 1, dim))
        self.beta = nn.Parameter(torch.zeros(1, 1, 1, dim))


class ConvNeXtBlock(nn.Module):
    """dwconv 7x7 -> LayerNorm -> Linear 4x -> GELU -> [GRN] -> Linear -> [x gamma] -> DropPath + residual
    (convnext.py:26-51; convnextv2.py:83-113).  The layer scale is folded into pwconv2's parameters and the residual +
    per-sample DropPath scale into its GEMM epilogue."""

    def __init__(self, dim, dpr=0., init_value=1e-6, v2=False):
        super().__init__()
        self.dim, self.v2 = dim, v2
        self.dwconv = ConvWeights(dim, dim, 7, 1, 3, groups=dim)
        self.norm = LayerNormWeights(dim, eps=1e-6)
        self.pwconv1 = LinearWeights(dim, 4 * dim)
        if v2:
            self.grn = GRNWeights(4 * dim)           # registered between the linears, as in convnextv2.py:92-94 (state_dict order)
        self.pwconv2 = LinearWeights(4 * dim, dim)
        if v2:
            pass
        elif init_value > 0:
            self.gamma = nn.Parameter(init_value * torch.ones((dim)), requires_grad=True)
        self.drop_prob = float(dpr)
        self.fp8 = False          # forward products of pwconv1 / pwconv2 on the fp8 matrix pipe (SegmentationModel.set_fp8)

    def tokens(self, x, B, H, W, scale):
        x, xc = Fh.fork(x, 2)                        # residual + depthwise conv both read x
        h = Fh.dwconv7x7(xc, self.dwconv.weight, self.dwconv.bias, B, H, W)
        h = Fh.layer_norm(h, self.norm.weight, self.norm.bias, self.norm.eps)
        h = Fh.linear(h, self.pwconv1.weight, self.pwconv1.bias, fp8=self.fp8)
        if self.v2:
            # act -> grn (convnextv2.py:92-94) as one op: the GELU is applied inside the GRN kernels, gelu(h) is never written
            if Fh.hip.policy('no_gelu_grn'):
                h = Fh.grn(Fh.gelu(h), self.grn.gamma, self.grn.beta, B, H * W)
            else:
                h = Fh.grn(h, self.grn.gamma, self.grn.beta, B, H * W, pre_gelu=True)
            return Fh.linear(h, self.pwconv2.weight, self.pwconv2.bias, residual=x, rscale=scale, rows_per_group=H * W, fp8=self.fp8)
        h = Fh.gelu(h)
        if hasattr(self, 'gamma'):
            return Fh.linear_layer_scale(h, self.pwconv2.weight, self.pwconv2.bias, self.gamma, residual=x, rscale=scale,
                                         rows_per_group=H * W)
        return Fh.linear(h, self.pwconv2.weight, self.pwconv2.bias, residual=x, rscale=scale, rows_per_group=H * W)


class _ConvNeXtBase(nn.Module):
    """Shared trunk of ConvNeXt (convnext.py:79-120) and ConvNeXtV2 (convnextv2.py:116-178): stem conv k4 s4 + channels-first
    LayerNorm, three [LayerNorm + conv k2 s2] downsamplers, four block stages, one output LayerNorm per stage.  The
    channels-first LayerNorm of the reference is the ordinary row LayerNorm on NHWC tokens (eps 1e-6, biased variance)."""

    def _build(self, depths, dims, drop_path_rate, v2):
        self.channels = dims
        self.depths = depths
        self.compute_dtype = torch.bfloat16
        self.stochastic_override = None      # tests: {'drop_path': keep[n_draws, B]}
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.Sequential(ConvWeights(3, dims[0], 4, 4), ChannelsFirstLayerNormWeights(dims[0])))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(ChannelsFirstLayerNormWeights(dims[i]), ConvWeights(dims[i], dims[i + 1], 2, 2)))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.stages = nn.ModuleList()
        cur = 0
        for i in range(4):
            self.stages.append(nn.Sequential(*[ConvNeXtBlock(dims[i], dpr[cur + j], v2=v2) for j in range(depths[i])]))
            cur += depths[i]
        for i in range(4):
            self.add_module(f'norm{i}', ChannelsFirstLayerNormWeights(dims[i]))
        self.apply(_init_convnext)

    def _drop_path_scales(self, B, device):
        """One keep/kp scale row per block with rate > 0 (drop_path.py:18-25; rate-0 blocks are nn.Identity)."""
        rates = [blk.drop_prob for st in self.stages for blk in st]
        if not self.training or all(r == 0 for r in rates):
            return [None] * len(rates)
        draws = [r for r in rates if r > 0]
        if self.stochastic_override is not None and 'drop_path' in self.stochastic_override:
            kp = 1.0 - torch.tensor(draws, dtype=torch.float32, device=device)[:, None]
            scale = (self.stochastic_override['drop_path'].to(device=device, dtype=torch.float32) / kp).contiguous()
        else:
            scale = Fh.stochastic_scales(self, tuple(1.0 - r for r in draws), B, device)
        out, i = [], 0
        for r in rates:
            out.append(scale[i] if r > 0 else None)
            i += 1 if r > 0 else 0
        return out

    def forward_tokens(self, x):
        B, _, H, W = x.shape
        dtype = self.compute_dtype
        scales = self._drop_path_scales(B, x.device)
        outs, bi = [], 0
        t = None
        for i in range(4):
            ds = self.downsample_layers[i]
            if i == 0:
                conv, ln = ds[0], ds[1]
                t = Fh.conv_patch(x, conv.weight, conv.bias, (B, H, W, 3, 4, 4, 0), image=True, dtype=dtype)
                H, W = (H - 4) // 4 + 1, (W - 4) // 4 + 1
                t = Fh.layer_norm(t, ln.weight, ln.bias, ln.eps)
            else:
                ln, conv = ds[0], ds[1]
                t = Fh.layer_norm(t, ln.weight, ln.bias, ln.eps)
                t = Fh.conv_patch(t, conv.weight, conv.bias, (B, H, W, self.channels[i - 1], 2, 2, 0))
                H, W = (H - 2) // 2 + 1, (W - 2) // 2 + 1
            for blk in self.stages[i]:
                t = blk.tokens(t, B, H, W, scales[bi])
                bi += 1
            nrm = getattr(self, f'norm{i}')
            if i < 3:
                t, th = Fh.fork(t, 2)                # the stage output feeds its output norm and the next downsampler's norm
            else:
                th = t
            outs.append(TokenMap(Fh.layer_norm(th, nrm.weight, nrm.bias, nrm.eps), B, H, W))
        return outs

    def forward(self, x):
        return [tm.nchw() for tm in self.forward_tokens(x)]


class ConvNeXt(_ConvNeXtBase):
    """models/backbones/convnext.py:79-120; the reference's plugin API only reaches the default variant 'T'."""

    def __init__(self, model_name: str = 'T') -> None:
        super().__init__()
        assert model_name in convnext_settings.keys(), f"ConvNeXt model name should be in {list(convnext_settings.keys())}"
        depths, embed_dims, drop_path_rate = convnext_settings[model_name]
        self._build(depths, embed_dims, drop_path_rate, v2=False)


class ConvNeXtV2(_ConvNeXtBase):
    """models/backbones/convnextv2.py:116-178 (GRN blocks, no layer scale)."""

    def __init__(self, in_chans=3, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768), drop_path_rate=0.):
        super().__init__()
        assert in_chans == 3
        self._build(list(depths), list(dims), drop_path_rate, v2=True)


# factory functions with the reference's names (incl. its 'convnext_pico' spelling), widths and stochastic-depth rates
# (convnextv2.py:181-233)
def convnextv2_atto(**kw):
    return ConvNeXtV2(depths=[2, 2, 6, 2], dims=[40, 80, 160, 320], drop_path_rate=0.0, **kw)


def convnextv2_femto(**kw):
    return ConvNeXtV2(depths=[2, 2, 6, 2], dims=[48, 96, 192, 384], drop_path_rate=0.0, **kw)


def convnext_pico(**kw):
    return ConvNeXtV2(depths=[2, 2, 6, 2], dims=[64, 128, 256, 512], drop_path_rate=0.0, **kw)


def convnextv2_nano(**kw):
    return ConvNeXtV2(depths=[2, 2, 8, 2], dims=[80, 160, 320, 640], drop_path_rate=0.0, **kw)


def convnextv2_tiny(**kw):
    return ConvNeXtV2(depths=[3, 3, 9, 3], dims=[96, 192, 384, 768], drop_path_rate=0.1, **kw)


def convnextv2_base(**kw):
    return ConvNeXtV2(depths=[3, 3, 27, 3], dims=[128, 256, 512, 1024], drop_path_rate=0.4, **kw)


def convnextv2_large(**kw):
    return ConvNeXtV2(depths=[3, 3, 27, 3], dims=[192, 384, 768, 1536], drop_path_rate=0.5, **kw)


def convnextv2_huge(**kw):
    return ConvNeXtV2(depths=[3, 3, 27, 3], dims=[352, 704, 1408, 2816], drop_path_rate=0.5, **kw)


# ---- MobileNetV2 (models/backbones/mobilenetv2.py) ------------------------------------------------------------------------
def _bn_act(x, bn, training, act):
    """BatchNorm2d (+ReLU6) of mobilenetv2.ConvModule (mobilenetv2.py:5-11) / the bare BatchNorm2d of :29."""
    y = Fh.batch_norm_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum, bn.eps, act=act)
    if training:
        Fh.hip.add_i64_(bn.num_batches_tracked, 1)
    return y


class MBConvModule(nn.Sequential):
    """Conv2d(bias=False) + BatchNorm2d + ReLU6 with the reference's keys `<name>.0.weight`, `<name>.1.*`
    (mobilenetv2.py:5-11; the ReLU6 at index 2 has no parameters)."""

    def __init__(self, c1, c2, k, s=1, p=0, g=1):
        super().__init__(ConvWeights(c1, c2, k, s, p, 1, g, bias=False), BatchNormWeights(c2))
        self.k, self.s, self.g = k, s, g


class InvertedResidual(nn.Module):
    """mobilenetv2.py:14-37: [1x1 expand + BN + ReLU6] -> 3x3 depthwise (stride s) + BN + ReLU6 -> 1x1 project + BN
    (+ residual when s == 1 and c1 == c2)."""

    def __init__(self, c1, c2, s, expand_ratio):
        super().__init__()
        ch = int(round(c1 * expand_ratio))
        self.use_res_connect = s == 1 and c1 == c2
        self.stride, self.ch, self.expand = s, ch, expand_ratio != 1
        layers = []
        if self.expand:
            layers.append(MBConvModule(c1, ch, 1))
        layers.extend([MBConvModule(ch, ch, 3, s, 1, g=ch), ConvWeights(ch, c2, 1, bias=False), BatchNormWeights(c2)])
        self.conv = nn.Sequential(*layers)

    def tokens(self, x, B, H, W, training):
        if self.use_res_connect:
            x, h = Fh.fork(x, 2)
        else:
            h = x
        li = 0
        if self.expand:
            m = self.conv[li]
            h = _bn_act(Fh.linear(h, m[0].weight), m[1], training, 2)
            li += 1
        m = self.conv[li]
        h = Fh.dwconv3x3_gelu(h, m[0].weight, None, B, H, W, False)           # depthwise 3x3, pad 1, no bias, no GELU
        if self.stride > 1:
            h = Fh.subsample(h, B, H, W, self.stride)
            H, W = (H - 1) // self.stride + 1, (W - 1) // self.stride + 1
        h = _bn_act(h, m[1], training, 2)
        h = _bn_act(Fh.linear(h, self.conv[li + 1].weight), self.conv[li + 2], training, 0)
        return (Fh.add(x, h) if self.use_res_connect else h), H, W


class MobileNetV2(nn.Module):
    """models/backbones/mobilenetv2.py:45-92 (width 1.0): Conv-BN-ReLU6 stem + 17 inverted residuals, feature taps after
    features 3, 6, 13, 17 -> channels [24, 32, 96, 320] at strides 4 / 8 / 16 / 32."""

    def __init__(self, variant: str = None):
        super().__init__()
        self.out_indices = [3, 6, 13, 17]
        self.channels = [24, 32, 96, 320]
        self.compute_dtype = torch.bfloat16
        input_channel = 32
        setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]   # t, c, n, s
        self.features = nn.ModuleList([MBConvModule(3, input_channel, 3, 2, 1)])
        for t, c, n, s in setting:
            for i in range(n):
                self.features.append(InvertedResidual(input_channel, c, s if i == 0 else 1, t))
                input_channel = c
        for m in self.modules():                                # mobilenetv2.py:68-79
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward_tokens(self, x):
        B, _, H, W = x.shape
        tr = self.training
        stem = self.features[0]
        t = Fh.conv_patch(x, stem[0].weight, None, (B, H, W, 3, 3, 2, 1), image=True, dtype=self.compute_dtype)
        H, W = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        t = _bn_act(t, stem[1], tr, 2)
        outs = []
        for i in range(1, len(self.features)):
            t, H, W = self.features[i].tokens(t, B, H, W, tr)
            if i in self.out_indices:
                if i != len(self.features) - 1:
                    t, th = Fh.fork(t, 2)            # feature tap: decode head + the next inverted residual
                else:
                    th = t
                outs.append(TokenMap(th, B, H, W))
        return outs

    def forward(self, x):
        return [tm.nchw() for tm in self.forward_tokens(x)]
