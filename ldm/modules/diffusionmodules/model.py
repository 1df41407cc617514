"""AutoencoderKL encoder / decoder (f = 8) on MI355X — class names, constructor arguments and
parameter names of ldm/modules/diffusionmodules/model.py in zhanwenchen/pbe (ResnetBlock :84-143,
AttnBlock :152-204, Downsample :62-81, Upsample :44-59, Encoder :370-471, Decoder :474-580),
executed as HIP kernels over NHWC fp16:

  * convs are matrix-core implicit GEMMs (nearest-2x upsample folded into the gather; the
    encoder's asymmetric (0,1,0,1) pad + stride-2 conv is the gather's pad=0 mode),
  * GroupNorm(eps 1e-6)+swish is the fused two-pass kernel,
  * the mid-block attention is ONE head of d = 512 over N = HW tokens (model.py:180-204): too wide
    for the register-resident flash kernel, and it runs once per image, so it is two batched GEMMs
    around a row-softmax kernel with fp16 scores in HBM (32 MiB per 512x512 sample).
"""
from types import SimpleNamespace

import torch
from torch import nn

from pbe_amd import ops
from pbe_amd.hipmodule import HipModule, f32, require_gpu
from pbe_amd.lib import PbeError


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


def _pad8(c):
    return (c + 7) // 8 * 8


def _conv_any(x, conv_pack, **kw):
    """3x3 conv choosing the implicit-GEMM gather (Cin % 64 == 0) or im2col+GEMM (tiny Cin)."""
    w, b, small = conv_pack
    if small:
        return ops.conv3x3_small(x, w, b, **kw)
    return ops.conv3x3(x, w, b, **kw)


def _pack_conv(conv):
    cin = conv.weight.shape[1]
    if cin % 64 == 0:
        return ops.pack_conv3x3(conv.weight), f32(conv.bias), False
    return ops.pack_conv3x3(conv.weight, _pad8(cin)), f32(conv.bias), True


class Upsample(HipModule):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise PbeError("VAE Upsample without conv is not on the Paint-by-Example path")
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def _pack(self):
        # interpolate(nearest, 2x) + conv3x3 (model.py:44-53) as four 2x2 convs on the source grid (ops.pack_conv3x3_up_phases): 4 / 9 of the MACs
        return SimpleNamespace(c=(ops.pack_conv3x3_up_phases(self.conv.weight), f32(self.conv.bias)))

    def run(self, x):
        return ops.conv3x3(x, self.pk().c[0], self.pk().c[1], upsample=True)


class Downsample(HipModule):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise PbeError("VAE Downsample without conv is not on the Paint-by-Example path")
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def _pack(self):
        return SimpleNamespace(c=_pack_conv(self.conv))

    def run(self, x):
        return _conv_any(x, self.pk().c, stride=2, pad=0)      # == pad (0,1,0,1) then conv s2 p0 (model.py:74-78)


class ResnetBlock(HipModule):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        if conv_shortcut or temb_channels > 0:
            raise PbeError("VAE ResnetBlock: conv_shortcut / temb are not used by AutoencoderKL (temb_ch = 0)")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.use_conv_shortcut = in_channels, out_channels, conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def _pack(self):
        ns = SimpleNamespace(g1=f32(self.norm1.weight), b1=f32(self.norm1.bias), g2=f32(self.norm2.weight), b2=f32(self.norm2.bias),
                             eps=self.norm1.eps, c1=_pack_conv(self.conv1), c2=_pack_conv(self.conv2), ws=None)
        if self.in_channels != self.out_channels:
            ns.ws, ns.bs = ops.pack_linear(self.nin_shortcut.weight), f32(self.nin_shortcut.bias)
        return ns

    def run(self, x):
        p = self.pk()
        B, H, W, Cc = x.shape
        h = _conv_any(ops.groupnorm(x, p.g1, p.b1, p.eps, True), p.c1)
        h = ops.groupnorm(h, p.g2, p.b2, p.eps, True)
        xs = x if p.ws is None else ops.gemm(x.view(B * H * W, Cc), p.ws, p.bs).view(B, H, W, -1)
        return _conv_any(h, p.c2, resid=xs)


class AttnBlock(HipModule):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)

    def _pack(self):
        return SimpleNamespace(g=f32(self.norm.weight), b=f32(self.norm.bias), eps=self.norm.eps,
                               wqk=ops.pack_linear(torch.cat([self.q.weight, self.k.weight], 0)), bqk=f32(torch.cat([self.q.bias, self.k.bias], 0)),
                               wv=ops.pack_linear(self.v.weight), bv=f32(self.v.bias),
                               wo=ops.pack_linear(self.proj_out.weight), bo=f32(self.proj_out.bias))

    def run(self, x):
        p = self.pk()
        B, H, W, Cc = x.shape
        N = H * W
        if N % 8:
            raise PbeError(f"VAE AttnBlock: HW = {N} must be a multiple of 8")
        hn = ops.groupnorm(x, p.g, p.b, p.eps, False).view(B * N, Cc)
        qk = ops.gemm(hn, p.wqk, p.bqk).view(B, N, 2 * Cc)
        vt = ops.gemm(p.wv.unsqueeze(0).expand(B, -1, -1), hn.view(B, N, Cc), p.bv, bias_per_row=True)      # [B, C, N] = V^T
        s = ops.gemm(qk[:, :, :Cc], qk[:, :, Cc:])                                                            # [B, Nq, Nk] raw scores
        pr = ops.softmax_rows(s, float(int(Cc) ** -0.5))
        o = ops.gemm(pr, vt)                                                                                  # [B, N, C]
        return ops.gemm(o.view(B * N, Cc), p.wo, p.bo, resid=x.view(B * N, Cc)).view(B, H, W, Cc)


def make_attn(in_channels, attn_type="vanilla"):
    if attn_type == "vanilla":
        return AttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity(in_channels)
    raise PbeError(f"attn_type {attn_type} is not on the Paint-by-Example path")


class Encoder(HipModule):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, double_z=True, use_linear_attn=False, attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if use_linear_attn or len(attn_resolutions):
            raise PbeError("VAE Encoder: per-level / linear attention is not configured by configs/v1.yaml (attn_resolutions: [])")
        self.ch, self.temb_ch, self.num_resolutions = ch, 0, len(ch_mult)
        self.num_res_blocks, self.resolution, self.in_channels = num_res_blocks, resolution, in_channels
        self.conv_in = nn.Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for lvl in range(self.num_resolutions):
            block_in, block_out = ch * in_ch_mult[lvl], ch * ch_mult[lvl]
            stage = nn.Module()
            stage.block = nn.ModuleList()
            stage.attn = nn.ModuleList()
            for _ in range(num_res_blocks):
                stage.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            if lvl != self.num_resolutions - 1:
                stage.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(stage)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)

    def _pack(self):
        return SimpleNamespace(cin=_pack_conv(self.conv_in), cin_pad=_pad8(self.in_channels), g=f32(self.norm_out.weight),
                               b=f32(self.norm_out.bias), eps=self.norm_out.eps, cout=_pack_conv(self.conv_out))

    def run(self, x):
        """x [B,H,W,cin_pad] fp16 -> [B,H/8,W/8,2*z] fp16."""
        p = self.pk()
        h = _conv_any(x, p.cin)
        for lvl, stage in enumerate(self.down):
            for blk in stage.block:
                h = blk.run(h)
            if lvl != self.num_resolutions - 1:
                h = stage.downsample.run(h)
        h = self.mid.block_2.run(self.mid.attn_1.run(self.mid.block_1.run(h)))
        return _conv_any(ops.groupnorm(h, p.g, p.b, p.eps, True), p.cout)

    def forward(self, x):
        require_gpu(x, "Encoder")
        return ops.nhwc_to_nchw(self.run(ops.nchw_to_nhwc(x.float(), self.pk().cin_pad))).to(x.dtype)


class Decoder(HipModule):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False, use_linear_attn=False, attn_type="vanilla",
                 **ignorekwargs):
        super().__init__()
        if use_linear_attn or len(attn_resolutions) or give_pre_end or tanh_out:
            raise PbeError("VAE Decoder: attention levels / pre-end / tanh output are not configured by configs/v1.yaml")
        self.ch, self.temb_ch, self.num_resolutions = ch, 0, len(ch_mult)
        self.num_res_blocks, self.resolution, self.in_channels = num_res_blocks, resolution, in_channels
        self.z_channels = z_channels
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for lvl in reversed(range(self.num_resolutions)):
            block_out = ch * ch_mult[lvl]
            stage = nn.Module()
            stage.block = nn.ModuleList()
            stage.attn = nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                stage.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            if lvl != 0:
                stage.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, stage)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def _pack(self):
        return SimpleNamespace(cin=_pack_conv(self.conv_in), cin_pad=_pad8(self.z_channels), g=f32(self.norm_out.weight),
                               b=f32(self.norm_out.bias), eps=self.norm_out.eps, cout=_pack_conv(self.conv_out))

    def run(self, z):
        """z [B,h,w,cin_pad] fp16 -> [B,8h,8w,out_ch] fp16."""
        p = self.pk()
        h = _conv_any(z, p.cin)
        h = self.mid.block_2.run(self.mid.attn_1.run(self.mid.block_1.run(h)))
        for lvl in reversed(range(self.num_resolutions)):
            for blk in self.up[lvl].block:
                h = blk.run(h)
            if lvl != 0:
                h = self.up[lvl].upsample.run(h)
        return _conv_any(ops.groupnorm(h, p.g, p.b, p.eps, True), p.cout)

    def forward(self, z):
        require_gpu(z, "Decoder")
        return ops.nhwc_to_nchw(self.run(ops.nchw_to_nhwc(z.float(), self.pk().cin_pad))).to(z.dtype)
