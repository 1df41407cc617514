"""AutoencoderKL (f = 8, z = 4) first stage — API and state_dict keys of ldm/models/autoencoder.py
:19-69 in zhanwenchen/pbe (encoder.*, decoder.*, quant_conv.*, post_quant_conv.*); inference
only (the Lightning training/validation/logging methods :80-162 are out of scope)."""
from types import SimpleNamespace

import torch
from torch import nn

from pbe_amd import ops
from pbe_amd.hipmodule import HipModule, f32, require_gpu
from ldm.modules.diffusionmodules.model import Decoder, Encoder
from ldm.modules.distributions.distributions import DiagonalGaussianDistribution


class AutoencoderKL(HipModule):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None):
        super().__init__()
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        assert ddconfig["double_z"]
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        if monitor is not None:
            self.monitor = monitor
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=()):
        from pbe_amd.checkpoint import read_state_dict
        sd = read_state_dict(path)
        sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
        self.load_state_dict(sd, strict=False)

    def _pack(self):
        zc = self.post_quant_conv.weight.shape[1]
        wpq = torch.zeros(self.post_quant_conv.weight.shape[0], 8)
        wpq[:, :zc] = self.post_quant_conv.weight.detach().reshape(-1, zc).float().cpu()
        return SimpleNamespace(wq=ops.pack_linear(self.quant_conv.weight), bq=f32(self.quant_conv.bias),
                               wpq=wpq.to(torch.float16).to(self.post_quant_conv.weight.device), bpq=f32(self.post_quant_conv.bias))

    # ---- NHWC fast paths ------------------------------------------------------------------------
    def encode_nhwc(self, x_nhwc):
        """x [B,H,W,cin_pad] fp16 -> moments [B,h,w,2*embed] fp16."""
        p = self.pk()
        h = self.encoder.run(x_nhwc)
        B, hh, ww, Cc = h.shape
        return ops.gemm(h.view(B * hh * ww, Cc), p.wq, p.bq).view(B, hh, ww, -1)

    def decode_nhwc(self, z_nhwc8):
        """z [B,h,w,8] fp16 (first embed_dim channels real) -> image [B,8h,8w,out_ch] fp16."""
        p = self.pk()
        B, hh, ww, Cc = z_nhwc8.shape
        zc = p.wpq.shape[0]
        zp = torch.zeros((B * hh * ww, self.decoder.pk().cin_pad), dtype=torch.float16, device=z_nhwc8.device)
        ops.gemm(z_nhwc8.view(B * hh * ww, Cc), p.wpq, p.bpq, out=zp[:, :zc])   # post_quant_conv into the padded decoder input
        return self.decoder.run(zp.view(B, hh, ww, -1))

    # ---- reference-shaped API (NCHW) -------------------------------------------------------------
    def encode(self, x) -> DiagonalGaussianDistribution:
        require_gpu(x, "AutoencoderKL.encode")
        return DiagonalGaussianDistribution(self.encode_nhwc(ops.nchw_to_nhwc(x.float(), self.encoder.pk().cin_pad)))

    def decode(self, z):
        require_gpu(z, "AutoencoderKL.decode")
        return ops.nhwc_to_nchw(self.decode_nhwc(ops.nchw_to_nhwc(z.float(), 8))).to(z.dtype)

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior
