"""Frozen CLIP ViT-L/14 image encoder + mapper of Paint-by-Example on MI355X.

``FrozenCLIPImageEmbedder`` keeps the constructor, ``encode`` / ``forward`` and the state_dict keys
of ldm/modules/encoders/modules.py:138-171 in zhanwenchen/pbe:
``transformer.vision_model.*`` (Hugging Face CLIPVisionModel, transformers==4.19.2 naming),
``mapper.resblocks.N.*`` and ``final_ln.*``.

The reference fetches the tower with ``CLIPVisionModel.from_pretrained("openai/clip-vit-large-patch14")``
(modules.py:142); that arithmetic lives in the third-party `transformers` package, not in the
reference tree.  Here the tower is built from the published ViT-L/14 geometry (no hub access, the
weights come from the Paint-by-Example checkpoint like every other tensor) and runs as HIP
kernels: patch conv = patchify + GEMM with the position embedding added in the epilogue, fused
q|k projection, V^T projection, flash attention over 257 tokens (16 heads x 64), quick-GELU and
residuals as GEMM epilogues, pooled output = post_layernorm(CLS).
"""
from types import SimpleNamespace

import torch
from torch import nn

from pbe_amd import ops
from pbe_amd.hipmodule import HipModule, f32, require_gpu
from ldm.modules.encoders.xf import LayerNorm, Transformer

CLIP_VIT_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224,
                    patch_size=14, layer_norm_eps=1e-5)


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        n = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
        self.class_embedding = nn.Parameter(torch.randn(cfg["hidden_size"]))
        self.patch_embedding = nn.Conv2d(3, cfg["hidden_size"], kernel_size=cfg["patch_size"], stride=cfg["patch_size"], bias=False)
        self.position_embedding = nn.Embedding(n, cfg["hidden_size"])
        self.register_buffer("position_ids", torch.arange(n).expand((1, -1)).clone())


class _SelfAttn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)


class _Mlp(nn.Module):
    def __init__(self, d, inner):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(d, inner), nn.Linear(inner, d)


class _EncoderLayer(HipModule):
    def __init__(self, cfg):
        super().__init__()
        d = cfg["hidden_size"]
        self.heads, self.eps = cfg["num_attention_heads"], cfg["layer_norm_eps"]
        self.self_attn = _SelfAttn(d)
        self.layer_norm1 = nn.LayerNorm(d, eps=self.eps)
        self.mlp = _Mlp(d, cfg["intermediate_size"])
        self.layer_norm2 = nn.LayerNorm(d, eps=self.eps)

    def _pack(self):
        a, m = self.self_attn, self.mlp
        return SimpleNamespace(g1=f32(self.layer_norm1.weight), b1=f32(self.layer_norm1.bias), g2=f32(self.layer_norm2.weight),
                               b2=f32(self.layer_norm2.bias),
                               wqk=ops.pack_linear(torch.cat([a.q_proj.weight, a.k_proj.weight], 0)), bqk=f32(torch.cat([a.q_proj.bias, a.k_proj.bias], 0)),
                               wv=ops.pack_linear(a.v_proj.weight), bv=f32(a.v_proj.bias), wo=ops.pack_linear(a.out_proj.weight), bo=f32(a.out_proj.bias),
                               w1=ops.pack_linear(m.fc1.weight), c1=f32(m.fc1.bias), w2=ops.pack_linear(m.fc2.weight), c2=f32(m.fc2.bias))

    def run(self, x, B, N, vt):
        """x [B*N, d] fp16 residual stream; vt = scratch [B, d, round8(N)] for V^T."""
        p = self.pk()
        d = x.shape[1]
        dh = d // self.heads
        npad = vt.shape[2]
        h = ops.layernorm(x, p.g1, p.b1, self.eps)
        qk = ops.gemm(h, p.wqk, p.bqk)
        ops.gemm(p.wv.unsqueeze(0).expand(B, -1, -1), h.view(B, N, d), p.bv, bias_per_row=True, out=vt[:, :, :N] if npad != N else vt)
        o = ops.attention(qk, qk[:, d:], vt, B, self.heads, N, N, dh, dh ** -0.5, q_strides=(N * 2 * d, 2 * d), k_strides=(N * 2 * d, 2 * d),
                          vt_strides=(d * npad, npad))
        x = ops.gemm(o.view(B * N, d), p.wo, p.bo, resid=x)
        h = ops.gemm(ops.layernorm(x, p.g2, p.b2, self.eps), p.w1, p.c1, act=ops.ACT_QUICK_GELU)
        return ops.gemm(h, p.w2, p.c2, resid=x)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(cfg) for _ in range(cfg["num_hidden_layers"])])


class _VisionTransformer(HipModule):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = dict(cfg)
        d = cfg["hidden_size"]
        self.embeddings = _Embeddings(cfg)
        self.pre_layrnorm = nn.LayerNorm(d, eps=cfg["layer_norm_eps"])        # (sic) Hugging Face's spelling
        self.encoder = _Encoder(cfg)
        self.post_layernorm = nn.LayerNorm(d, eps=cfg["layer_norm_eps"])

    def _pack(self):
        e, cfg = self.embeddings, self.cfg
        k = 3 * cfg["patch_size"] ** 2
        kp = (k + 7) // 8 * 8
        wp = torch.zeros(cfg["hidden_size"], kp)
        wp[:, :k] = e.patch_embedding.weight.detach().reshape(cfg["hidden_size"], k).float().cpu()
        dev = e.class_embedding.device
        return SimpleNamespace(kp=kp, wpatch=wp.to(torch.float16).to(dev), cls=e.class_embedding.detach().to(torch.float16).contiguous(),
                               pos=e.position_embedding.weight.detach().to(torch.float16).contiguous(),
                               gpre=f32(self.pre_layrnorm.weight), bpre=f32(self.pre_layrnorm.bias),
                               gpost=f32(self.post_layernorm.weight), bpost=f32(self.post_layernorm.bias))

    def pooled(self, pixels):
        """pixels fp32 [B,3,S,S] (CLIP-normalised) -> pooler_output [B, hidden] fp16."""
        p, cfg = self.pk(), self.cfg
        B = pixels.shape[0]
        d, P, eps = cfg["hidden_size"], cfg["patch_size"], cfg["layer_norm_eps"]
        G = cfg["image_size"] // P
        N = G * G + 1
        patches = ops.clip_patchify(pixels, P, p.kp)                                  # [B*G*G, kp]
        tokens = torch.empty((B, N, d), dtype=torch.float16, device=pixels.device)
        ops.gemm(patches.view(B, G * G, p.kp), p.wpatch.unsqueeze(0), resid=p.pos[1:], out=tokens[:, 1:, :])
        ops.bcast_row(p.cls, p.pos[0], tokens, B, N * d)
        x = ops.layernorm(tokens.view(B * N, d), p.gpre, p.bpre, eps)
        vt = torch.zeros((B, d, (N + 7) // 8 * 8), dtype=torch.float16, device=pixels.device)
        for layer in self.encoder.layers:
            x = layer.run(x, B, N, vt)
        return ops.layernorm(x.view(B, N, d)[:, 0, :], p.gpost, p.bpost, eps)


class CLIPVisionTower(nn.Module):
    """Stands where the reference holds ``CLIPVisionModel`` (attribute ``transformer``): parameters
    live under ``vision_model.*`` exactly as in the Paint-by-Example checkpoint."""

    def __init__(self, cfg=None):
        super().__init__()
        self.vision_model = _VisionTransformer(cfg or CLIP_VIT_L14)

    def forward(self, pixel_values):
        return SimpleNamespace(pooler_output=self.vision_model.pooled(pixel_values.float()))


class FrozenCLIPImageEmbedder(AbstractEncoder):
    """CLIP image tower -> pooled [B,1024] -> unsqueeze(1) -> 5-layer one-token mapper -> final LayerNorm."""

    def __init__(self, version="openai/clip-vit-large-patch14", clip_config=None, mapper_layers=5):
        super().__init__()
        cfg = clip_config or CLIP_VIT_L14
        self.transformer = CLIPVisionTower(cfg)
        self.final_ln = LayerNorm(cfg["hidden_size"])
        self.mapper = Transformer(1, cfg["hidden_size"], mapper_layers, 1)
        self.freeze()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for param in self.parameters():
            param.requires_grad = False

    def forward(self, image):
        require_gpu(image, "FrozenCLIPImageEmbedder")
        z = self.transformer(pixel_values=image).pooler_output                 # [B, hidden] fp16
        z = self.mapper.run(z)
        z = ops.layernorm(z, self.final_ln.weight.float(), self.final_ln.bias.float(), self.final_ln.eps)
        return z.unsqueeze(1)                                                  # [B, 1, hidden]

    def encode(self, image):
        return self(image)
