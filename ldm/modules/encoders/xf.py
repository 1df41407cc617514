"""The 5-layer one-token mapper behind the CLIP exemplar embedding — class and parameter names of
ldm/modules/encoders/xf.py:22-130 in zhanwenchen/pbe (LayerNorm, MultiheadAttention, MLP,
ResidualAttentionBlock, Transformer).

Paint-by-Example instantiates ``Transformer(n_ctx=1, width=1024, layers=5, heads=1)``
(ldm/modules/encoders/modules.py:144-149): with a single token the softmax over keys is exactly 1,
so ``attn(x) = c_proj(V)`` where V is the last third of ``c_qkv(ln_1(x))`` (xf.py:61-77, heads = 1).
Each block is therefore LayerNorm -> GEMM(V rows of c_qkv) -> GEMM(c_proj)+residual ->
LayerNorm -> GEMM(c_fc)+GELU -> GEMM(c_proj)+residual, all HIP kernels with fused epilogues."""
from types import SimpleNamespace

import torch
from torch import nn

from pbe_amd import ops
from pbe_amd.hipmodule import HipModule, f32, require_gpu
from pbe_amd.lib import PbeError


class LayerNorm(nn.LayerNorm):
    """fp32 statistics regardless of the input dtype (xf.py:22-28)."""

    def forward(self, x):
        require_gpu(x, "xf.LayerNorm")
        y = ops.layernorm(x.to(torch.float16).contiguous(), self.weight.float(), self.bias.float(), self.eps)
        return y.to(x.dtype)


class QKVMultiheadAttention(nn.Module):
    def __init__(self, n_heads, n_ctx):
        super().__init__()
        self.n_heads, self.n_ctx = n_heads, n_ctx


class MultiheadAttention(nn.Module):
    def __init__(self, n_ctx, width, heads):
        super().__init__()
        self.n_ctx, self.width, self.heads = n_ctx, width, heads
        self.c_qkv = nn.Linear(width, width * 3)
        self.c_proj = nn.Linear(width, width)
        self.attention = QKVMultiheadAttention(heads, n_ctx)


class MLP(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.width = width
        self.c_fc = nn.Linear(width, width * 4)
        self.c_proj = nn.Linear(width * 4, width)
        self.gelu = nn.GELU()


class ResidualAttentionBlock(HipModule):
    def __init__(self, n_ctx, width, heads):
        super().__init__()
        if n_ctx != 1 or heads != 1:
            raise PbeError("xf.ResidualAttentionBlock: Paint-by-Example's mapper is n_ctx = 1, heads = 1")
        self.attn = MultiheadAttention(n_ctx, width, heads)
        self.ln_1 = LayerNorm(width)
        self.mlp = MLP(width)
        self.ln_2 = LayerNorm(width)

    def _pack(self):
        w = self.attn.width
        return SimpleNamespace(g1=f32(self.ln_1.weight), b1=f32(self.ln_1.bias), g2=f32(self.ln_2.weight), b2=f32(self.ln_2.bias),
                               eps=self.ln_1.eps, wv=ops.pack_linear(self.attn.c_qkv.weight[2 * w:]), bv=f32(self.attn.c_qkv.bias[2 * w:]),
                               wp=ops.pack_linear(self.attn.c_proj.weight), bp=f32(self.attn.c_proj.bias),
                               wf=ops.pack_linear(self.mlp.c_fc.weight), bf=f32(self.mlp.c_fc.bias),
                               wo=ops.pack_linear(self.mlp.c_proj.weight), bo=f32(self.mlp.c_proj.bias))

    def run(self, x):
        """x [B, width] fp16."""
        p = self.pk()
        v = ops.gemm(ops.layernorm(x, p.g1, p.b1, p.eps), p.wv, p.bv)
        x = ops.gemm(v, p.wp, p.bp, resid=x)
        h = ops.gemm(ops.layernorm(x, p.g2, p.b2, p.eps), p.wf, p.bf, act=ops.ACT_GELU)
        return ops.gemm(h, p.wo, p.bo, resid=x)

    def forward(self, x):
        require_gpu(x, "xf.ResidualAttentionBlock")
        B, n, w = x.shape
        return self.run(x.to(torch.float16).reshape(B * n, w).contiguous()).view(B, n, w).to(x.dtype)


class Transformer(HipModule):
    def __init__(self, n_ctx, width, layers, heads):
        super().__init__()
        self.n_ctx, self.width, self.layers = n_ctx, width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(n_ctx, width, heads) for _ in range(layers)])

    def run(self, x):
        for blk in self.resblocks:
            x = blk.run(x)
        return x

    def forward(self, x):
        require_gpu(x, "xf.Transformer")
        B, n, w = x.shape
        return self.run(x.to(torch.float16).reshape(B * n, w).contiguous()).view(B, n, w).to(x.dtype)
