"""Transformer pieces of the U-Net, same class names / constructor arguments / parameter names as
ldm/modules/attention.py in zhanwenchen/pbe (GEGLU :38-45, FeedForward :48-65, CrossAttention
:189-230, BasicTransformerBlock :233-252, SpatialTransformer :255-298), executed by HIP kernels:

  * activations stay ``[B, N, C]`` fp16 (tokens x channels == NHWC, so the reference's two
    rearranges per block are free),
  * to_q | to_k run as ONE GEMM, to_v as a second GEMM with swapped operands that emits V^T
    (what the fused attention kernel consumes), softmax(QK^T)V never touches HBM,
  * the Paint-by-Example context is ONE token per sample, so attn2's softmax over a single key
    is exactly 1 and ``attn2(x, ctx) = to_out(to_v(ctx))`` for every query (attention.py:207-230;
    SURVEY.md K6): it is computed once per context as a [B, C] vector and added inside the
    epilogue of attn1's output projection.  norm2 / attn2.to_q / attn2.to_k keep their
    parameters (checkpoint compatibility) but are dead arithmetic on this path,
  * bias, residual and the row-broadcast adds are GEMM epilogues.
"""
from types import SimpleNamespace

import torch
from torch import nn

from pbe_amd import ops
from pbe_amd.hipmodule import HipModule, f32, require_gpu
from pbe_amd.lib import PbeError


def exists(val):
    return val is not None


def default(val, d):
    return val if val is not None else (d() if callable(d) else d)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def Normalize(in_channels):
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


def _tokens(x):
    """[B, N, C] any float dtype -> contiguous fp16 on the GPU."""
    require_gpu(x, "attention")
    return x.to(torch.float16).contiguous()


class GEGLU(HipModule):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def _pack(self):
        wi, bi = ops.pack_geglu(self.proj.weight.detach(), self.proj.bias.detach())
        return SimpleNamespace(w=wi, b=bi, f8=None)          # (value, gate) rows interleaved: gating fused in the GEMM epilogue

    def run(self, x2d):
        p = self.pk()
        return ops.gemm(x2d, p.w, p.b, act=ops.ACT_GEGLU)

    def run_f8(self, x8, sx):
        """x8 / sx: the fp8 LayerNorm output and its row scales -> same result shape as run(); weights e4m3 with a scale per (interleaved) row."""
        p = self.pk()
        if p.f8 is None:
            p.f8 = ops.pack_linear_f8(p.w.float())
        return ops.gemm_f8(x8, sx, p.f8[0], p.f8[1], p.b, act=ops.ACT_GEGLU)

    def forward(self, x):
        x = _tokens(x)
        return self.run(x.view(-1, x.shape[-1])).view(*x.shape[:-1], -1)


class FeedForward(HipModule):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        inner = int(dim * mult)
        dim_out = default(dim_out, dim)
        if not glu:
            raise PbeError("FeedForward(glu=False) is not on the Paint-by-Example path (BasicTransformerBlock uses gated_ff=True)")
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out))

    def _pack(self):
        return SimpleNamespace(w2=ops.pack_linear(self.net[2].weight), b2=f32(self.net[2].bias))

    def run(self, x2d, resid=None):
        """x2d [M, C] fp16 -> Linear(GEGLU(x)) (+ resid)."""
        p = self.pk()
        return ops.gemm(self.net[0].run(x2d), p.w2, p.b2, resid=resid)

    def run_f8(self, x8, sx, resid=None):
        """fp8 operands for the GEGLU projection (its input comes from the fp8 LayerNorm); the output projection stays fp16: its input
        is the GEGLU product, whose per-row scale would need a pass over all 4C columns of a row."""
        p = self.pk()
        return ops.gemm(self.net[0].run_f8(x8, sx), p.w2, p.b2, resid=resid)

    def forward(self, x):
        x = _tokens(x)
        return self.run(x.view(-1, x.shape[-1])).view(x.shape[0], x.shape[1], -1)


class CrossAttention(HipModule):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner = dim_head * heads
        self.is_self = context_dim is None
        context_dim = default(context_dim, query_dim)
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))

    def _pack(self):
        ns = SimpleNamespace(wv=ops.pack_linear(self.to_v.weight), wo=ops.pack_linear(self.to_out[0].weight), bo=f32(self.to_out[0].bias))
        if self.to_q.weight.shape[1] == self.to_k.weight.shape[1]:
            ns.wqk = ops.pack_linear(torch.cat([self.to_q.weight, self.to_k.weight], 0))
        ns.wq, ns.wk = ops.pack_linear(self.to_q.weight), ops.pack_linear(self.to_k.weight)
        ns.f8 = None
        return ns

    def self_attention_f8(self, x8, sx, B, N):
        """self_attention with fp8 (e4m3) operands for the q|k and V^T projections: x8 [B*N, C] uint8 + row scales sx [B*N] from
        pbe_layernorm_f8, weights e4m3 with one scale per output channel.  The attention core itself stays fp16 (fp32 softmax)."""
        p = self.pk()
        if p.f8 is None:
            p.f8 = (ops.pack_linear_f8(torch.cat([self.to_q.weight, self.to_k.weight], 0)), ops.pack_linear_f8(self.to_v.weight))
        (wqk8, sqk), (wv8, sv) = p.f8
        inner = self.heads * self.dim_head
        qk = ops.gemm_f8(x8, sx, wqk8, sqk)
        npad = (N + 7) // 8 * 8
        vt = torch.empty((B, inner, npad), dtype=torch.float16, device=x8.device)
        ops.gemm_f8(wv8.unsqueeze(0).expand(B, -1, -1), sv, x8.view(B, N, -1), sx.view(B, N), out=vt[:, :, :N] if npad != N else vt)
        o = ops.attention(qk, qk[:, inner:], vt, B, self.heads, N, N, self.dim_head, self.scale,
                          q_strides=(N * 2 * inner, 2 * inner), k_strides=(N * 2 * inner, 2 * inner), vt_strides=(inner * npad, npad))
        return o.view(B * N, inner)

    # ---- fast paths used by BasicTransformerBlock -------------------------------------------
    def self_attention(self, xn, B, N):
        """xn [B*N, C] (already normed) -> per-head softmax(QK^T)V as [B*N, inner] (before to_out)."""
        p = self.pk()
        inner = self.heads * self.dim_head
        qk = ops.gemm(xn, p.wqk)                                              # [M, 2*inner]
        npad = (N + 7) // 8 * 8
        vt = torch.empty((B, inner, npad), dtype=torch.float16, device=xn.device)
        ops.gemm(p.wv.unsqueeze(0).expand(B, -1, -1), xn.view(B, N, -1), out=vt[:, :, :N] if npad != N else vt)
        o = ops.attention(qk, qk[:, inner:], vt, B, self.heads, N, N, self.dim_head, self.scale,
                          q_strides=(N * 2 * inner, 2 * inner), k_strides=(N * 2 * inner, 2 * inner), vt_strides=(inner * npad, npad))
        return o.view(B * N, inner)

    def self_attention_fused(self, x2d, stats, bp, B, N):
        """LayerNorm + to_q | to_k | to_v + attention core with ONE projection launch: x2d [B*N, C] is the RAW residual stream, `stats` its
        row statistics (from the producer's epilogue), bp the block's pack (weights pre-multiplied by norm1's gain, attention.py:248).
        The GEMM folds the LayerNorm into its epilogue, multiplies the q columns by scale log2(e) in fp32 (the attention kernel then
        runs exp2 on the MFMA output directly) and stores the v columns transposed (V^T, what the attention kernel streams)."""
        inner = self.heads * self.dim_head
        qk = torch.empty((B * N, 2 * inner), dtype=torch.float16, device=x2d.device)
        vt = torch.empty((B, inner, N), dtype=torch.float16, device=x2d.device)
        ops.gemm(x2d, bp.wqkv, bp.c2qkv, ln=(stats, bp.c1qkv, bp.eps1), alpha=bp.qscale, alpha_cols=inner, out=qk, vt=vt, vt_col0=2 * inner, vt_tokens=N)
        o = ops.attention(qk, qk[:, inner:], vt, B, self.heads, N, N, self.dim_head, self.scale, q_strides=(N * 2 * inner, 2 * inner),
                          k_strides=(N * 2 * inner, 2 * inner), vt_strides=(inner * N, N), q_prescaled=True)
        return o.view(B * N, inner)

    def single_token_context(self, context):
        """context [B, 1, Dc] -> to_out(to_v(context)) as [B, C] fp16 (softmax over one key == 1)."""
        p = self.pk()
        c = _tokens(context)
        if c.shape[1] != 1:
            raise PbeError(f"CrossAttention: Paint-by-Example conditions on ONE exemplar token, got {c.shape[1]}")
        v = ops.gemm(c.view(c.shape[0], -1), p.wv)
        return ops.gemm(v, p.wo, p.bo)

    # ---- reference-shaped entry point ------------------------------------------------------------
    def forward(self, x, context=None, mask=None):
        if exists(mask):
            raise PbeError("CrossAttention: masks are not used on the Paint-by-Example path")
        x = _tokens(x)
        B, N, _ = x.shape
        p = self.pk()
        if context is None:
            o = self.self_attention(x.view(B * N, -1), B, N)
            return ops.gemm(o, p.wo, p.bo).view(B, N, -1)
        c = _tokens(context)
        if c.shape[1] == 1:
            return self.single_token_context(c)[:, None, :].expand(B, N, -1).contiguous()
        # general context length (not on the hot path): separate q / k / v projections
        inner, Nk = self.heads * self.dim_head, c.shape[1]
        q = ops.gemm(x.view(B * N, -1), p.wq)
        k = ops.gemm(c.view(B * Nk, -1), p.wk)
        npad = (Nk + 7) // 8 * 8
        vt = torch.zeros((B, inner, npad), dtype=torch.float16, device=x.device)
        ops.gemm(p.wv.unsqueeze(0).expand(B, -1, -1), c, out=vt[:, :, :Nk] if npad != Nk else vt)
        o = ops.attention(q, k, vt, B, self.heads, N, Nk, self.dim_head, self.scale, q_strides=(N * inner, inner),
                          k_strides=(Nk * inner, inner), vt_strides=(inner * npad, npad))
        return ops.gemm(o.view(B * N, inner), p.wo, p.bo).view(B, N, -1)


class BasicTransformerBlock(HipModule):
    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint

    def _pack(self):
        """Both LayerNorms are FOLDED into the GEMMs that read them (pbe_gemm_desc.ln_stats): norm1 into ONE q | k | v projection, norm3 into
        the GEGLU projection - the weights carry the gain, the bias carries W beta, the row statistics come from the producer's epilogue."""
        a, n1, n3 = self.attn1, self.norm1, self.norm3
        ns = SimpleNamespace(g1=f32(n1.weight), b1=f32(n1.bias), g3=f32(n3.weight), b3=f32(n3.bias), eps1=n1.eps, eps3=n3.eps, wqkv=None)
        inner = a.heads * a.dim_head
        if a.to_q.weight.shape[1] == a.to_k.weight.shape[1] == a.to_v.weight.shape[1]:
            ns.qscale = a.scale * 1.4426950408889634             # q leaves the projection as scale log2(e) q (fp32 epilogue)
            ns.wqkv, ns.c2qkv, ns.c1qkv = ops.pack_linear_ln(torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0), None, n1.weight, n1.bias)
            ns.c2qkv[:inner] *= ns.qscale                         # (alpha multiplies the product, the bias is added after it)
        w, b = self.ff.net[0].proj.weight.detach().float(), self.ff.net[0].proj.bias.detach().float()
        F = w.shape[0] // 2
        wi, bi = torch.stack([w[:F], w[F:]], 1).reshape(2 * F, -1), torch.stack([b[:F], b[F:]], 1).reshape(2 * F)   # (value, gate) rows interleaved
        ns.wg, ns.c2g, ns.c1g = ops.pack_linear_ln(wi, bi, n3.weight, n3.bias)
        return ns

    linear_fp8 = False          # pbe_amd.precision.set_linear_precision(model, "fp8") turns the LayerNorm-fed projections to e4m3 operands
    fold_layernorm = True       # False: the separate LayerNorm launches of rounds 1-2 (A/B runs, tools/)

    def _folded(self, p, N):
        return self.fold_layernorm and not self.linear_fp8 and p.wqkv is not None and N % 8 == 0

    def run(self, x2d, B, N, ctx_vec, stats=None):
        """x2d [B*N, C] fp16 residual stream; ctx_vec [B, C] = attn2's constant (single_token_context); stats = ops.RowStats of x2d's
        rows when its producer emitted them (SpatialTransformer's proj_in does), else they are computed here."""
        p = self.pk()
        a1 = self.attn1.pk()
        if self.linear_fp8:                      # BASELINE configs[4]: LayerNorm emits e4m3 + a scale per token; q|k, V^T and the GEGLU projection read it
            a = self.attn1.self_attention_f8(*ops.layernorm_f8(x2d, p.g1, p.b1, p.eps1), B, N)
            x1 = ops.gemm(a, a1.wo, a1.bo, rowvec=ctx_vec, group_rows=N, resid=x2d)
            return self.ff.run_f8(*ops.layernorm_f8(x1, p.g3, p.b3, p.eps3), resid=x1)
        if self._folded(p, N):                   # 5 launches: q|k|v^T, attention, to_out (+ row statistics), GEGLU, ff out
            a = self.attn1.self_attention_fused(x2d, stats if stats is not None else ops.row_stats(x2d), p, B, N)
            x1, st3 = ops.gemm(a, a1.wo, a1.bo, rowvec=ctx_vec, group_rows=N, resid=x2d, row_stats=True)   # attn1 + x, + attn2 constant
            h = ops.gemm(x1, p.wg, p.c2g, act=ops.ACT_GEGLU, ln=(st3, p.c1g, p.eps3))
            fp = self.ff.pk()
            return ops.gemm(h, fp.w2, fp.b2, resid=x1)
        a = self.attn1.self_attention(ops.layernorm(x2d, p.g1, p.b1, p.eps1), B, N)
        x1 = ops.gemm(a, a1.wo, a1.bo, rowvec=ctx_vec, group_rows=N, resid=x2d)          # attn1 + x, + attn2 constant
        return self.ff.run(ops.layernorm(x1, p.g3, p.b3, p.eps3), resid=x1)

    def run_paired(self, x2d, B, N, ctx_vec, stats=None):
        """Guidance pair with a SHARED input (x2d [B*N, C] serves both halves, ctx_vec [2B, C] differs): LayerNorm, q/k/v
        and the attention core do not depend on the context, so they run once at batch B; the two halves part where
        attn2's constant is added (the out-projection epilogue).  Returns [2B*N, C]."""
        p = self.pk()
        x1 = torch.empty((2 * B * N, x2d.shape[1]), dtype=torch.float16, device=x2d.device)
        if self._folded(p, N):
            a1 = self.attn1.pk()
            maxp = (x2d.shape[1] + 63) // 64
            st3 = ops.RowStats(torch.empty((maxp, 2 * B * N, 2), dtype=torch.float32, device=x2d.device), maxp, 2 * B * N)
            with ops.pinned_batch_scale(2):          # batch-B launches take the tile (= statistics partials) of the batch-2B layer: same bits
                a = self.attn1.self_attention_fused(x2d, stats if stats is not None else ops.row_stats(x2d), p, B, N)
                for half in (0, 1):
                    _, got = ops.gemm(a, a1.wo, a1.bo, rowvec=ctx_vec[half * B:(half + 1) * B], group_rows=N, resid=x2d, out=x1[half * B * N:(half + 1) * B * N],
                                      row_stats=ops.RowStats(st3.buf, maxp, st3.ld, half * B * N))
            h = ops.gemm(x1, p.wg, p.c2g, act=ops.ACT_GEGLU, ln=(ops.RowStats(st3.buf, got.parts, st3.ld), p.c1g, p.eps3))
            fp = self.ff.pk()
            return ops.gemm(h, fp.w2, fp.b2, resid=x1)
        with ops.pinned_batch_scale(2):              # batch-B launches take the split-K factor of the batch-2B layer: same bits
            if self.linear_fp8:                      # (fp8 GEMMs never split K: nothing to pin for them)
                a = self.attn1.self_attention_f8(*ops.layernorm_f8(x2d, p.g1, p.b1, p.eps1), B, N)
            else:
                a = self.attn1.self_attention(ops.layernorm(x2d, p.g1, p.b1, p.eps1), B, N)
            a1 = self.attn1.pk()
            for half in (0, 1):
                ops.gemm(a, a1.wo, a1.bo, rowvec=ctx_vec[half * B:(half + 1) * B], group_rows=N, resid=x2d, out=x1[half * B * N:(half + 1) * B * N])
        if self.linear_fp8:
            return self.ff.run_f8(*ops.layernorm_f8(x1, p.g3, p.b3, p.eps3), resid=x1)
        return self.ff.run(ops.layernorm(x1, p.g3, p.b3, p.eps3), resid=x1)

    def forward(self, x, context=None):
        x = _tokens(x)
        B, N, Cc = x.shape
        if context is None or context.shape[1] != 1:
            raise PbeError("BasicTransformerBlock: the HIP path expects a one-token context [B, 1, D]")
        return self.run(x.view(B * N, Cc), B, N, self.attn2.single_token_context(context)).view(B, N, Cc)


class SpatialTransformer(HipModule):
    """GroupNorm(eps 1e-6) -> 1x1 proj_in -> transformer block(s) -> 1x1 proj_out -> + input."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None):
        super().__init__()
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.proj_in = nn.Conv2d(in_channels, inner, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim) for _ in range(depth)])
        self.proj_out = zero_module(nn.Conv2d(inner, in_channels, kernel_size=1, stride=1, padding=0))

    def _pack(self):
        return SimpleNamespace(g=f32(self.norm.weight), b=f32(self.norm.bias), eps=self.norm.eps,
                               wi=ops.pack_linear(self.proj_in.weight), bi=f32(self.proj_in.bias),
                               wo=ops.pack_linear(self.proj_out.weight), bo=f32(self.proj_out.bias))

    def context_vectors(self, context):
        return [blk.attn2.single_token_context(context) for blk in self.transformer_blocks]

    def run(self, x, ctx_vecs):
        """x [B, H, W, C] fp16 NHWC -> same shape."""
        p = self.pk()
        B, H, W, Cc = x.shape
        N = H * W
        h, st = ops.gemm(ops.groupnorm(x, p.g, p.b, p.eps, False).view(B * N, Cc), p.wi, p.bi, row_stats=True)   # statistics for the first block's norm1
        for blk, cv in zip(self.transformer_blocks, ctx_vecs):
            h, st = blk.run(h, B, N, cv, stats=st), None
        return ops.gemm(h, p.wo, p.bo, resid=x.view(B * N, Cc)).view(B, H, W, Cc)

    def run_paired(self, x, ctx_vecs):
        """x [B, H, W, C] shared by the two halves of a guidance pair, ctx_vecs for 2B samples -> [2B, H, W, C]."""
        p = self.pk()
        B, H, W, Cc = x.shape
        N = H * W
        x2d = x.view(B * N, Cc)
        with ops.pinned_batch_scale(2):
            h, st = ops.gemm(ops.groupnorm(x, p.g, p.b, p.eps, False).view(B * N, Cc), p.wi, p.bi, row_stats=True)
        h = self.transformer_blocks[0].run_paired(h, B, N, ctx_vecs[0], stats=st)
        for blk, cv in zip(list(self.transformer_blocks)[1:], ctx_vecs[1:]):
            h = blk.run(h, 2 * B, N, cv)
        y = torch.empty((2 * B * N, Cc), dtype=torch.float16, device=x.device)
        with ops.pinned_batch_scale(2):
            for half in (0, 1):
                ops.gemm(h[half * B * N:(half + 1) * B * N], p.wo, p.bo, resid=x2d, out=y[half * B * N:(half + 1) * B * N])
        return y.view(2 * B, H, W, Cc)

    def forward(self, x, context=None):
        """Reference layout: x [B, C, H, W] -> [B, C, H, W]."""
        require_gpu(x, "SpatialTransformer")
        if context is None:
            raise PbeError("SpatialTransformer: context is required on the Paint-by-Example path")
        y = self.run(ops.nchw_to_nhwc(x.float()), self.context_vectors(context))
        return ops.nhwc_to_nchw(y).to(x.dtype)
