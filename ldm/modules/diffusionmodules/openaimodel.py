"""The 9-channel-input SD-v1 U-Net of Paint-by-Example on MI355X.

Same class names, constructor arguments and ``state_dict`` keys as
ldm/modules/diffusionmodules/openaimodel.py in zhanwenchen/pbe (UNetModel :528-889, ResBlock
:163-275, Upsample :91-119, Downsample :134-160, TimestepEmbedSequential :74-88) for the branch
``configs/v1.yaml`` selects (use_spatial_transformer=True, legacy=False, no scale-shift norm, no
resblock up/down); the arithmetic is HIP kernels over NHWC fp16 activations:

  * every 3x3 conv is an implicit GEMM on the matrix cores with bias, the timestep-embedding add
    (openaimodel.py:273) and the residual add (:275) fused in its epilogue; nearest-2x upsample
    (:116) and the skip concat (:883) are folded into the conv's gather — no upsampled or
    concatenated tensor is ever written,
  * GroupNorm32+SiLU is one fused two-pass kernel (fp32 statistics) that also reads the concat
    pair in place,
  * the 22 ``emb_layers`` Linear(SiLU(emb)) are ONE GEMM per forward,
  * the 16 cross-attention layers collapse to 16 per-sample bias vectors computed once per
    context and cached across the 51 PLMS calls.
"""
from types import SimpleNamespace

import torch
from torch import nn

from pbe_amd import ops
from pbe_amd.hipmodule import HipModule, f32, require_gpu
from pbe_amd.lib import PbeError
from ldm.modules.attention import SpatialTransformer
from ldm.modules.diffusionmodules.util import conv_nd, linear, normalization, timestep_embedding, zero_module


class TimestepBlock(nn.Module):
    """Marker base: blocks whose run() takes the timestep-embedding slice."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Container with the reference's name so state_dict keys line up (input_blocks.N.M.*)."""


def _small_cin_pad(c):
    return 8 if c <= 8 else 16


class Upsample(HipModule):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if not use_conv or dims != 2 or padding != 1:
            raise PbeError("Upsample: only the conv_resample 2-D form is on the Paint-by-Example path")
        self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=padding)

    def _pack(self):
        # F.interpolate(nearest, 2x) + conv3x3 (openaimodel.py:109-119) as four 2x2 convs on the source grid: 4 / 9 of the MACs
        return SimpleNamespace(w=ops.pack_conv3x3_up_phases(self.conv.weight), b=f32(self.conv.bias))

    def run(self, x):
        p = self.pk()
        return ops.conv3x3(x, p.w, p.b, upsample=True)

    def forward(self, x):
        require_gpu(x, "Upsample")
        return ops.nhwc_to_nchw(self.run(ops.nchw_to_nhwc(x.float()))).to(x.dtype)


class Downsample(HipModule):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, out_channels or channels, use_conv
        if not use_conv or dims != 2 or padding != 1:
            raise PbeError("Downsample: only the conv_resample 2-D form is on the Paint-by-Example path")
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)

    def _pack(self):
        return SimpleNamespace(w=ops.pack_conv3x3(self.op.weight), b=f32(self.op.bias))

    def run(self, x):
        p = self.pk()
        return ops.conv3x3(x, p.w, p.b, stride=2, pad=1)

    def forward(self, x):
        require_gpu(x, "Downsample")
        return ops.nhwc_to_nchw(self.run(ops.nchw_to_nhwc(x.float()))).to(x.dtype)


class ResBlock(HipModule, TimestepBlock):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, up=False, down=False):
        super().__init__()
        if use_scale_shift_norm or up or down or use_conv or dims != 2:
            raise PbeError("ResBlock: scale-shift / resblock_updown / conv-skip variants are not on the Paint-by-Example path")
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_checkpoint = use_checkpoint
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        self.skip_connection = nn.Identity() if self.out_channels == channels else conv_nd(dims, channels, self.out_channels, 1)

    def _pack(self):
        n1, c1, n2, c2 = self.in_layers[0], self.in_layers[2], self.out_layers[0], self.out_layers[3]
        ns = SimpleNamespace(g1=f32(n1.weight), b1=f32(n1.bias), eps1=n1.eps, w1=ops.pack_conv3x3(c1.weight), cb1=f32(c1.bias),
                             g2=f32(n2.weight), b2=f32(n2.bias), eps2=n2.eps, w2=ops.pack_conv3x3(c2.weight), cb2=f32(c2.bias), ws=None)
        if not isinstance(self.skip_connection, nn.Identity):
            ns.ws, ns.bs = ops.pack_linear(self.skip_connection.weight), f32(self.skip_connection.bias)
        return ns

    def run(self, x, emb_out, skip=None):
        """x [B,H,W,C1] (+ skip [B,H,W,C2] = the concat partner) ; emb_out [B, Cout] = Linear(SiLU(emb))."""
        p = self.pk()
        B, H, W, C1 = x.shape
        # GroupNorm reads the concat pair in place and writes the normalised concat as ONE tensor: the conv has a single source
        # (group_stats: the conv's copy-out leaves the statistics of the GroupNorm32 that reads its output - out_layers[0] here, the next
        #  block's or the transformer's norm for the block output - so that norm runs its normalisation pass only; openaimodel.py:213-227)
        h = ops.conv3x3(ops.groupnorm(x, p.g1, p.b1, p.eps1, True, x2=skip), p.w1, p.cb1, rowvec=emb_out, group_stats=32)
        h = ops.groupnorm(h, p.g2, p.b2, p.eps2, True)
        if p.ws is not None:
            a2 = None if skip is None else skip.view(B * H * W, -1)
            xs = ops.gemm(x.view(B * H * W, C1), p.ws, p.bs, a2=a2).view(B, H, W, -1)
        elif skip is not None:
            raise PbeError("ResBlock: identity skip with a concat input is impossible (channel mismatch)")
        else:
            xs = x
        return ops.conv3x3(h, p.w2, p.cb2, resid=xs, group_stats=32)

    def emb_out(self, emb):
        """emb [B, emb_channels] (pre-SiLU, as the reference passes it) -> [B, Cout] fp16."""
        l = self.emb_layers[1]
        e = torch.nn.functional.silu(emb.float()).to(torch.float16)     # boundary-only path; UNetModel batches this on the GPU
        return ops.gemm(e.contiguous(), ops.pack_linear(l.weight), f32(l.bias))

    def forward(self, x, emb):
        require_gpu(x, "ResBlock")
        y = self.run(ops.nchw_to_nhwc(x.float()), self.emb_out(emb))
        return ops.nhwc_to_nchw(y).to(x.dtype)


class UNetModel(HipModule):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, use_fp16=False,
                 num_heads=-1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None,
                 legacy=True, add_conv_in_front_of_unet=False):
        super().__init__()
        if not use_spatial_transformer or context_dim is None:
            raise PbeError("UNetModel: the HIP path implements the use_spatial_transformer=True branch (configs/v1.yaml:41-43)")
        if num_classes is not None or n_embed is not None or resblock_updown or use_scale_shift_norm or add_conv_in_front_of_unet or dims != 2:
            raise PbeError("UNetModel: class-conditional / codebook / resblock_updown / scale-shift / front-conv variants are out of scope")
        if not conv_resample:
            raise PbeError("UNetModel: conv_resample=False is not on the Paint-by-Example path")
        if num_heads == -1 and num_head_channels == -1:
            raise PbeError("Either num_heads or num_head_channels has to be set")
        if isinstance(context_dim, (list, tuple)):
            context_dim = list(context_dim)[0]
        channel_mult = tuple(channel_mult)
        attention_resolutions = tuple(attention_resolutions)
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.dropout, self.channel_mult = num_res_blocks, attention_resolutions, dropout, channel_mult
        self.conv_resample, self.num_classes, self.use_checkpoint = conv_resample, num_classes, use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.num_heads, self.num_head_channels = num_heads, num_head_channels
        self.num_heads_upsample = num_heads if num_heads_upsample == -1 else num_heads_upsample
        self.predict_codebook_ids = False
        self.context_dim = context_dim

        ted = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, ted), nn.SiLU(), linear(ted, ted))

        def res(cin, cout):
            return ResBlock(cin, ted, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint)

        def attn(ch):
            heads, dh = (num_heads, ch // num_heads) if num_head_channels == -1 else (ch // num_head_channels, num_head_channels)
            if legacy:
                dh = ch // heads
            return SpatialTransformer(ch, heads, dh, depth=transformer_depth, context_dim=context_dim)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        skip_chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(attn(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                skip_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                skip_chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res(ch, ch), attn(ch), res(ch, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [res(ch + skip_chans.pop(), model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(attn(ch))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))
        self.__dict__["_ctx_cache"] = None

    # ---- packs -------------------------------------------------------------------------------------
    def _resblocks(self):
        return [m for m in self.modules() if isinstance(m, ResBlock)]

    def _transformers(self):
        return [m for m in self.modules() if isinstance(m, SpatialTransformer)]

    def _pack(self):
        cin = self.in_channels
        c_in = self.input_blocks[0][0]
        rbs = self._resblocks()
        offs, o = [], 0
        for rb in rbs:
            offs.append(o)
            o += rb.out_channels
        ns = SimpleNamespace(
            cin_pad=_small_cin_pad(cin), w_in=ops.pack_conv3x3(c_in.weight, _small_cin_pad(cin)), b_in=f32(c_in.bias),
            te0_w=ops.pack_linear(self.time_embed[0].weight), te0_b=f32(self.time_embed[0].bias),
            te2_w=ops.pack_linear(self.time_embed[2].weight), te2_b=f32(self.time_embed[2].bias),
            emb_w=ops.pack_linear(torch.cat([rb.emb_layers[1].weight for rb in rbs], 0)),
            emb_b=f32(torch.cat([rb.emb_layers[1].bias for rb in rbs], 0)),
            emb_off={id(rb): (off, rb.out_channels) for rb, off in zip(rbs, offs)},
            go=f32(self.out[0].weight), bo=f32(self.out[0].bias), eps_o=self.out[0].eps,
            w_out=ops.pack_conv3x3(self.out[2].weight), b_out=f32(self.out[2].bias))
        self.__dict__["_ctx_cache"] = None
        self.__dict__["_emb_cache"] = {}
        return ns

    def invalidate_packs(self):
        super().invalidate_packs()
        self.__dict__["_ctx_cache"] = None
        self.__dict__["_emb_cache"] = {}

    def context_vectors(self, context):
        """The 16 per-sample cross-attention constants for this context (cached while the same
        context tensor is presented, i.e. across the 51 calls of one PLMS run)."""
        key = (context.data_ptr(), context._version, tuple(context.shape), context.dtype)
        c = self.__dict__.get("_ctx_cache")
        if c is None or c[0] != key:
            vecs = {id(st): st.context_vectors(context) for st in self._transformers()}
            c = (key, vecs, context)          # keep `context` alive so its data_ptr cannot be recycled
            self.__dict__["_ctx_cache"] = c
        return c[1]

    # ---- forward ---------------------------------------------------------------------------------
    def _run_block(self, block, h, emb_all, p, ctx, skip=None):
        for layer in block:
            if isinstance(layer, ResBlock):
                off, n = p.emb_off[id(layer)]
                h = layer.run(h, emb_all[:, off:off + n], skip)
                skip = None
            elif isinstance(layer, SpatialTransformer):
                h = layer.run(h, ctx[id(layer)])
            else:
                h = layer.run(h)
        return h

    def embedding_rows(self, step: int, device):
        """Linear(SiLU(time_embed(t))) of all 22 emb_layers for ONE timestep value, [1, sum Cout] fp16, cached per value: inside a sampler
        run every sample of every call shares the step's timestep (plms.py:144 torch_full), and the 50 schedule values recur for every batch,
        so the three GEMMs + the sinusoid kernel (~55 us of launches per U-Net call, M = 2B rows of pure launch latency) run once per value
        and weight set instead of 51 times per batch.  Same kernels, same bits as the per-call evaluation (a row of a GEMM does not depend on
        the other rows).  Dropped with the packs."""
        p = self.pk()
        cache = self.__dict__.setdefault("_emb_cache", {})
        row = cache.get((int(step), device))
        if row is None:
            t = torch.full((1,), int(step), device=device, dtype=torch.int64)
            e = ops.gemm(timestep_embedding(t, self.model_channels), p.te0_w, p.te0_b, act=ops.ACT_SILU)
            e = ops.gemm(e, p.te2_w, p.te2_b, act=ops.ACT_SILU)
            row = cache[(int(step), device)] = ops.gemm(e, p.emb_w, p.emb_b)
        return row

    def forward_nhwc(self, x16, timesteps, context, paired=False, step=None):
        """x16 [B,H,W,cin_pad] fp16 (channels >= in_channels zero) -> eps [B,H,W,out_channels] fp16.

        paired=True is the classifier-free-guidance call of the samplers (plms.py:182-189): the reference feeds
        cat([x]*2), cat([t]*2), cat([uc, c]) - both halves share x and t and differ ONLY in the context.  Then x16 holds the
        B shared inputs while timesteps / context hold 2B rows, and everything upstream of the first context-dependent
        operation (conv_in, the first ResBlock, the first transformer's GroupNorm / proj_in / LayerNorm / q,k,v /
        attention core) is computed once and serves both halves.  Common-subexpression elimination: every kernel is
        deterministic and per-sample, and the only batch-dependent choice that changes bits - the split-K factor, i.e. the
        fp32 summation order - is pinned to the batch-2B choice, so the duplicated evaluation gives the same bits
        (tools/layer_diff.py shows where un-pinned batch sizes part ways)."""
        p = self.pk()
        ctx = self.context_vectors(context)
        if step is not None:                                           # the caller vouches that every entry of `timesteps` equals `step` (the samplers do)
            emb_all = self.embedding_rows(step, x16.device).expand(timesteps.shape[0], -1)      # one cached row, stride-0 broadcast over the samples
        else:
            t_emb = timestep_embedding(timesteps, self.model_channels)
            e = ops.gemm(t_emb, p.te0_w, p.te0_b, act=ops.ACT_SILU)
            e = ops.gemm(e, p.te2_w, p.te2_b, act=ops.ACT_SILU)          # SiLU(emb): emb is only ever consumed through SiLU
            emb_all = ops.gemm(e, p.emb_w, p.emb_b)                        # all 22 emb_layers at once
        blocks = list(self.input_blocks)[1:]
        if paired:
            B = x16.shape[0]
            first = list(blocks[0])
            if timesteps.shape[0] != 2 * B or len(first) != 2 or not isinstance(first[0], ResBlock) or not isinstance(first[1], SpatialTransformer):
                raise PbeError("UNetModel.forward_nhwc(paired=True): needs 2B timesteps and a [ResBlock, SpatialTransformer] first block")
            # the prefix runs at batch B with the tile config / split-K factor of batch 2B (ops.pinned_batch_scale): it
            # reproduces the duplicated evaluation bit for bit (tests/test_model_gpu.py::test_full_unet_shared_guidance_prefix)
            with ops.pinned_batch_scale(2):
                h0 = ops.conv3x3_small(x16, p.w_in, p.b_in)
                off, n = p.emb_off[id(first[0])]
                r = first[0].run(h0, emb_all[:B, off:off + n])
            h = first[1].run_paired(r, ctx[id(first[1])])          # pins its own batch-B launches
            hs = [torch.cat([h0, h0], 0), h]
            blocks = blocks[1:]
        else:
            h = ops.conv3x3_small(x16, p.w_in, p.b_in)
            hs = [h]
        for block in blocks:
            h = self._run_block(block, h, emb_all, p, ctx)
            hs.append(h)
        h = self._run_block(self.middle_block, h, emb_all, p, ctx)
        for block in self.output_blocks:
            h = self._run_block(block, h, emb_all, p, ctx, skip=hs.pop())
        h = ops.groupnorm(h, p.go, p.bo, p.eps_o, True)
        return ops.conv3x3(h, p.w_out, p.b_out)

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """x [N, in_channels, H, W] (fp32 or fp16), timesteps [N] int, context [N, 1, context_dim] ->
        eps [N, out_channels, H, W] in fp16 (what the reference returns under torch.autocast)."""
        if y is not None:
            raise PbeError("UNetModel: class labels are not supported (num_classes is None)")
        require_gpu(x, "UNetModel")
        if context is None or timesteps is None:
            raise PbeError("UNetModel.forward needs timesteps and context")
        p = self.pk()
        x16 = ops.nchw_to_nhwc(x.float(), p.cin_pad)
        out = self.forward_nhwc(x16, timesteps, context)
        return ops.nhwc_to_nchw(out).to(torch.float16)
