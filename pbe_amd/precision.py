"""Operand precision of the U-Net's LayerNorm-fed linear projections (BASELINE configs[4]: "fp8 MFMA attention/linear path").

"fp16" (default): the reference's autocast precision - fp16 operands, fp32 accumulate.
"fp8":  in every BasicTransformerBlock the two LayerNorms emit OCP e4m3 with one scale per token (pbe_layernorm_f8) and the
        projections that read them - q|k, V^T and the GEGLU projection, 42 % of the linear FLOPs of a U-Net forward - run
        v_mfma_f32_16x16x32_fp8_fp8 on e4m3 weights with one scale per output channel (fp32 accumulate, scales applied in the
        epilogue).  The attention core, the output projections, the convolutions, the VAE and CLIP stay fp16.
        Not the reference's precision: its results are held to a separate, re-validated tolerance (tests/test_model_gpu.py).
"""
from __future__ import annotations

import torch


def set_linear_precision(model: torch.nn.Module, precision: str = "fp16") -> int:
    """Switch every BasicTransformerBlock under `model`; returns how many were switched."""
    from ldm.modules.attention import BasicTransformerBlock
    if precision not in ("fp16", "fp8"):
        raise ValueError(f"precision must be 'fp16' or 'fp8', got {precision!r}")
    n = 0
    for m in model.modules():
        if isinstance(m, BasicTransformerBlock):
            m.linear_fp8 = precision == "fp8"
            n += 1
    return n
