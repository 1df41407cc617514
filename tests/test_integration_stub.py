"""The ctypes stub INTEGRATION.md section 2 shows a reference maintainer is executed as written: extracted from the markdown,
bound to the in-tree libpbe_hip.so.  CPU: it loads, passes its own ABI / sizeof assertions and its struct equals pbe_amd.lib's.
GPU: its linear_f16 replaces torch.nn.functional.linear (attention.py:210-213) within fp16 tolerance."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", md, flags=re.S) if "class GemmDesc" in b]
    assert len(blocks) == 1, "INTEGRATION.md must hold exactly one ctypes stub"
    from pbe_amd import lib as L
    L.load()                                          # refuses a stale binary; the stub then binds the same file
    os.environ["PBE_LIB_PATH"] = L.LIB_PATH
    ns = {}
    try:
        exec(compile(blocks[0], "INTEGRATION.md:stub", "exec"), ns)   # noqa: S102  (our own documentation)
    finally:
        os.environ.pop("PBE_LIB_PATH", None)
    return ns, L


def test_documented_stub_matches_the_abi():
    ns, L = _stub_namespace()
    G = ns["GemmDesc"]
    assert C.sizeof(G) == C.sizeof(L.GemmDesc) == L.load().pbe_sizeof_gemm_desc()
    assert [(n, C.sizeof(t)) for n, t in G._fields_] == [(n, C.sizeof(t)) for n, t in L.GemmDesc._fields_]
    assert [getattr(G, n).offset for n, _ in G._fields_] == [getattr(L.GemmDesc, n).offset for n, _ in L.GemmDesc._fields_]
    assert L.load().pbe_abi_version() == L.ABI_VERSION == 8
    assert C.sizeof(L.Conv3x3Desc) == L.load().pbe_sizeof_conv3x3_desc() and C.sizeof(L.AttnDesc) == L.load().pbe_sizeof_attn_desc()


@pytest.mark.gpu
def test_documented_stub_runs_linear(dev):
    ns, _ = _stub_namespace()
    g = torch.Generator().manual_seed(5)
    for (M, N, K) in ((256, 320, 320), (77, 1280, 640), (1028, 1024, 4096)):
        x = (torch.randn(M, K, generator=g) * 0.5).to(dev).half()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).half()
        b = torch.randn(N, generator=g).to(dev)
        y = ns["linear_f16"](x, w, b)
        ref = torch.nn.functional.linear(x.float(), w.float(), b)
        rel = ((y.float() - ref).norm() / ref.norm()).item()
        assert rel < 1e-3, (M, N, K, rel)             # fp16 output rounding of an fp32-accumulated product: ~3e-4
