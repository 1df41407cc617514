"""Build libpbe_hip.so (gfx950 only) in-tree with hipcc.  ``python -m pbe_amd.build [--force]``.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
``pbe_amd/libpbe_hip.so`` is git-ignored but travels with the tree to the GPU box.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libpbe_hip.so")
import importlib.util as _ilu

_spec = _ilu.spec_from_file_location("_pbe_lib_for_build", os.path.join(HERE, "lib.py"))      # lib.py alone: no torch import at module level
_libmod = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(_libmod)
SOURCES = list(_libmod.SOURCES)
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-ffp-contract=fast"]
# per-file extras: keep MFMA results in VGPRs where the VALU consumes them right away (attention softmax),
# saving ~100 v_accvgpr_read/write per K/V tile
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
         # SLP re-pairs the whole accumulator tile into (r0,r2)/(r1,r3) operands for v_pk_fma_f32 (no faster than v_fma_f32 on
         # gfx950, tools/ubench_valu.hip) above the epilogue: a second 128-160 VGPRs, i.e. scratch spills in the 256-row tiles.
         **{f: ["-fno-slp-vectorize"] for f in ("igemm.hip", "igemm_dense.hip", "igemm_conv.hip", "igemm_halo.hip", "igemm_f8.hip", "igemm_ex.hip", "igemm_ex_ln.hip",
                                                  "igemm_ex_st.hip", "igemm_ex_qkv.hip", "igemm_ex_all.hip", "igemm_astat.hip")}}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpbe_hip.so cannot be built")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, extra) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, "common.h"), os.path.join(CSRC, "igemm_kernel.h"), os.path.join(HERE, "..", "include", "pbe_hip.h"),
            os.path.abspath(__file__)]
    hash_flag = []
    if src == "runtime.hip":                     # the library's identity: rebuilt whenever ANY source changed
        hash_flag = [f'-DPBE_SRC_HASH="{_libmod.source_hash()}"']
        deps = deps + [os.path.join(HERE, rel) for rel in _libmod.HASHED]
    if _stale(obj, deps):
        cmd = [_hipcc(), *FLAGS, *EXTRA.get(src, []), *extra, *hash_flag, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build_diagnostic(defines, out_path: str) -> str:
    """A separate library with extra -D flags (e.g. PBE_STAMPS) for tools/: own object directory, never the shipped .so."""
    obj_dir = out_path + ".obj"
    os.makedirs(obj_dir, exist_ok=True)
    objs = []

    def one(src):
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc(), *FLAGS, *EXTRA.get(src, []), *[f"-D{d}" for d in defines], "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(one, SOURCES))
    r = subprocess.run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", out_path], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return out_path


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
        if os.path.exists(LIB):
            os.remove(LIB)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, list(extra)), SOURCES))
    if _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.2f} MB)")
    return LIB


if __name__ == "__main__":
    for a in sys.argv[1:]:
        if a.startswith("--diag="):                   # --diag=name:DEFINE[=v][,DEFINE...]  ->  tools/_dbg/libpbe_hip_<name>.so
            name, defs = a[len("--diag="):].split(":", 1)
            print(build_diagnostic(defs.split(","), os.path.join(HERE, "..", "tools", "_dbg", f"libpbe_hip_{name}.so")))
            sys.exit(0)
    if "--stamps" in sys.argv:
        print(build_diagnostic(["PBE_STAMPS"], os.path.join(HERE, "..", "tools", "_dbg", "libpbe_hip_stamps.so")))
    else:
        build(force="--force" in sys.argv, verbose=True)
