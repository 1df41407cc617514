"""Reading a Paint-by-Example / Stable-Diffusion Lightning checkpoint without executing anything from it.

The reference does ``torch.load(ckpt, map_location="cpu")["state_dict"]`` (scripts/inference.py:60-62,
ldm/models/diffusion/ddpm.py:245-260) — a full unpickle.  The published ``model.ckpt`` is a Lightning 1.4 file: next
to ``state_dict`` its pickle holds ``callbacks`` / ``hyper_parameters`` / ``optimizer_states`` objects whose classes
live in ``pytorch_lightning`` / ``omegaconf``.  ``torch.load(..., weights_only=True)`` refuses the WHOLE file on the
first such global, although only ``state_dict`` is wanted.

``read_state_dict`` therefore tries the weights-only loader first and, when it refuses the file, reads the zip
container itself with a pickle reader whose ``find_class`` resolves ONLY the tensor-rebuild helpers below and maps
every other global to an inert placeholder (calling, constructing or ``__setstate__``-ing it does nothing).  No code
named by the file is ever imported or run; tensors come from the archive's raw storage records.
"""
from __future__ import annotations

import collections
import io
import pickle
import zipfile
from typing import Any, Dict, Mapping

import torch

_STORAGE_DTYPES = {
    "FloatStorage": torch.float32, "HalfStorage": torch.float16, "BFloat16Storage": torch.bfloat16, "DoubleStorage": torch.float64,
    "LongStorage": torch.int64, "IntStorage": torch.int32, "ShortStorage": torch.int16, "CharStorage": torch.int8,
    "ByteStorage": torch.uint8, "BoolStorage": torch.bool,
}


class CheckpointError(RuntimeError):
    pass


class _Inert:
    """Stand-in for any global outside the allow-list: absorbs construction, calls and state."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __setstate__(self, state):
        pass

    def __setitem__(self, k, v):
        pass

    def append(self, v):
        pass

    def extend(self, v):
        pass

    def update(self, *a, **k):
        pass

    def add(self, v):
        pass


class _StorageRef:
    def __init__(self, dtype):
        self.dtype = dtype


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    if not isinstance(storage, torch.Tensor):
        raise CheckpointError("tensor record without a storage")
    return torch.as_strided(storage, tuple(size), tuple(stride), int(storage_offset))


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None, *rest):
    return data


_ALLOWED = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("torch._utils", "_rebuild_tensor_v2"): _rebuild_tensor_v2,
    ("torch._utils", "_rebuild_parameter"): _rebuild_parameter,
    ("torch._utils", "_rebuild_parameter_with_state"): _rebuild_parameter,
    ("torch", "Size"): tuple,
}


class _StateDictUnpickler(pickle.Unpickler):
    def __init__(self, file, zf: zipfile.ZipFile, prefix: str):
        super().__init__(file)
        self._zf, self._prefix, self._cache = zf, prefix, {}
        self.skipped = set()

    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return _ALLOWED[(module, name)]
        if module in ("torch", "torch.storage") and name in _STORAGE_DTYPES:
            return _StorageRef(_STORAGE_DTYPES[name])
        if module == "torch" and name in ("float32", "float16", "bfloat16", "float64", "int64", "int32", "int16", "int8", "uint8", "bool"):
            return getattr(torch, name)
        self.skipped.add(f"{module}.{name}")
        return _Inert

    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage"):
            raise CheckpointError(f"unsupported persistent id {pid!r}")
        _, stype, key, _location, numel = pid[:5]
        if not isinstance(stype, _StorageRef):
            raise CheckpointError("storage record with an unknown storage type")
        key = str(key)
        if key not in self._cache:
            raw = self._zf.read(f"{self._prefix}/data/{key}")
            need = int(numel) * torch.empty((), dtype=stype.dtype).element_size()
            if len(raw) < need:
                raise CheckpointError(f"storage {key}: {len(raw)} bytes on file, {need} expected")
            self._cache[key] = torch.frombuffer(bytearray(raw[:need]), dtype=stype.dtype) if need else torch.empty(0, dtype=stype.dtype)
        return self._cache[key]


def _read_restricted(path: str) -> Any:
    if not zipfile.is_zipfile(path):
        raise CheckpointError(f"{path}: not a zip-format torch checkpoint (torch < 1.6 legacy files are not supported); re-save it "
                              "with torch.save(torch.load(...)['state_dict'], ...) in the environment that wrote it")
    with zipfile.ZipFile(path) as zf:
        pkl = [n for n in zf.namelist() if n.endswith("/data.pkl")]
        if len(pkl) != 1:
            raise CheckpointError(f"{path}: expected exactly one data.pkl record, found {pkl}")
        prefix = pkl[0][:-len("/data.pkl")]
        up = _StateDictUnpickler(io.BytesIO(zf.read(pkl[0])), zf, prefix)
        obj = up.load()
    return obj, sorted(up.skipped)


def read_state_dict(path: str, verbose: bool = False) -> Dict[str, torch.Tensor]:
    """``torch.load(path)["state_dict"]`` of the reference (scripts/inference.py:60-62) with no code execution.
    Accepts a Lightning checkpoint (``{"state_dict": ...}``) or a bare state dict."""
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
        skipped = []
    except pickle.UnpicklingError as e:
        obj, skipped = _read_restricted(path)
        if verbose:
            print(f"[checkpoint] weights-only loader refused {path} ({str(e).splitlines()[0][:120]}); "
                  f"read with the restricted reader, {len(skipped)} foreign globals ignored")
    if isinstance(obj, Mapping) and "state_dict" in obj:
        obj = obj["state_dict"]
    if not isinstance(obj, Mapping):
        raise CheckpointError(f"{path}: no state_dict found (top-level object is {type(obj).__name__})")
    sd = {k: v for k, v in obj.items() if isinstance(k, str) and isinstance(v, torch.Tensor)}
    if not sd:
        raise CheckpointError(f"{path}: state_dict holds no tensors")
    return sd
