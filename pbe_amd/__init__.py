"""pbe_amd: MI355X (gfx950) native hot path for Paint-by-Example PLMS inference.

The package holds only what the hot path needs:

* ``csrc/``       hand-written HIP kernels and the C-ABI (``libpbe_hip.so``)
* ``lib``         ctypes loader for the C-ABI (fails loudly when the library is absent)
* ``ops``         thin tensor -> pointer wrappers around the C-ABI entry points
* ``spec``        the path's topology derived from ``configs/v1.yaml`` (names + shapes of
                  every weight tensor, in the reference's ``state_dict`` key layout)
* ``weights``     deterministic name-seeded weight synthesis and checkpoint key remap
* ``shard``       batch-axis sharding, RCCL weight broadcast and result gather

The reference-facing API (``ldm.*`` import paths) lives in the top-level ``ldm`` package.
"""

__version__ = "0.1.0"
