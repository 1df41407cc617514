"""``ldm.data.test_bench_dataset.COCOImageDataset`` of zhanwenchen/pbe (test_bench_dataset.py:61-105):
same constructor argument, item layout and id zero-padding; PIL + numpy only (the reference imports
torchvision / albumentations / clip / bezier at module level without using them on this path)."""
from pbe_amd.testbench import COCOImageDataset, id_stem  # noqa: F401
