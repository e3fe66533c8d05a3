"""MI355X-native SGEMM backend behind the reference's MY_MMult entry point.

Layout (only what the hot path needs):
  csrc/      hand-written gfx950 HIP kernels + the C-ABI shim (libmmult_hip.so)
  harness/   C++ host side: MY_MMult forwarder and the test_MMult sweep driver
  api.py     ctypes mirror of include/mmult_hip.h (+ torch device-pointer glue)
  shard.py   one-process-per-GPU row-panel shard over torch.distributed (RCCL)
  build.py   hipcc recipe
"""
from .api import *  # noqa: F401,F403
