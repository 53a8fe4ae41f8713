"""vtc_amd -- MI355X-native (gfx950) implementation of the VTC retrieval forward/eval hot path.

Layout:
  csrc/    hand-written HIP kernels + the C ABI (include/vtc_hip.h) -> lib/libvtc_hip.so
  _lib.py  ctypes binding (fails loudly when the library is missing; no fallback)
  ops.py   torch-tensor wrappers over the C ABI
  towers.py weight packing + tower launches
  host/    host-side mirror of the reference's interface for this path
           (model.model, model.loss, model.metric, evaluation.eval, utils.parse_config)
  dist.py  one-process-per-GPU sharded gallery encode + sweep (RCCL all-gather / all-reduce)
"""
__version__ = "0.1.0"
