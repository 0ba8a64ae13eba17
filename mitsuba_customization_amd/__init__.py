"""mitsuba_customization_amd — MI355X (gfx950) implementation of the MERL / customized_measurement
BSDF eval()/sample()/pdf() hot path behind the Mitsuba plugin interfaces.

Layout:
  csrc/      HIP kernels + the C ABI (include/merl_hip.h)  -> lib/libmerl_hip.so
  adapters/  Mitsuba 0.6 / Mitsuba 3 plugin classes over the C ABI (C++)
  host.py    ctypes host over the C ABI (tests, bench, sharding plumbing)
  shard.py   index-tile sharding + result gather over torch.distributed (RCCL on the GPU box)
  synth.py   synthetic MERL-layout tables and .binary file helpers
"""
__all__ = ["host", "synth", "build"]
