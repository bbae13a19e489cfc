"""bramble_amd: MI355X-native genome -> transcriptome projection (the hot path of
zrudnick/bramble behind its own plain-struct API).  See DESIGN.md.

  lib     ctypes binding of libbramble_amd.so (C ABI: include/bramble_amd.h)
  batch   struct-of-arrays alignment batches
  device  torch plumbing for HBM-resident batches
  synth   seeded synthetic annotation / alignments
"""
from . import batch  # noqa: F401

__all__ = ["batch", "lib", "device", "synth"]
