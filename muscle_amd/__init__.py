"""muscle_amd — MI355X-native MPCFlat all-pairs posterior stage (MUSCLE5 hot path).

The product is the C-ABI library `muscle_amd/csrc/libmpcgpu.so` (hand-written HIP for gfx950,
declared in include/mpcgpu.h). This package is only the thin Python plumbing used by tests and
bench.py (ctypes binding + torch.distributed sharding helpers). There is no CPU fallback here.
"""
