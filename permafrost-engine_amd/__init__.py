"""permafrost-engine navigation hot path on MI355X (gfx950).

Holds only what the path needs: `csrc/` (hand-written HIP kernels + the C-ABI
library libnavhip.so declared in include/navhip.h), `navhip.py` (host-side
mirror of the reference's N_* interface over that C ABI), `synth.py`
(deterministic synthetic maps / requests / agents for BASELINE.json's configs)
and `dist.py` (one-process-per-GPU sharding over RCCL).
"""
__all__ = ["navhip", "synth", "dist", "build"]
