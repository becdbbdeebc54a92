"""pantas_amd -- MI355X-native augment+call engine behind the `pantas augment` / `pantas call`
command lines of AlgoLab/pantas.  The compute lives in libpantas_gpu.so (HIP, gfx950)."""
__all__ = ["engine", "augment", "synth"]
