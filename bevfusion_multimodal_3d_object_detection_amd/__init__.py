"""MI355X-native BEV-fusion detector hot path (gfx950 HIP kernels behind the reference's module API)."""
__version__ = "0.1.0"
