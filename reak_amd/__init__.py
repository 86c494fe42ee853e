"""reak_amd -- MI355X-native sampling-based-planning hot path for ReaK (HIP kernels behind a C-ABI)."""
__version__ = "0.1.0"
