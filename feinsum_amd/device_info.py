"""
Peak flop-rate / bandwidth tables for the roofline model.

Same two tables as the reference's ``feinsum.data.device_info`` (reference:
``src/feinsum/data/device_info.py:5-26``), with the MI355X row the build adds.
MI355X: fp64 vector = matrix = 78.6 TFLOP/s (256 CUs x 128 flop/clk x 2.4 GHz),
fp32 157.3 TFLOP/s, HBM3E 8.0 TB/s (datasheet; ~6.3 TB/s is what a streaming
read achieves -- /opt/skills/guides/MI355X_MICROARCH.md, profiles/r01_microbench_gfx950.txt).
"""

from typing import Mapping

#: device name -> dtype name -> peak GFLOP/s
DEV_TO_PEAK_GFLOPS: Mapping[str, Mapping[str, float]] = {
    "NVIDIA TITAN V": {"float32": 12288, "float64": 6144},
    "NVIDIA GeForce GTX 1650": {"float32": 3916.0, "float64": 122.4},
    "NVIDIA H200 NVL": {"float32": 67000, "float64": 34000},
    "AMD Instinct MI355X": {"float32": 157300, "float64": 78600},
}

#: device name -> peak global-memory bandwidth in GB/s
DEV_TO_PEAK_BW: Mapping[str, float] = {
    "NVIDIA TITAN V": 652.8,
    "NVIDIA GeForce GTX 1650": 192.0,
    "NVIDIA H200 NVL": 4800,
    "AMD Instinct MI355X": 8000.0,
}

MI355X = "AMD Instinct MI355X"


def normalize_device_name(name: str) -> str:
    """Map the spellings a ROCm stack reports for the part to the table key."""
    low = (name or "").lower()
    if "mi355" in low or "gfx950" in low:
        return MI355X
    return name
