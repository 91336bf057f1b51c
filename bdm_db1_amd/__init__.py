"""bdm_db1_amd: MI355X-native (gfx950) core for the DB1 hot path.

Importing the package does not need a GPU; constructing a model or calling an op does, and raises
if libdb1_hip.so or a gfx950 device is missing (there is no CPU fallback).
"""
__all__ = ["TransformerXL", "initialize", "mpu", "GraphedMemoryStep", "GraphedRingStep", "RingMemory", "GraphedTrainStep"]


def __getattr__(name):
    if name == "TransformerXL":
        from .model import TransformerXL
        return TransformerXL
    if name == "initialize":
        from .engine import initialize
        return initialize
    if name == "GraphedMemoryStep":
        from .decode import GraphedMemoryStep
        return GraphedMemoryStep
    if name in ("GraphedRingStep", "RingMemory"):
        from . import decode
        return getattr(decode, name)
    if name == "GraphedTrainStep":
        from .graphed_train import GraphedTrainStep
        return GraphedTrainStep
    if name == "mpu":
        import importlib
        return importlib.import_module(".mpu", __name__)
    raise AttributeError(name)
