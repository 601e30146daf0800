"""movedepth_amd: MI355X-native hot path of MOVEDepth (cost volume + photometric loss training step).

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed) over a C-ABI HIP library
(libmovedepth_hip.so, include/movedepth_hip.h).  The modules mirror the reference's interface for the path:
`layers` (operator signatures of movedepth/layers.py), `trainer.Trainer` (process_batch / compute_losses).
"""
__version__ = "0.1.0"
