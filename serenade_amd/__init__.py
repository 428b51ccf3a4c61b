"""serenade_amd -- MI355X (gfx950) implementation of Serenade's VMIS-kNN `predict_next` hot path.

The package holds only what that path needs: the HIP kernels + C ABI (csrc/, built into
libserenade_hip.so), a ctypes binding (capi), the host-side mirror of the reference interface
(vmisknn) and the synthetic workload generator used by bench.py (synth)."""
from .vmisknn import CSR, ItemScore, VMISIndex, SerenadeError, predict, predict_batch, predict_batch_debug, predict_batch_device, reserve  # noqa: F401

__all__ = ["CSR", "ItemScore", "VMISIndex", "SerenadeError", "predict", "predict_batch", "predict_batch_debug", "predict_batch_device", "reserve"]
