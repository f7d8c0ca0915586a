"""hyperpocket_amd — MI355X-native HyperPocket training-step hot path.

Python host side of the drop-in for gmum/3d-point-clouds-autocomplete: the same module layout
and call signatures as the reference (``model.full_model.FullModel``, ``losses.champfer_loss.
ChamferLoss``, ``utils.pytorch_structural_losses.{nn_distance,match_cost,StructuralLossesBackend}``)
over the C ABI of ``libhyperpocket_hip.so`` (include/hyperpocket_hip.h).  PyTorch is used for
device memory, streams, autograd bookkeeping and torch.distributed only.

There is NO CPU fallback: every op raises ``HipExtensionError`` when the HIP library or a GPU is
missing.
"""
from ._lib import HipExtensionError, library_path, load_library  # noqa: F401

__all__ = ["HipExtensionError", "library_path", "load_library"]
