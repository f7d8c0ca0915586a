"""Drop-in for the reference package utils/pytorch_structural_losses (same module names)."""
