"""MI355X-native forget/retain training step for LLaVA-1.5 (drop-in for mm/trainer.py compute_loss)."""
__version__ = "0.1.0"
