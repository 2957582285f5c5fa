"""MI355X-native TAO-Amodal evaluation hot path."""
__all__ = ["columns", "synth"]
