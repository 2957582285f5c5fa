"""MI355X-native TAO-Amodal evaluation hot path."""
import os as _os

# libgomp's idle threads SPIN by default.  The host side of this package runs
# several OpenMP teams side by side (the two native readers, the sorts of the
# table build) on boxes whose CPU quota is far below their core count: spinning
# teams starve each other -- one 3 M-key sort took 0.64 s instead of 0.016 s
# behind another team's region.  Read by libgomp when it is loaded, so this must
# run before the first library that brings it in (ours, or torch).
_os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

__all__ = ["columns", "synth"]
