"""MI355X-native TAO-Amodal evaluation hot path.

(libgomp's idle threads SPIN by default, and the host side of this package
runs several OpenMP teams side by side on boxes whose CPU quota is far below
their core count: the ENTRY POINTS -- tools/eval_on_tao_amodal.py, bench.py --
set OMP_WAIT_POLICY=PASSIVE before the first library that loads libgomp;
importing the package does not touch the environment.)"""

__all__ = ["columns", "synth"]
