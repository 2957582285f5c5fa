"""Track-level (TAO-Amodal) evaluation -- drop-in names for
``tao_amodal.evaluation.tao_amodal`` of the reference."""
import logging

from .tao import Tao
from .results import TaoResults
from .eval import TaoEval, Params

logging.basicConfig(
    format="[%(asctime)s] %(name)s %(levelname)s: %(message)s",
    datefmt="%m/%d %H:%M:%S", level=logging.WARN)

__all__ = ["Tao", "TaoResults", "TaoEval", "Params"]
