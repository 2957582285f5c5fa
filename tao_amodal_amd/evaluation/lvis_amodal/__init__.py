"""Image-level (LVIS-style, amodal) evaluation -- drop-in names for
``tao_amodal.evaluation.lvis_amodal`` of the reference."""
import logging

from .lvis import LVIS
from .results import LVISResults
from .eval import LVISEval, Params

# same root-logger setup as the reference package (lvis_amodal/__init__.py:7-10)
logging.basicConfig(
    format="[%(asctime)s] %(name)s %(levelname)s: %(message)s",
    datefmt="%m/%d %H:%M:%S", level=logging.WARN)

__all__ = ["LVIS", "LVISResults", "LVISEval", "Params"]
