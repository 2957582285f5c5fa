"""Make the *reference* evaluator importable in the development container.

Only used by ``make_golden.py`` (fixture generation) and by the optional
``-m reference`` tests; nothing here runs on the GPU box, where
``/root/reference`` does not exist.  Nothing from the reference is copied into
the repository: the vendored pycocotools source is compiled *in a scratch
directory under /tmp* from where it lies in ``/root/reference``, and the
modules the reference imports but this image lacks (numba, cv2, detectron2)
are replaced by the minimal stand-ins below (shims
1-5).
"""
import os
import shutil
import subprocess
import sys

REF = "/root/reference"
PCOCO_SRC = os.path.join(
    REF, "visualization/tao/third_party/pysot/training_dataset/coco/"
    "pycocotools")

_SHIMS = {
    "numba/__init__.py": (
        "def jit(*a, **k):\n"
        "    if len(a) == 1 and callable(a[0]) and not k:\n"
        "        return a[0]\n"
        "    return lambda f: f\n"),
    "cv2/__init__.py": "",
    "detectron2/__init__.py": "",
    "detectron2/utils/__init__.py": "",
    "detectron2/utils/logger.py": (
        "from tabulate import tabulate\n"
        "def create_small_table(small_dict):\n"
        "    keys, values = tuple(zip(*small_dict.items()))\n"
        "    return tabulate([values], headers=keys, tablefmt='pipe',\n"
        "                    floatfmt='.3f', stralign='center',\n"
        "                    numalign='center')\n"),
    "detectron2/evaluation/__init__.py": (
        "def inference_on_dataset(*a, **k): raise NotImplementedError\n"
        "def print_csv_format(*a, **k): raise NotImplementedError\n"),
    "sitecustomize.py": (
        "import numpy as np\n"
        "if not hasattr(np, 'float'): np.float = float\n"),
}


def available():
    return os.path.isdir(os.path.join(REF, "tao_amodal"))


def setup(workdir="/tmp/tao_ref_env"):
    """Build/locate the scratch environment; return the sys.path entries."""
    if not available():
        raise RuntimeError("reference tree not present")
    shims = os.path.join(workdir, "shims")
    for rel, text in _SHIMS.items():
        p = os.path.join(shims, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(text)
    pc = os.path.join(workdir, "pcoco")
    built = [f for f in (os.listdir(os.path.join(pc, "pycocotools"))
                         if os.path.isdir(os.path.join(pc, "pycocotools"))
                         else []) if f.startswith("_mask") and f.endswith(".so")]
    if not built:
        shutil.rmtree(pc, ignore_errors=True)
        shutil.copytree(PCOCO_SRC, os.path.join(pc, "pycocotools"))
        subprocess.run("chmod -R u+w .", shell=True, check=True,
                       cwd=os.path.join(pc, "pycocotools"))
        subprocess.run([sys.executable, "setup.py", "build_ext", "--inplace"],
                       cwd=os.path.join(pc, "pycocotools"), check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [shims, pc, REF]


def import_reference(workdir="/tmp/tao_ref_env"):
    """Return (lvis_amodal module, tao_amodal module) of the reference."""
    paths = setup(workdir)
    for p in reversed(paths):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    if not hasattr(np, "float"):
        np.float = float
    import tao_amodal.evaluation.lvis_amodal as ref_lvis
    import tao_amodal.evaluation.tao_amodal as ref_tao
    return ref_lvis, ref_tao


def cli_env(workdir="/tmp/tao_ref_env"):
    paths = setup(workdir)
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join(paths[:2])
    return env
