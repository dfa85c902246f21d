"""Import-name alias: ``pydream`` is ``pydream_amd``.

A script written for PyDREAM (``from pydream.core import run_dream``, ``from pydream.parameters import SampledParam``,
``from pydream.convergence import Gelman_Rubin`` -- pydream/core.py:11, the shipped examples' imports) runs unchanged
on the MI355X engine when this repository precedes the reference on ``sys.path``.  Nothing is implemented here: every
submodule name is bound to the corresponding ``pydream_amd`` module object.
"""
import importlib
import sys

import pydream_amd

__version__ = pydream_amd.__version__

for _name in ("core", "Dream", "model", "parameters", "convergence", "Dream_shared_vars", "likelihoods"):
    _mod = importlib.import_module("pydream_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
del _name, _mod
