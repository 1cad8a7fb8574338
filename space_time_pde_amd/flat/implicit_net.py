"""Flat-name alias of space_time_pde_amd.implicit_net (the reference imports its modules by bare name after
``sys.path.append("../../src")``, experiments/rb2d/train.py:19-27)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("space_time_pde_amd.implicit_net")
