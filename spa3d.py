"""`import spa3d` -> the package in ./3dspa_code_amd (whose directory name is not an identifier)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
  sys.path.insert(0, _here)
_pkg = importlib.import_module('3dspa_code_amd')
sys.modules[__name__] = _pkg
