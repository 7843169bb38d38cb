"""ctypes binding of the CPU oracle built for BLS12-377 (oracle/_build/libripp_oracle_377.so, -DORC_BLS12_377) -- TEST INFRASTRUCTURE ONLY.
The same wrapper code as orclib.py, re-executed as its own module with the 377 library path and moduli."""
import importlib.util
import os
import sys

_spec = importlib.util.spec_from_file_location("orclib377", os.path.join(os.path.dirname(os.path.abspath(__file__)), "orclib.py"))
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
_m.LIB_PATH = os.path.join(_m.ORACLE_DIR, "_build", "libripp_oracle_377.so")
_x = 0x8508C00000000001
_m.R = _x**4 - _x**2 + 1
_m.P = (_x - 1) ** 2 * _m.R // 3 + _x
sys.modules[__name__] = _m
