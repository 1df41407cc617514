"""Import the CPU oracle (oracle/pbe_oracle.py) for tests — tests are the only place allowed to."""
import importlib.util
import os
import sys

_p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "pbe_oracle.py")
_spec = importlib.util.spec_from_file_location("pbe_oracle", _p)
O = importlib.util.module_from_spec(_spec)
sys.modules["pbe_oracle"] = O
_spec.loader.exec_module(O)
