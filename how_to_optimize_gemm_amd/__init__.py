"""Importable alias for the package directory `how-to-optimize-gemm_amd/`
(a hyphen cannot appear in a Python module name).  All code lives there."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..",
                                 "how-to-optimize-gemm_amd"))
from .api import *  # noqa: F401,F403,E402
from . import api, build  # noqa: F401,E402
