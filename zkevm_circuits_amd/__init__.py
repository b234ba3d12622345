"""Importable alias for the ``zkevm-circuits_amd/`` source tree (a dash is not a legal module
name).  Everything lives in ``zkevm-circuits_amd/``; this package only extends its search path."""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "zkevm-circuits_amd"))

from .binding import *  # noqa: F401,F403,E402
from . import binding  # noqa: E402
