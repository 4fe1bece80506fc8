"""FortranLibrary (MI355X build) -- Python entry point.

Mirrors the layout of the reference's ``FortranLibrary`` Python package
(/root/reference/FortranLibrary/__init__.py:1-4, basic.py:36: a ctypes shim over
``libFL.so``).  The reference package has no optimiser wrappers; this build adds
``FortranLibrary.NonlinearOptimization`` with the batched solvers.
"""
from .basic import FL, library_path  # noqa: F401
from .General import ShowTime, dScientificNotation  # noqa: F401
from . import NonlinearOptimization  # noqa: F401
