"""Load libFL.so (reference: FortranLibrary/basic.py:36 ``FL = CDLL('libFL.so')``).

The library is looked up next to this package first (in-tree build under
fortran-library_amd/lib), then by bare name on LD_LIBRARY_PATH like the reference
does.  There is no CPU fallback: if the HIP library is missing, importing fails.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path():
    if os.environ.get("FL_LIBRARY"):  # explicit override (tuning builds)
        return os.environ["FL_LIBRARY"]
    cand = os.path.join(os.path.dirname(_HERE), "lib", "libFL.so")
    return cand if os.path.exists(cand) else "libFL.so"


try:
    FL = ctypes.CDLL(library_path())
except OSError as exc:  # fail loudly: the product path is the HIP library, nothing else
    raise ImportError(
        "libFL.so (HIP build for gfx950) not found: build it with `make -C fortran-library_amd` "
        "or put it on LD_LIBRARY_PATH") from exc
