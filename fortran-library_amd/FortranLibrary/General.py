"""General utilities bound by the reference's Python package (/root/reference/FortranLibrary/General.py:1-16):
``ShowTime`` and ``dScientificNotation``.  Same names and call shapes; both are plain host functions of libFL.so."""
from ctypes import byref, c_double, c_int

from .basic import FL

ShowTime = FL.__general_MOD_showtime
ShowTime.restype = None


def dScientificNotation(x: float):
    """x = mantissa * 10**exponent with 1 <= mantissa < 10 -> (mantissa, exponent)"""
    xc, i = c_double(x), c_int(0)
    FL.__general_MOD_dscientificnotation(byref(xc), byref(i))
    return xc.value, i.value
