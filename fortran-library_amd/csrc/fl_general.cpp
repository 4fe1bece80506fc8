// fl_general.cpp -- the two host utilities of the reference's General module that its Python package binds
// (FortranLibrary/General.py:4-16; the ShowTime probe runs at import time) and its C++ header declares
// (cpp/FortranLibrary.hpp:48).  Host code by nature: no arithmetic of the optimiser path lives here.
#include <cstdio>
#include <ctime>

#include "../../include/fl_legacy.h"

extern "C" {

// subroutine ShowTime() General.f90:29-33, format (1x,I4,1x,A4,1x,I2,1x,A5,1x,I2,1x,A3,1x,I2,A1,I2,A1,I2)
void __general_MOD_showtime(void)
{
    const std::time_t now = std::time(nullptr);
    std::tm lt;
    localtime_r(&now, &lt);
    std::printf(" %4d year %2d month %2d day %2d:%2d:%2d\n", lt.tm_year + 1900, lt.tm_mon + 1, lt.tm_mday, lt.tm_hour,
                lt.tm_min, lt.tm_sec);
    std::fflush(stdout);
}
void general_mp_showtime_(void) { __general_MOD_showtime(); }

// subroutine dScientificNotation(x,i) General.f90:35-55: repeated *10 / /10 like the source (same rounding);
// like the source it does not terminate for x <= 0 -- guarded here: non-positive or non-finite x is returned as is
void __general_MOD_dscientificnotation(double *x, int *i)
{
    *i = 0;
    if (!(*x > 0.0) || *x > 1.7976931348623157e308) return;
    while (*x < 1.0) {
        *x = *x * 10.0;
        --*i;
    }
    while (*x >= 10.0) {
        *x = *x / 10.0;
        ++*i;
    }
}
void general_mp_dscientificnotation_(double *x, int *i) { __general_MOD_dscientificnotation(x, i); }

} // extern "C"
