// Umbrella header, named like the reference's cpp/FortranLibrary.hpp.  This build of libFL.so covers the
// line-search optimisers of NonlinearOptimization; plus the two helper
// namespaces the reference's umbrella header itself declares (General::dScientificNotation, LA::My_dgemm_T, LA::My_dsyev).
#ifndef FL_AMD_FortranLibrary_hpp
#define FL_AMD_FortranLibrary_hpp
#include "NonlinearOptimization.hpp"

namespace FL {
// the two helper namespaces of the reference's umbrella header (cpp/FortranLibrary.hpp:52-63)
namespace General {
inline void dScientificNotation(double &x, int &i) { __general_MOD_dscientificnotation(&x, &i); }
} // namespace General
namespace LA {
inline void My_dgemm_T(double *A, double *B, double *C, const int &M, const int &K, const int &N)
{
    __linearalgebra_MOD_my_dgemm_t(A, B, C, &M, &K, &N);
}
inline void My_dsyev(const char &jobtype, double *A, double *eigval, const int &N)
{
    __linearalgebra_MOD_my_dsyev(&jobtype, A, eigval, &N, 1);
}
} // namespace LA
} // namespace FL
#endif
