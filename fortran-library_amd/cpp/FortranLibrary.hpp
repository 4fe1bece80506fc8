// Umbrella header, named like the reference's cpp/FortranLibrary.hpp.  This build of libFL.so covers the
// line-search optimisers of NonlinearOptimization; the other namespaces of the
// reference (General, Mathematics, LinearAlgebra, Chemistry, ...) are out of scope and not declared here.
#ifndef FL_AMD_FortranLibrary_hpp
#define FL_AMD_FortranLibrary_hpp
#include "NonlinearOptimization.hpp"
#endif
