// SteepestDescent / ConjugateGradient for 512 < n <= 1024: one wave x 16 elements per thread (fl_solver_launch.hpp)
#include "fl_solver_launch.hpp"
namespace fl {
template hipError_t launch_vec<1, 16>(int, int, const SolveArgs &, hipStream_t);
}
