// SteepestDescent / ConjugateGradient: 4 waves x 16 elements per thread (fl_solver_launch.hpp, select_fused_geometry)
#include "fl_solver_launch.hpp"
namespace fl {
template hipError_t launch_vec<4, 16>(int, int, const SolveArgs &, hipStream_t);
}
