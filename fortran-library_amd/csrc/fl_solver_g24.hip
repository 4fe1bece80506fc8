// NewtonRaphson for 256 < n <= 512: two waves x 4 elements per thread -- its one-workgroup Cholesky wants the threads
// (fl_solver_launch.hpp, select_fused_geometry)
#include "fl_solver_launch.hpp"
namespace fl {
template hipError_t launch_newton<2, 4>(int, int, const SolveArgs &, hipStream_t);
}
