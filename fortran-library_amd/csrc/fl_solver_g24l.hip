// latency geometry 2 waves x 4 elements per thread: SD / CG / L-BFGS (+ the augmented Lagrangian around CG / L-BFGS) for
// batches that under-fill the chip (fl_solver_launch.hpp: launch_lat; fl_solver_kernels.hip: select_fused_geometry)
#include "fl_solver_launch.hpp"
namespace fl {
template hipError_t launch_lat<2, 4>(int, int, int, const SolveArgs &, hipStream_t);
}
