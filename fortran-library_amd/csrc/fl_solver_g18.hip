// the fused solver kernels of the geometry 1 wave x 8 elements per thread (fl_solver_launch.hpp)
#include "fl_solver_launch.hpp"
namespace fl {
template hipError_t launch_o<1, 8>(int, int, int, const SolveArgs &, hipStream_t);
}
