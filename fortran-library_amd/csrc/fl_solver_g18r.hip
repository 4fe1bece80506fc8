// replicated groups for the one-wave geometry 1 x 8: the augmented-Lagrangian kernels with their objective-only shrink loop
// shared out by trial among 2 / 4 complete copies of the machine (fl_solver_launch.hpp: fl_solve_rep_kernel)
#include "fl_solver_launch.hpp"
namespace fl {
template hipError_t launch_rep<1, 8>(int, int, int, const SolveArgs &, hipStream_t);
}
