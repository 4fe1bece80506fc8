// fl_host.hpp -- host-side helpers shared by the translation units of libFL.so
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/fl_nlopt.h"

namespace fl {

// status of the launch(es) just issued on this thread: a failed launch (too much LDS for the device, an invalid
// grid, a missing code object for the GPU found) is FL_ERR_LAUNCH -- not to be confused with "no GPU at all"
// (FL_ERR_NO_DEVICE, decided by hipGetDeviceCount before anything is launched).  hipGetLastError also clears the
// sticky error, so one bad call does not fail the next one.
static inline int launch_status(hipError_t e) { return e == hipSuccess ? FL_OK : FL_ERR_LAUNCH; }
static inline int launch_status() { return launch_status(hipGetLastError()); }

} // namespace fl
