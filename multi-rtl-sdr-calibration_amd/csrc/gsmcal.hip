// gsmcal.hip -- the one translation unit of libgsmcal.so: kernels (kernels_*.h), host plans (host_plan.h), C ABI (abi_*.h).
//
// The calibration chain is enqueued as a fixed sequence of kernels on one HIP stream; all
// data-dependent control lives in StreamState on the device (see state.h).  The same building
// blocks serve the per-function MATLAB-signature entry points (level 0 = a complex array handed in)
// and the batched hot path (level 0 = FIR of the raw bytes, evaluated lazily window by window).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <sys/stat.h>
#include <time.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include "../../include/gsmcal.h"
#include "builtin_taps.h"
#include "kernels_detect.h"
#include "kernels_estim.h"
#include "kernels_frontend.h"
#include "kernels_demod.h"
#include "state.h"

#define GSMCAL_VERSION "gsmcal-mi355x 0.1 (gfx950)"
#define TILE 1024

#include "host_plan.h"
#include "abi_report.h"

// ================================================================================================
// C ABI (include/gsmcal.h)
// ================================================================================================
extern "C" {
#include "abi_calls.h"
#include "abi_comm_ring.h"
#include "abi_util.h"
}  // extern "C"
