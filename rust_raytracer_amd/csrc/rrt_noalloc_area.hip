// Staging area of rrt_render_device_noalloc: the per-sample radiance slab of a render pass, held in a code object of its own so
// that the entry point never asks the runtime for memory - and so that only processes that use that entry point pay for it: the
// runtime loads a code object onto a device at the first access to something in it, and bringing up 128 MiB of zero-initialised
// device data takes about 45 ms.  Each GPU the entry point runs on holds its own copy until the process ends.
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "../../include/rrt.h"
#include "rrt_error.hpp"
#include "rrt_launch.hpp"

namespace {

// In host code the array's NAME is only a handle for hipGetSymbolAddress; kernels, here as everywhere, get the address that call
// returned for the device they run on.
static __device__ __align__(256) unsigned char scratch[RRT_NOALLOC_STAGING_BYTES];

constexpr int kMaxDevices = 64;
std::mutex g_lock;
float* g_area[kMaxDevices] = {};  // per device: where that device's copy of `scratch` lies

}  // namespace

namespace rrt {

// The current device's copy of the staging area.  `device` must be the current device (every GPU holds its own copy of the array at
// its own address, and the look-up answers for the device that is current).  The first call on a device loads the area's code
// object - the one moment at which this path makes the runtime allocate; later calls only read the table.
int noalloc_area(int device, float** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return rrt_set_error(RRT_ERR_LIMIT, "rrt_render_device_noalloc: device " + std::to_string(device) + " is beyond the table of staging areas");
    std::lock_guard<std::mutex> hold(g_lock);
    if (!g_area[device]) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) return rrt_set_error(RRT_ERR_DEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e));
        if (current != device) return rrt_set_error(RRT_ERR_INVALID, "rrt_render_device_noalloc: the staging area of device " + std::to_string(device) + " was asked for while device " + std::to_string(current) + " is current");
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return rrt_set_error(RRT_ERR_DEVICE, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_area[device] = (float*)p;
    }
    *out = g_area[device];
    return RRT_OK;
}

}  // namespace rrt
