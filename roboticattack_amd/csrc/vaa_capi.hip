// vaa_capi.hip — error plumbing and device probing for libvaa_hip.so (include/vaa.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "vaa_common.h"

namespace vaa {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return VAA_E_LAUNCH;
    }
    return VAA_OK;
}

}  // namespace vaa

extern "C" {

const char* vaa_last_error(void) { return vaa::g_err; }

int vaa_version(void) { return 100; /* 0.1.0 */ }

int vaa_device_check(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        vaa::set_error("no HIP device visible");
        (void)hipGetLastError();
        return VAA_E_NODEVICE;
    }
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        vaa::set_error("hipGetDeviceProperties failed");
        return VAA_E_NODEVICE;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        vaa::set_error("device %d is %s; libvaa_hip.so is built for gfx950 only", dev, p.gcnArchName);
        return VAA_E_NODEVICE;
    }
    return VAA_OK;
}

}  // extern "C"
