// Error plumbing + device info for the C ABI.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void dh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dh_last_error(void) { return g_err; }
extern "C" int dh_abi_version(void) { return DH_ABI_VERSION; }

extern "C" int dh_device_info(char* h_buf, int h_buf_len, int* h_num_cu, int64_t* h_hbm_bytes) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    DH_CHECK(e == hipSuccess && n > 0, "no HIP device visible (%s)", hipGetErrorString(e));
    int dev = 0;
    DH_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    DH_HIP(hipGetDeviceProperties(&p, dev));
    if (h_buf && h_buf_len > 0) {
        strncpy(h_buf, p.gcnArchName, h_buf_len - 1);
        h_buf[h_buf_len - 1] = 0;
    }
    if (h_num_cu) *h_num_cu = p.multiProcessorCount;
    if (h_hbm_bytes) *h_hbm_bytes = (int64_t)p.totalGlobalMem;
    return 0;
}
