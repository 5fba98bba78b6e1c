// runtime.hip — error plumbing and device probes of libkai0hip.so (host side only).
#include "common.h"
#include "../../include/kai0hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void kai0_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int kai0_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        kai0_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

KAI0_API const char* kai0_last_error(void) { return g_err; }
KAI0_API int kai0_abi_version(void) { return 1; }
KAI0_API int kai0_gemm_desc_size(void) { return (int)sizeof(kai0_gemm_desc); }

KAI0_API int kai0_device_info(int device, int* n_cu, int* lds_bytes, char* arch_name64) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        kai0_set_error("kai0_device_info: %s", hipGetErrorString(e));
        return -2;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (arch_name64) {
        strncpy(arch_name64, prop.gcnArchName, 63);
        arch_name64[63] = 0;
    }
    return 0;
}
