// api.hip -- library-level entry points: error text, ABI version, device summary.
#include "skr_common.h"

#include <cstdarg>

#define SKR_ABI_VERSION 9

namespace skr {
static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
int max_row_len(const int64_t* d_rowptr, int n_rows, int* d_scratch, hipStream_t st, int* out);
}  // namespace skr

extern "C" {

int skr_abi_version(void) { return SKR_ABI_VERSION; }
const char* skr_last_error(void) { return skr::g_err.c_str(); }

int skr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int skr_device_summary(char* buf, size_t n) {
    SKR_REQUIRE(buf && n > 0, "skr_device_summary: NULL buffer");
    int dev = 0;
    SKR_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    SKR_HIP(hipGetDeviceProperties(&p, dev));
    snprintf(buf, n, "%s %s CUs=%d clock=%dMHz hbm=%.1fGiB lds/block=%zuB", p.name, p.gcnArchName, p.multiProcessorCount,
             p.clockRate / 1000, static_cast<double>(p.totalGlobalMem) / (1024.0 * 1024.0 * 1024.0), p.sharedMemPerBlock);
    return SKR_OK;
}

int skr_csr_max_row_len(const int64_t* d_rowptr, int n_rows, int* out_max, void* stream) {
    SKR_REQUIRE(d_rowptr && out_max && n_rows >= 0, "skr_csr_max_row_len: bad argument");
    if (n_rows == 0) {
        *out_max = 0;
        return SKR_OK;
    }
    int* d_scratch = nullptr;
    SKR_HIP(hipMalloc(&d_scratch, sizeof(int)));
    int rc = skr::max_row_len(d_rowptr, n_rows, d_scratch, skr::as_stream(stream), out_max);
    (void)hipFree(d_scratch);
    return rc;
}

}  // extern "C"
