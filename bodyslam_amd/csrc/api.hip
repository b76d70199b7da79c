// Library state: device selection, the zero page used by the implicit-GEMM padding taps, and the
// thread-local error string behind bs_last_error().
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace bs {

static thread_local char g_err[512] = "";
static void* g_zero = nullptr;
static int g_device = -1;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const void* zero_page() { return g_zero; }
bool initialized() { return g_zero != nullptr; }
static int g_cus = 256;
int cu_count() { return g_cus; }

}  // namespace bs

extern "C" int bs_init(int device) {
    using namespace bs;
    if (g_zero && g_device == device) return BS_OK;
    BS_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    BS_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("bs_init: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return BS_ERR_INVALID;
    }
    if (!g_zero) {
        BS_CHECK_HIP(hipMalloc(&g_zero, 4096));
        BS_CHECK_HIP(hipMemset(g_zero, 0, 4096));
    }
    g_device = device;
    g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return BS_OK;
}

extern "C" const char* bs_last_error(void) { return bs::g_err; }
extern "C" int bs_version(void) { return 1; }

extern "C" int bs_copy_f32(const float* src, float* dst, int64_t n, void* stream) {
    using namespace bs;
    if (!initialized()) { set_error("bs_copy_f32: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(src && dst && n >= 0, "bs_copy_f32: bad argument");
    if (n == 0) return BS_OK;
    BS_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)));
    return BS_OK;
}
