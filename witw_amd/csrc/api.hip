// C-ABI plumbing shared by every entry point: thread-local error text, version, device probe.
#include "common.h"
#include <string.h>
#include <atomic>

static thread_local char g_err[512] = "";

void witw_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_variant[128] = "";

void witw_note_variant(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_variant, sizeof(g_variant), fmt, ap);
    va_end(ap);
}

// Compute units of the CURRENT device (256 on MI355X), looked up once PER DEVICE (a process may drive several: the
// --single-device rehearsals, mixed parts); 256 when the query fails. The cache slots are relaxed atomics: two threads that
// race on a slot both store the same value.
int witw_cu_count() {
    constexpr int MAXDEV = 64;
    static std::atomic<int> cache[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;
    if (dev < MAXDEV) {
        const int hit = cache[dev].load(std::memory_order_relaxed);
        if (hit > 0) return hit;
    }
    hipDeviceProp_t prop;
    const int n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    if (dev < MAXDEV) cache[dev].store(n, std::memory_order_relaxed);
    return n;
}

// Does a grid of `workgroups` one-per-CU workgroups (the 8-wave conv tiles: 256 registers per wave, one workgroup per CU) use the
// chip well? Yes from two full rounds on (the tail is then at most a third of the launch), and below that when its LAST round is at
// least 90 % full -- e.g. exactly one workgroup per CU: the 16 x 64 maps of the trunk at the reference's default batch of 32
// (model/cvig_semantic.py:416), which round 3's ">= 512" rule sent to the 4-wave kernels.
bool witw_fills_rounds(long long workgroups) {
    const long long cu = witw_cu_count();
    if (workgroups >= 2 * cu) return true;
    if (workgroups <= 0) return false;
    const long long rounds = (workgroups + cu - 1) / cu;
    return 10 * workgroups >= 9 * cu * rounds;
}

extern "C" {

const char* witw_last_error(void) { return g_err; }

// Name of the kernel instantiation the calling thread's most recent conv launcher picked (template arguments as in the
// rocprof kernel names), "" before the first launch. Lets a parity test assert WHICH kernel it compared with the oracle.
const char* witw_last_kernel_variant(void) { return g_variant; }

int witw_version(void) { return 100; }  // 0.1.0

// 0 when device `dev` exists and is a gfx950 part; the message says what was found otherwise.
int witw_device_check(int dev) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        witw_set_error("no HIP device visible (hipGetDeviceCount: %s, count %d)", hipGetErrorString(e), n);
        return WITW_ERR_NODEVICE;
    }
    if (dev < 0 || dev >= n) {
        witw_set_error("device %d out of range (%d visible)", dev, n);
        return WITW_ERR_INVALID;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        witw_set_error("hipGetDeviceProperties failed");
        return WITW_ERR_NODEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        witw_set_error("device %d is %s; this library carries gfx950 code only", dev, prop.gcnArchName);
        return WITW_ERR_NODEVICE;
    }
    return WITW_OK;
}

}  // extern "C"
