// runtime.hip -- device binding, memory helpers and the O(1)-per-frame host math
// (view matrix, projection, frustum planes/corners) of libclapgpu.
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <string>
#include <atomic>
#include "common.h"
#include "lm_dev.h"

#ifdef CLAPGPU_EXPERIMENT               // an A/B or sensitivity build (common.h): never loadable as the product
#define CLAPGPU_ABI_VERSION (32u | 0x80000000u)
#else
#define CLAPGPU_ABI_VERSION 32u
#endif

namespace clapgpu {

static thread_local std::string g_last_error;

int hip_fail(hipError_t err, const char *what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(err);
    (void)hipGetLastError();                       // clear the sticky launch error
    if (err == hipErrorOutOfMemory)
        return CLAPGPU_ERR_NOMEM;
    if (err == hipErrorInvalidValue || err == hipErrorInvalidDevicePointer)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    return CLAPGPU_ERR_UNKNOWN;
}

void set_last_error(const char *what) { g_last_error = what; }

// clapgpu_test_fail_after(): < 0 = off; otherwise the number of launch checks that still pass.  Read by every launch
// check on any thread (the bindings' workers, a frame thread): atomic, and armed only in a process that asked for the
// hook in its environment -- a stray call in a shipping process must not turn a healthy device into permanent host
// fallback.
static std::atomic<int> g_fail_after{-1};

hipError_t launch_error()
{
    const hipError_t e = hipGetLastError();
    int left = g_fail_after.load(std::memory_order_relaxed);
    if (left < 0 || e != hipSuccess) return e;
    while (left > 0)
        if (g_fail_after.compare_exchange_weak(left, left - 1, std::memory_order_relaxed)) return e;
    return left == 0 ? hipErrorLaunchFailure : e;                // injected: the kernel itself ran
}

} // namespace clapgpu

using namespace clapgpu;

extern "C" void clapgpu_test_fail_after(int launches)
{
    static const bool armed = getenv("CLAPGPU_TEST_HOOKS") != nullptr;   // tests/test_dropin.py, oracle/ref/dropin.c `fail` set it
    if (armed || launches < 0) g_fail_after.store(launches, std::memory_order_relaxed);
}

extern "C" uint32_t clapgpu_abi_version(void) { return CLAPGPU_ABI_VERSION; }

extern "C" const char *clapgpu_last_error(void) { return g_last_error.c_str(); }

extern "C" int clapgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int clapgpu_init(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        g_last_error = "no HIP device";
        return CLAPGPU_ERR_INIT_FAILED;
    }
    if (device < 0 || device >= n)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    CLAPGPU_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    CLAPGPU_HIP(hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950")) {     // the code object holds gfx950 ISA only
        g_last_error = std::string("device is ") + prop.gcnArchName + ", libclapgpu is built for gfx950";
        return CLAPGPU_ERR_NOT_SUPPORTED;
    }
    return CLAPGPU_OK;
}

extern "C" int clapgpu_malloc(void **dev, size_t bytes)
{
    if (!dev) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    CLAPGPU_HIP(hipMalloc(dev, bytes ? bytes : 1));
    return CLAPGPU_OK;
}

extern "C" int clapgpu_free(void *dev)
{
    CLAPGPU_HIP(hipFree(dev));
    return CLAPGPU_OK;
}

extern "C" int clapgpu_host_malloc(void **host, size_t bytes)
{
    if (!host) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    CLAPGPU_HIP(hipHostMalloc(host, bytes ? bytes : 1, hipHostMallocDefault));
    return CLAPGPU_OK;
}

// Page-locked host memory the device can address (zero-copy): *dev_alias is the pointer kernels use.  Coherent
// (fine-grained): a kernel's stores are visible to the host once the kernel has completed, no copy, no cache flush call.
extern "C" int clapgpu_host_malloc_mapped(void **host, void **dev_alias, size_t bytes)
{
    if (!host || !dev_alias) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    CLAPGPU_HIP(hipHostMalloc(host, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocCoherent));
    const hipError_t e = hipHostGetDevicePointer(dev_alias, *host, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(*host);
        *host = nullptr;
        return hip_fail(e, "hipHostGetDevicePointer");
    }
    return CLAPGPU_OK;
}

extern "C" int clapgpu_wait_word(const volatile uint32_t *word, uint32_t value, void *stream)
{
    if (!word) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    for (uint32_t spins = 0;; spins++) {
        if (*word == value) return CLAPGPU_OK;
        __builtin_ia32_pause();
        if ((spins & 0xffffu) == 0xffffu) {                      // ~ every millisecond: is anything still running?
            const hipError_t e = hipStreamQuery(as_stream(stream));
            if (e == hipSuccess) {
                if (*word == value) return CLAPGPU_OK;
                g_last_error = "clapgpu_wait_word: the stream drained without the word being stored";
                return CLAPGPU_ERR_UNKNOWN;
            }
            if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery");
        }
    }
}

extern "C" int clapgpu_host_free(void *host)
{
    CLAPGPU_HIP(hipHostFree(host));
    return CLAPGPU_OK;
}

extern "C" int clapgpu_memcpy_h2d(void *dev, const void *host, size_t bytes, void *stream)
{
    CLAPGPU_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return CLAPGPU_OK;
}

extern "C" int clapgpu_memcpy_d2h(void *host, const void *dev, size_t bytes, void *stream)
{
    CLAPGPU_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return CLAPGPU_OK;
}

extern "C" int clapgpu_memset(void *dev, int value, size_t bytes, void *stream)
{
    CLAPGPU_HIP(hipMemsetAsync(dev, value, bytes, as_stream(stream)));
    return CLAPGPU_OK;
}

extern "C" int clapgpu_stream_sync(void *stream)
{
    CLAPGPU_HIP(hipStreamSynchronize(as_stream(stream)));
    return CLAPGPU_OK;
}

// ---------------------------------------------------------------------------
// Host-side per-frame constants.  Same arithmetic as the kernels (lm_dev.h).
// ---------------------------------------------------------------------------

// transform.c:132-138: I * R(quat), transpose the 3x3, translate_in_place(-pos)
extern "C" void clapgpu_view_matrix(const float pos[3], const float quat[4], float view_mx[16])
{
    float m[16], r[16];
    lmd::identity(m);
    lmd::from_quat(r, quat[0], quat[1], quat[2], quat[3]);
    lmd::mul(m, m, r);
    float t[16];
    for (int i = 0; i < 16; i++) t[i] = m[i];
    for (int c = 0; c < 3; c++)                     // linmath.h:408-416
        for (int rr = 0; rr < 3; rr++)
            t[4 * c + rr] = m[4 * rr + c];
    lmd::translate_in_place(t, -pos[0], -pos[1], -pos[2]);
    memcpy(view_mx, t, sizeof(t));
}

// linmath.h:611-651 / 959-987: the kernels' own arithmetic for host callers (the scene loader's bind = invert(invmx),
// model.c:532; a skin's root pose, gltf.c:1246-1252)
extern "C" void clapgpu_mat4_invert(const float m[16], float out[16])
{
    float a[16], t[16];
    memcpy(a, m, sizeof(a));
    lmd::invert(t, a);
    memcpy(out, t, sizeof(t));
}

extern "C" void clapgpu_mat4_from_quat(const float quat_xyzw[4], float out[16])
{
    float t[16];
    lmd::from_quat(t, quat_xyzw[0], quat_xyzw[1], quat_xyzw[2], quat_xyzw[3]);
    memcpy(out, t, sizeof(t));
}

// linmath.h:709-734 (NDC z in [-1,1]) / 753-776 (NDC z in [0,1])
extern "C" void clapgpu_perspective(float fov, float aspect, float n, float f,
                                    int ndc_z_zero_one, float proj_mx[16])
{
    const float a = (float)(1.0 / tan((double)(fov / 2.f)));
    memset(proj_mx, 0, 16 * sizeof(float));
    proj_mx[0] = a / aspect;
    proj_mx[5] = a;
    proj_mx[11] = -1.f;
    if (ndc_z_zero_one) {
        proj_mx[10] = -(f / (f - n));
        proj_mx[14] = -((f * n) / (f - n));
    } else {
        proj_mx[10] = -((f + n) / (f - n));
        proj_mx[14] = -((2.f * f * n) / (f - n));
    }
}

// view.c:248-289
extern "C" void clapgpu_frustum_calc(const float view_mx[16], const float proj_mx[16],
                                     int ndc_z_zero_one, clapgpu_frustum *out)
{
    float v[16], p[16], mvp[16], inv[16];
    memcpy(v, view_mx, sizeof(v));
    memcpy(p, proj_mx, sizeof(p));
    lmd::mul(mvp, p, v);
    lmd::invert(inv, mvp);
    // planes = row3(mvp) +/- row{0,1,2}(mvp)  (view.c:271,275-280 via the transpose)
    for (int k = 0; k < 4; k++) {
        const float r0 = mvp[4 * k + 0], r1 = mvp[4 * k + 1], r2 = mvp[4 * k + 2], r3 = mvp[4 * k + 3];
        out->planes[0][k] = r3 + r0;
        out->planes[1][k] = r3 - r0;
        out->planes[2][k] = r3 + r1;
        out->planes[3][k] = r3 - r1;
        out->planes[4][k] = r3 + r2;
        out->planes[5][k] = r3 - r2;
    }
    const float zn = ndc_z_zero_one ? 0.f : -1.f;  // view.c:252-265
    for (int i = 0; i < 8; i++) {
        const float sx = (i == 0 || i == 3 || i == 4 || i == 7) ? -1.f : 1.f;
        const float sy = ((i & 3) < 2) ? -1.f : 1.f;
        const float c[4] = { sx, sy, i < 4 ? zn : 1.f, 1.f };
        float q[4];
        lmd::mul_vec4(q, inv, c);
        const float s = 1.f / q[3];
        for (int k = 0; k < 4; k++)
            out->corners[i][k] = q[k] * s;
    }
}
