// tools/stream_mix_probe.hip -- what does this box sustain for a kernel that only streams R bytes in and W bytes out?
// The byte mixes of the hot kernels (entity update at 1 M / 2 M / 4 M entities: 48 r + 168 w MB per million; skinning:
// 670 r + 244 w; particles: 107 r + 52 w; a pure fill and a pure copy for reference), each as one launch of a
// grid-stride kernel with 16-byte accesses (non-temporal stores, as the hot kernels use) timed with HIP events.
//     hipcc -O3 --offload-arch=gfx950 tools/stream_mix_probe.hip -o /tmp/stream_mix_probe && /tmp/stream_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

// variant 0: plain grid-stride, one float4 per thread and round; variant 1: four independent float4 per thread and round
// (a workgroup covers 16 KiB contiguous bytes per round), loads issued together before they are consumed
template <int UNROLL>
__global__ __launch_bounds__(256) void k_mix(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n_in, size_t n_out, int nt)
{
    const size_t nthr = (size_t)gridDim.x * blockDim.x;
    const size_t base = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
    f4 acc = { 0, 0, 0, 0 };
    for (size_t i = base; i < n_in; i += nthr * UNROLL) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = i + (size_t)u * 256 < n_in ? in[i + (size_t)u * 256] : acc;
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    for (size_t i = base; i < n_out; i += nthr * UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const size_t x = i + (size_t)u * 256;
            if (x < n_out) {
                const f4 v = acc + (float)x;
                if (nt) __builtin_nontemporal_store(v, &out[x]); else out[x] = v;
            }
        }
    }
}

static double run(const f4 *in, f4 *out, size_t rb, size_t wb, int blocks, int nt, int variant)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ms;
    for (int it = 0; it < 25; it++) {
        hipEventRecord(a);
        if (variant) hipLaunchKernelGGL(k_mix<4>, dim3(blocks), dim3(256), 0, 0, in, out, rb / 16, wb / 16, nt);
        else         hipLaunchKernelGGL(k_mix<1>, dim3(blocks), dim3(256), 0, 0, in, out, rb / 16, wb / 16, nt);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        if (it >= 5) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2] * 1e-3;
}

int main()
{
    const size_t cap = (size_t)1 << 30;
    f4 *in, *out;
    if (hipMalloc(&in, cap) != hipSuccess || hipMalloc(&out, cap) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(in, 0, cap); hipMemset(out, 0, cap);
    struct { const char *name; double r_mb, w_mb; } mixes[] = {
        { "entities 1 M  (48 r + 168 w MB)", 48.3, 168.2 }, { "entities 2 M  (96 r + 337 w MB)", 96.4, 336.6 },
        { "entities 4 M  (193 r + 673 w MB)", 192.5, 673.2 }, { "skinning     (670 r + 244 w MB)", 670.2, 244.4 },
        { "particles    (107 r + 52 w MB)", 107.0, 51.7 }, { "pose         (9 r + 384 w MB)", 8.7, 384.0 },
        { "fill 216 MB", 0.0, 216.0 }, { "copy 108 + 108 MB", 108.0, 108.0 }, { "fill 1 GB", 0.0, 1000.0 },
    };
    printf("%-36s %8s %8s %10s %10s\n", "byte mix", "blocks", "stores", "us", "TB/s");
    for (auto &m : mixes) {
        const size_t rb = (size_t)(m.r_mb * 1e6) / 16 * 16, wb = (size_t)(m.w_mb * 1e6) / 16 * 16;
        double best = 1e9; int bb = 0, bn = 0, bv = 0;
        for (int blocks : { 512, 1024, 2048, 4096, 8192, 16384, 65536 })
            for (int nt = 0; nt < 2; nt++)
                for (int variant = 0; variant < 2; variant++) {
                    const double s = run(in, out, rb, wb, blocks, nt, variant);
                    if (s < best) { best = s; bb = blocks; bn = nt; bv = variant; }
                }
        printf("%-36s %8d %8s %10.1f %10.2f\n", m.name, bb, bn ? (bv ? "nt x4" : "nt") : (bv ? "plain x4" : "plain"), best * 1e6,
               (rb + wb) / best / 1e12);
    }
    return 0;
}
