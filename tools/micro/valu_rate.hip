// Issue cost of a few VALU instructions on gfx950, one wavefront per SIMD and three per SIMD: cycles per instruction
// by s_memtime over N back-to-back independent (4 chains) and dependent (1 chain) instructions.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/micro/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(x) x x x x x x x x x x x x x x x x
#define N_INNER 64            // 64 x 16 = 1024 instructions per measurement

template <int KIND>
__global__ void k(uint64_t *out, double seed)
{
    double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3;
    float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    const double m = 1.0000001;
    const float mf = 1.0000001f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = { f0, f1 }, p1 = { f2, f3 }, p2 = { f1, f2 }, p3 = { f3, f0 }, pm = { mf, mf };
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_INNER; i++) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(m));) }                 // dependent f64 fma
        if (KIND == 1) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(m));) }
        if (KIND == 2) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f0) : "v"(mf));) }
        if (KIND == 3) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(mf));) }
        if (KIND == 4) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));) }
        if (KIND == 5) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(m));) }
        if (KIND == 6) { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(m));) }
        if (KIND == 7) { REP16(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));) }
        if (KIND == 8) { REP16(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));) }
        if (KIND == 9) { REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (KIND == 10) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (KIND == 11) { REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        if (KIND == 12) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));) }
        if (KIND == 13) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pm));) }         // dependent packed
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (d0 + d1 + d2 + d3 + f0 + f1 + f2 + f3 + p0.x + p1.x + p2.x + p3.x == 12345.678) out[1] = 1;       // keep the values alive
}

template <int KIND>
static void run(const char *name, int per_rep, uint64_t *dev)
{
    for (int waves = 1; waves <= 3; waves += 2) {
        uint64_t h[2] = { 0, 0 };
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(256 * waves), 0, 0, dev, 1.5);          // 4 SIMDs x `waves` wavefronts
        hipMemcpy(h, dev, 16, hipMemcpyDeviceToHost);
        printf("%-28s %d wave(s)/SIMD: %6.2f cycles per instruction per wavefront\n", name, waves, (double)h[0] / (N_INNER * 16.0 * per_rep));
    }
}

int main()
{
    uint64_t *dev;
    hipMalloc(&dev, 64);
    run<0>("v_fma_f64 dependent", 1, dev);
    run<1>("v_fma_f64 4 chains", 4, dev);
    run<2>("v_fma_f32 dependent", 1, dev);
    run<3>("v_fma_f32 4 chains", 4, dev);
    run<4>("v_pk_fma_f32 4 chains", 4, dev);
    run<13>("v_pk_fma_f32 dependent", 1, dev);
    run<12>("v_pk_mul_f32 4 chains", 4, dev);
    run<5>("v_mul_f64 4 chains", 4, dev);
    run<6>("v_add_f64 4 chains", 4, dev);
    run<7>("v_cvt_f64_f32 x4", 4, dev);
    run<8>("v_cvt_f32_f64 x4", 4, dev);
    run<9>("v_rcp_f64 4 chains", 4, dev);
    run<10>("v_rcp_f32 4 chains", 4, dev);
    run<11>("v_sqrt_f32 4 chains", 4, dev);
    return 0;
}
