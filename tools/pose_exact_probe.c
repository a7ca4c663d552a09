/* tools/pose_exact_probe.c -- how often does k_pose's fp64 sin / cos (clap_amd/csrc/pose.hip sincos_halfpi: Taylor
 * polynomials evaluated with FMAs on [0, pi/2]) round to a DIFFERENT float than glibc's sin / cos, the calls the
 * reference's quat_slerp makes (interp.h:107-113)?  The same polynomials, the same FMAs, on the host:
 *     gcc -O2 -fopenmp -ffp-contract=off -o /tmp/pose_exact_probe tools/pose_exact_probe.c -lm
 *     /tmp/pose_exact_probe            the three legs below with their default sizes (minutes on 8 cores)
 *     /tmp/pose_exact_probe <samples>  the sampled legs with that many samples each
 *
 * 1. sin, EXHAUSTIVE.  The slerp's sin_theta = (float)sin((double)theta) takes a FLOAT theta = fac * theta_0 in
 *    [0, pi/2]: 1 070 141 404 bit patterns (0x00000000 .. 0x3fc90fdb).  Every one of them is checked.
 * 2. cos - u, uniform.  _rfac = (float)(cos((double)theta) - (double)u) with theta as above and u ANY float in [0, 1]:
 *    a two-dimensional space, sampled (half of the samples crowd the top of the interval, where cos itself is small).
 * 3. cos - u, as the slerp produces it.  u is not any float: u = dot * sin_theta / sin_theta_0 (fp32), and
 *    cos(theta) - u = sin(theta_0 - theta) / sin(theta_0) CANCELS as fac -> 1.  Where the difference is small its float
 *    ulp is small, and a 1-ulp fp64 disagreement between the polynomial and glibc's cos (neither is correctly rounded)
 *    is no longer far below it.  Sampled with theta_0 uniform over the slerp's range (dot <= 0.9995) and fac uniform in
 *    [0, 1] -- the distribution a game's clock produces -- plus a leg with fac crowded against 1 (1 - 2^-k, k uniform in
 *    [1, 24]) that shows the mechanism.
 * The numbers of the last run are quoted in DESIGN.md section 4 and in pose.hip's header.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void sincos_halfpi(double x, double *sn, double *cs)
{
    const double z = x * x;
    double ps = -1.9572941063391263e-20;
    ps = fma(ps, z, 8.2206352466243295e-18);
    ps = fma(ps, z, -2.8114572543455206e-15);
    ps = fma(ps, z, 7.6471637318198164e-13);
    ps = fma(ps, z, -1.6059043836821613e-10);
    ps = fma(ps, z, 2.5052108385441720e-08);
    ps = fma(ps, z, -2.7557319223985893e-06);
    ps = fma(ps, z, 1.9841269841269841e-04);
    ps = fma(ps, z, -8.3333333333333332e-03);
    ps = fma(ps, z, 1.6666666666666666e-01);
    *sn = fma(-(x * z), ps, x);
    double pc = -8.8967913924505741e-22;
    pc = fma(pc, z, 4.1103176233121648e-19);
    pc = fma(pc, z, -1.5619206968586225e-16);
    pc = fma(pc, z, 4.7794773323873853e-14);
    pc = fma(pc, z, -1.1470745597729725e-11);
    pc = fma(pc, z, 2.0876756987868100e-09);
    pc = fma(pc, z, -2.7557319223985888e-07);
    pc = fma(pc, z, 2.4801587301587302e-05);
    pc = fma(pc, z, -1.3888888888888889e-03);
    pc = fma(pc, z, 4.1666666666666664e-02);
    pc = fma(pc, z, -0.5);
    *cs = fma(z, pc, 1.0);
}

static inline uint64_t xs(uint64_t *s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
static inline double u01(uint64_t *s) { return (double)(xs(s) >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 10000000000L;
    /* ---- 1: every float in [0, pi/2] ---- */
    const uint32_t top = 0x3fc90fdbu;                                /* (float)(pi / 2) */
    long bad_sin = 0, cos_differs_fp64 = 0;
#pragma omp parallel for reduction(+ : bad_sin, cos_differs_fp64) schedule(static)
    for (uint32_t b = 0; b <= top; b++) {
        float th;
        memcpy(&th, &b, 4);
        double sd, cd;
        sincos_halfpi((double)th, &sd, &cd);
        bad_sin += (float)sd != (float)sin((double)th);
        cos_differs_fp64 += cd != cos((double)th);
    }
    printf("sin: all %u floats of [0, pi/2]: (float)sin differs %ld times  (the fp64 cos polynomial differs from glibc's cos in its last bits for %ld of them)\n",
           top + 1, bad_sin, cos_differs_fp64);
    /* ---- 2: cos - u, u any float ---- */
    long bad_cos = 0;
#pragma omp parallel reduction(+ : bad_cos)
    {
        uint64_t s = 88172645463325252ull;
#ifdef _OPENMP
        s += 0x9e3779b97f4a7c15ull * (uint64_t)(omp_get_thread_num() + 1);
#endif
#pragma omp for schedule(static)
        for (long i = 0; i < n; i++) {
            float th = (float)(u01(&s) * 1.5707964);
            if (i & 1) th = 1.5707964f - th * 0.01f;
            const float u = (float)((double)((xs(&s) >> 20) & 0xffffff) * (1.0 / 16777216.0));
            double sd, cd;
            sincos_halfpi((double)th, &sd, &cd);
            bad_cos += (float)(cd - (double)u) != (float)(cos((double)th) - (double)u);
        }
    }
    printf("cos - u, u any float in [0, 1]: %ld samples, rounds differently %ld times\n", n, bad_cos);
    /* ---- 3: cos - u as quat_slerp forms it (interp.h:104-113) ---- */
    for (int leg = 0; leg < 2; leg++) {
        long flips = 0, small = 0;
#pragma omp parallel reduction(+ : flips, small)
        {
            uint64_t s = 0x2545f4914f6cdd1dull;
#ifdef _OPENMP
            s += 0x9e3779b97f4a7c15ull * (uint64_t)(omp_get_thread_num() + 7);
#endif
#pragma omp for schedule(static)
            for (long i = 0; i < n; i++) {
                const float dot = (float)(u01(&s) * 0.9995);                 /* the slerp branch: dot <= 0.9995 (after the sign flip) */
                const float theta_0 = (float)acos((double)dot);
                float fac;
                if (leg == 0) fac = (float)u01(&s);
                else fac = 1.0f - (float)ldexp(u01(&s), -(int)(1 + (xs(&s) >> 40) % 24));
                const float theta = fac * theta_0;
                const float sin_theta = (float)sin((double)theta), sin_theta_0 = (float)sin((double)theta_0);
                const float u = dot * sin_theta / sin_theta_0;
                double sd, cd;
                sincos_halfpi((double)theta, &sd, &cd);
                const float ref = (float)(cos((double)theta) - (double)u), got = (float)(cd - (double)u);
                flips += ref != got;
                small += fabsf(ref) < 1e-4f;
            }
        }
        printf("cos - u as the slerp forms it, fac %s: %ld samples (%ld with |_rfac| < 1e-4), _rfac rounds differently %ld times\n",
               leg == 0 ? "uniform in [0, 1]" : "crowded against 1 (1 - 2^-k * U)", n, small, flips);
    }
    return bad_sin != 0;
}
