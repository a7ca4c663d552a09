/* tools/pose_exact_probe.c -- how often does k_pose's fp64 sin / cos (clap_amd/csrc/pose.hip sincos_halfpi: Taylor
 * polynomials evaluated with FMAs on [0, pi/2]) round to a DIFFERENT float than glibc's sin / cos, the calls the
 * reference's quat_slerp makes (interp.h:107-113)?  The same polynomials, the same FMAs, on the host:
 *     gcc -O2 -ffp-contract=off -o /tmp/pose_exact_probe tools/pose_exact_probe.c -lm && /tmp/pose_exact_probe [samples]
 * sin_theta = (float)sin((double)theta) and _rfac = (float)(cos((double)theta) - (double)u), u a float in [0, 1], for
 * theta a float in [0, pi/2] (half of the samples crowd the top of the interval, where cos cancels).
 * Measured here: 0 differences in 2 * 10^8 samples for both. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 200000000L;
    uint64_t s = 88172645463325252ull;
    long bad_sin = 0, bad_cos = 0;
    for (long i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        float th = (float)((double)(s >> 11) * (1.0 / 9007199254740992.0) * 1.5707964);
        if (i & 1) th = 1.5707964f - th * 0.01f;
        const float u = (float)((double)((s >> 20) & 0xffffff) * (1.0 / 16777216.0));
        double sd, cd;
        sincos_halfpi((double)th, &sd, &cd);
        bad_sin += (float)sd != (float)sin((double)th);
        bad_cos += (float)(cd - (double)u) != (float)(cos((double)th) - (double)u);
    }
    printf("%ld samples: sin rounds differently %ld times, cos - u %ld times\n", n, bad_sin, bad_cos);
    return bad_sin || bad_cos;
}
