/* Host check of an identity k_scan_region's window decode relies on (pigo_amd/csrc/pigo_kernels.hip.inc, reg_split):
 *     (int)((float)f + 0.5f) * inv)  ==  f / nj      for every 0 <= f < 65536, 1 <= nj <= 65535,
 * with inv = 1.0f / (float)nj -- and still with inv off by up to 3 ulp either way, which covers a reciprocal that is not
 * correctly rounded.  The kernel computes exactly these float32 operations (v_cvt_f32_u32, v_add_f32, v_mul_f32, v_cvt_i32_f32:
 * IEEE single precision, round to nearest, truncating conversion; the library is built with -ffp-contract=off), so the
 * correction steps it carried until round 4 could never fire.  All 2^32 pairs per perturbation: a few seconds with OpenMP.
 * Build: gcc -O2 -fopenmp -ffp-contract=off.  Exit status 0 = identity holds. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void)
{
    static const int ulps[] = {-3, 0, 3};
    long long bad = 0;
    for (int d = 0; d < 3; ++d) {
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 256)
        for (int nj = 1; nj <= 65535; ++nj) {
            float inv = 1.0f / (float)nj;
            int32_t bits;
            memcpy(&bits, &inv, 4);
            bits += ulps[d];
            memcpy(&inv, &bits, 4);
            for (uint32_t f = 0; f < 65536u; ++f) {
                volatile float x = (float)f + 0.5f; /* (volatile: one rounding per operation, no excess precision) */
                volatile float p = x * inv;
                if ((int)p != (int)(f / (uint32_t)nj)) ++bad;
            }
        }
    }
    printf("mismatches: %lld\n", bad);
    return bad != 0;
}
