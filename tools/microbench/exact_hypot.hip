// exact_hypot.hip - a float-only hypotf that returns the reference's bits.  NOT USED by the library:
// it is exact (0 mismatches in 2.6e10 pairs with its fallback) but no faster than the double-precision
// expression on MI355X, where v_fma_f64 issues at the rate of v_fma_f32 (csg1024 4K: 9.8 ms per frame
// with the double hypotf, 11.2 ms with this one and its per-call fallback branch).  Kept as the record
// of the experiment.
//   reference  (float)sqrt((double)x*x + (double)y*y)          (Include/cuda_defines.cuh hypotf)
//   candidate  squares and their sum as float pairs (error-free transforms), one Newton step on
//              v_rsq_f32, and a test that sends the cases a rounding boundary is too close to call
//              to the reference.
// Random pairs (all exponent gaps), special values, and adversarial pairs built to land the root next
// to the midpoint of two floats.  Prints mismatches (must be 0), the share of lanes that fall back and
// the largest error of the approximate root in units of the result's ulp.
//   hipcc -O3 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -o exact_hypot exact_hypot.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ float hypot_ref(float x, float y) {
    return (float)sqrt((double)x * (double)x + (double)y * (double)y);
}

// returns the candidate; *hard: this lane must take the reference; *r2out: v - rf
__device__ __forceinline__ float hypot_short(float x, float y, bool* hard, float* r2out, float band) {
    const float hx = x * x, hy = y * y;
    const float lx = __builtin_fmaf(x, x, -hx), ly = __builtin_fmaf(y, y, -hy);      // x*x = hx + lx exactly
    const float a = __builtin_fmaxf(hx, hy), b = __builtin_fminf(hx, hy);
    const float sh = a + b;
    const float e = b - (sh - a);                                                      // a + b = sh + e exactly (a >= b >= 0)
    const float sl = e + (lx + ly);
    const float r = __builtin_amdgcn_rsqf(sh);
    const float g = sh * r, h = 0.5f * r;
    const float d = __builtin_fmaf(-g, g, sh) + sl;                                    // S - g*g
    const float rf = __builtin_fmaf(d, h, g);
    const float r2 = __builtin_fmaf(d, h, g - rf);                                     // (g + d*h) - rf
    const float ulp = as_f((as_u(rf) & 0x7f800000u) - (23u << 23));
    const bool range = as_u(sh) - 0x17800000u >= 0x66800000u;                         // sh outside [2^-80, 2^126): also 0, inf, NaN
    const bool pow2 = (as_u(rf) & 0x007fffffu) == 0;
    const bool close = fabsf(fabsf(r2) - 0.5f * ulp) < band * ulp;
    *hard = range || pow2 || close;
    *r2out = r2;
    return rf;
}

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// out: [0] mismatches without fallback, [1] mismatches with fallback, [2] fallbacks, [3] samples, [4] max error (float bits, in ulp * 2^40)
__global__ void check(uint64_t seed, uint64_t per_thread, int mode, float band, unsigned long long* out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad0 = 0, bad1 = 0, fb = 0, n = 0, maxerr = 0;
    for (uint64_t k = 0; k < per_thread; ++k) {
        const uint64_t z = mix(seed + tid * per_thread + k), z2 = mix(z);
        float x, y;
        if (mode == 0) {            // random mantissas, exponent of x in [-30, 30], gap 0..40
            const int ex = 127 - 30 + (int)((z >> 23) % 61), gap = (int)((z >> 40) % 41);
            x = as_f(((uint32_t)z & 0x7fffffu) | ((uint32_t)ex << 23));
            const int ey = ex - gap < 1 ? 1 : ex - gap;
            y = as_f(((uint32_t)z2 & 0x7fffffu) | ((uint32_t)ey << 23));
            if (z2 >> 63) { float t = x; x = y; y = t; }
            if ((z2 >> 62) & 1) x = -x;
        } else if (mode == 1) {     // any two bit patterns
            x = as_f((uint32_t)z); y = as_f((uint32_t)z2);
        } else {                    // the root next to the midpoint of two floats
            const int em = 127 - 20 + (int)((z >> 23) % 41);
            const float m = as_f(((uint32_t)z & 0x7fffffu) | ((uint32_t)em << 23));
            const double ulp = ldexp(1.0, em - 127 - 23);
            const double t = (double)m + 0.5 * ulp * (1.0 + ldexp((double)((int)((z2 >> 8) & 0xff) - 128), -20 - (int)((z2 >> 20) % 30)));
            const float fx = (float)(t * ((double)((z2 >> 32) & 0xffffff) / 16777216.0));
            const double rest = t * t - (double)fx * (double)fx;
            x = fx; y = (float)sqrt(rest > 0 ? rest : 0.0);
        }
        bool hard; float r2;
        const float got = hypot_short(x, y, &hard, &r2, band);
        const float want = hypot_ref(x, y);
        const bool same = as_u(got) == as_u(want) || (got != got && want != want);
        ++n;
        if (!same) ++bad0;
        if (hard) ++fb; else if (!same) ++bad1;
        if (!hard) {
            const double V = sqrt((double)x * (double)x + (double)y * (double)y);
            const double v = (double)got + (double)r2;
            const double ulp = (double)as_f((as_u(got) & 0x7f800000u) - (23u << 23));
            const double err = fabs(v - V) / ulp * 1099511627776.0;          // ulp * 2^40
            const unsigned long long q = err > 1.8e19 ? ~0ull : (unsigned long long)err;
            if (q > maxerr) maxerr = q;
        }
    }
    atomicAdd(&out[0], bad0); atomicAdd(&out[1], bad1); atomicAdd(&out[2], fb); atomicAdd(&out[3], n);
    atomicMax(&out[4], maxerr);
}

int main(int argc, char** argv) {
    const float band = argc > 1 ? (float)atof(argv[1]) : 1.0f / 16384.0f;
    const uint64_t per_thread = argc > 2 ? strtoull(argv[2], nullptr, 0) : 4096;
    unsigned long long* d; unsigned long long h[5];
    hipMalloc(&d, sizeof h);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(d, 0, sizeof h);
        hipLaunchKernelGGL(check, dim3(256 * 8), dim3(256), 0, 0, 0x1234567ull + mode, per_thread, mode, band, d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("mode %d  band %g ulp: samples %llu  wrong without fallback %llu  WRONG WITH FALLBACK %llu  fallbacks %llu (%.3g)  max error of the root %.3g ulp\n",
               mode, band, h[3], h[0], h[1], h[2], (double)h[2] / (double)h[3], (double)h[4] / 1099511627776.0);
    }
    return 0;
}
