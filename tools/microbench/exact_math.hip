// exact_math.hip - exhaustive checks of the short correctly-rounded sqrt and
// constant-divisor division sequences of device_math.h against the compiler's
// IEEE sqrtf and '/' on gfx950, over every float bit pattern.
//   hipcc -O3 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -o exact_math exact_math.hip
//   ./exact_math [divisor ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cmath>
#include <vector>

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

// candidate: one Newton step on the hardware reciprocal square root, residual by fma
__device__ __forceinline__ float sqrt_short(float x) {
    float r = __builtin_amdgcn_rsqf(x);
    float g = x * r;
    float h = 0.5f * r;
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
// candidate: quotient by a constant whose correctly rounded reciprocal is known
__device__ __forceinline__ float div_short(float a, float c, float rc) {
    float q = a * rc;
    float e = __builtin_fmaf(-c, q, a);
    return __builtin_fmaf(e, rc, q);
}

// out[0..255]: mismatches by biased exponent of the input; out[256]: total
__global__ void check_sqrt(unsigned long long* out) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < 0x7f800000ull; i += stride) {
        const float x = as_f((uint32_t)i);
        const float want = sqrtf(x), got = sqrt_short(x);
        if (as_u(want) != as_u(got)) {
            atomicAdd(&out[(uint32_t)i >> 23], 1ull);
            atomicAdd(&out[256], 1ull);
        }
    }
}

__global__ void check_div(float c, float rc, unsigned long long* out) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < 0x100000000ull; i += stride) {
        const float a = as_f((uint32_t)i);
        const float want = a / c, got = div_short(a, c, rc);
        const bool both_nan = want != want && got != got;
        if (as_u(want) != as_u(got) && !both_nan) {
            atomicAdd(&out[((uint32_t)i >> 23) & 0xff], 1ull);
            atomicAdd(&out[256], 1ull);
        }
    }
}

// device sqrtf / division handed to the host for a cross-check against the host's libm / IEEE '/'
__global__ void dump_sqrt(uint32_t first, uint32_t n, float* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sqrtf(as_f(first + i));
}

static void report(const char* what, const std::vector<unsigned long long>& h) {
    printf("%s: %llu mismatches\n", what, h[256]);
    if (!h[256]) return;
    int lo = 256, hi = -1;
    for (int e = 0; e < 256; ++e) if (h[e]) { if (e < lo) lo = e; if (e > hi) hi = e; }
    printf("  biased exponents with mismatches: %d .. %d\n", lo, hi);
    for (int e = 0; e < 256; ++e) if (h[e]) printf("  exp %3d (2^%d): %llu\n", e, e - 127, h[e]);
}

int main(int argc, char** argv) {
    unsigned long long* d; hipMalloc(&d, 257 * 8);
    std::vector<unsigned long long> h(257);
    hipMemset(d, 0, 257 * 8);
    hipLaunchKernelGGL(check_sqrt, dim3(256 * 8), dim3(256), 0, 0, d);
    hipMemcpy(h.data(), d, 257 * 8, hipMemcpyDeviceToHost);
    report("sqrt_short vs sqrtf, all non-negative finite floats", h);

    // device sqrtf against the host's (correctly rounded) sqrtf: 2^26 inputs across four binades + denormals
    {
        const uint32_t n = 1u << 26;
        float* dd; hipMalloc(&dd, (size_t)n * 4);
        std::vector<float> got(n);
        unsigned long long bad = 0;
        for (uint32_t first : {0x3f000000u, 0x00000000u, 0x7b800000u}) {
            hipLaunchKernelGGL(dump_sqrt, dim3(n / 256), dim3(256), 0, 0, first, n, dd);
            hipMemcpy(got.data(), dd, (size_t)n * 4, hipMemcpyDeviceToHost);
            for (uint32_t i = 0; i < n; ++i) {
                uint32_t u = first + i; float x; memcpy(&x, &u, 4);
                float w = sqrtf(x);
                if (memcmp(&w, &got[i], 4)) ++bad;
            }
        }
        printf("device sqrtf vs host sqrtf on 3 x 2^26 inputs: %llu mismatches\n", bad);
        hipFree(dd);
    }

    std::vector<float> divisors = {260000.f, 10000.f, 54.f, 2.5f, 1.5f, 0.7f, 3.f, 1e-3f, 123456.789f, 7.f, 0.1f};
    for (int i = 1; i < argc; ++i) divisors.push_back(strtof(argv[i], nullptr));
    for (float c : divisors) {
        const float rc = 1.0f / c;
        uint32_t cb; memcpy(&cb, &c, 4);
        hipMemset(d, 0, 257 * 8);
        hipLaunchKernelGGL(check_div, dim3(256 * 8), dim3(256), 0, 0, c, rc, d);
        hipMemcpy(h.data(), d, 257 * 8, hipMemcpyDeviceToHost);
        char what[128];
        snprintf(what, sizeof what, "div_short by %.9g (bits 0x%08x, mantissa all ones: %s), all 2^32 numerators", c, cb,
                 (cb & 0x7fffff) == 0x7fffff ? "yes" : "no");
        report(what, h);
    }
    return 0;
}
