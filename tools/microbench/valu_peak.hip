// valu_peak.hip - what one SIMD of gfx950 issues per cycle for the instruction
// kinds the march kernel is made of (fp32, no contraction).  Build:
//   hipcc -O3 --offload-arch=gfx950 -o valu_peak valu_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x
#define N_ITER 2048

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float seed, unsigned long long* cyc) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = seed * 0.5f;
    const float2_ c2 = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N_ITER; ++i) {
        if (KIND == 0) {        // v_add_f32, 8 independent chains, 8 instr per body x8
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (KIND == 1) { // v_pk_add_f32
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
        } else if (KIND == 2) { // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (KIND == 3) { // v_sqrt_f32
            REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                              "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (KIND == 4) { // one dependent chain of v_add_f32
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n"
                              "v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (KIND == 5) { // v_pk_mul_f32 with an SGPR pair operand
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(c2));)
        } else if (KIND == 6) { // v_cndmask + v_cmp pairs
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// generic: 8 independent copies of one instruction text with operands %0..%7 (vgpr), %8 (vgpr const), %9 (sgpr const)
#define GEN(NAME, TXT) \
__global__ void __launch_bounds__(256) NAME(float* out, float seed, unsigned long long* cyc) { \
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    const float c = seed * 0.5f; float sc = seed * 0.25f; \
    unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int i = 0; i < N_ITER; ++i) { \
        REP8(asm volatile(TXT(0) TXT(1) TXT(2) TXT(3) TXT(4) TXT(5) TXT(6) TXT(7) \
             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(sc) : "vcc", "s20", "s21", "s22", "s23");) \
    } \
    unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; \
}
#define T_MUL_LIT(n)  "v_mul_f32 %" #n ", 0x3f1ce861, %" #n "\n"
#define T_ADD_SGPR(n) "v_add_f32 %" #n ", %9, %" #n "\n"
#define T_FMAC(n)     "v_fmac_f32 %" #n ", %8, %8\n"
#define T_FMA_LIT(n)  "v_fma_f32 %" #n ", %" #n ", %8, 1.0\n"
#define T_MAX(n)      "v_max_f32 %" #n ", %" #n ", %8\n"
#define T_MED3(n)     "v_med3_f32 %" #n ", %" #n ", %8, %8\n"
#define T_CMP32(n)    "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define T_CMP64(n)    "v_cmp_lt_f32 s[20:21], %" #n ", %8\n"
#define T_CND32(n)    "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define T_RSQ(n)      "v_rsq_f32 %" #n ", %" #n "\n"
#define T_FLOOR(n)    "v_floor_f32 %" #n ", %" #n "\n"
#define T_AND(n)      "v_and_b32 %" #n ", 0x7fffffff, %" #n "\n"
#define T_ADDABS(n)   "v_add_f32_e64 %" #n ", |%" #n "|, %9\n"
#define T_SUBNEG(n)   "v_sub_f32 %" #n ", %8, %" #n "\n"
#define T_MOV(n)      "v_mov_b32 %" #n ", %8\n"
#define T_ADDU(n)     "v_add_u32 %" #n ", 0x80800000, %" #n "\n"
#define T_CMPU(n)     "v_cmp_gt_u32 vcc, %9, %" #n "\n"
#define T_MAX3(n)     "v_max3_f32 %" #n ", %" #n ", %8, %8\n"
#define T_MAX3N(n)    "v_max3_f32 %" #n ", %" #n ", -%8, -%8\n"
#define T_MIN3U(n)    "v_min3_u32 %" #n ", %" #n ", %8, %8\n"
#define T_CNDS(n)     "v_cndmask_b32 %" #n ", %" #n ", %8, s[20:21]\n"
#define T_CMPCND(n)   "v_cmp_lt_f32 vcc, %" #n ", %8\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define T_MULLEG(n)   "v_mul_legacy_f32 %" #n ", %" #n ", %8\n"
#define T_MAXLIT(n)   "v_max_f32 %" #n ", 0, %" #n "\n"
#define T_OR(n)       "v_or_b32 %" #n ", %" #n ", %8\n"
#define T_CVT(n)      "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define T_CMPS(n)     "v_cmp_lt_f32 s[22:23], %" #n ", %8\n"
#define T_CMPX(n)     "v_cmp_class_f32 vcc, %" #n ", %8\n"
#define T_MULADD(n)   "v_mul_f32 %" #n ", %" #n ", %8\n v_max_f32 %" #n ", %" #n ", %8\n"
#define T_FMAK(n)     "v_fmaak_f32 %" #n ", %" #n ", %8, 0x3f1ce861\n"
#define T_MADMK(n)    "v_fmamk_f32 %" #n ", %" #n ", 0x3f1ce861, %8\n"
#define T_FMA3(n)     "v_fma_f32 %" #n ", %" #n ", %8, %" #n "\n"
#define T_FMAS(n)     "v_fma_f32 %" #n ", %" #n ", %9, %8\n"
GEN(g_addu, T_ADDU) GEN(g_cmpu, T_CMPU) GEN(g_max3, T_MAX3) GEN(g_max3n, T_MAX3N) GEN(g_min3u, T_MIN3U) GEN(g_cnds, T_CNDS)
GEN(g_cmpcnd, T_CMPCND) GEN(g_mulleg, T_MULLEG) GEN(g_maxlit, T_MAXLIT) GEN(g_or, T_OR) GEN(g_cvt, T_CVT) GEN(g_cmps, T_CMPS)
GEN(g_cmpx, T_CMPX) GEN(g_muladd, T_MULADD) GEN(g_fmak, T_FMAK) GEN(g_madmk, T_MADMK) GEN(g_fma3, T_FMA3) GEN(g_fmas, T_FMAS)
GEN(g_mul_lit, T_MUL_LIT) GEN(g_add_sgpr, T_ADD_SGPR) GEN(g_fmac, T_FMAC) GEN(g_fma_lit, T_FMA_LIT) GEN(g_max, T_MAX)
GEN(g_med3, T_MED3) GEN(g_cmp32, T_CMP32) GEN(g_cnd32, T_CND32) GEN(g_rsq, T_RSQ) GEN(g_floor, T_FLOOR)
GEN(g_and, T_AND) GEN(g_addabs, T_ADDABS) GEN(g_sub, T_SUBNEG) GEN(g_mov, T_MOV)

// scalar instructions alone, and interleaved with vector ones (K_V vector additions per scalar one): does a wave's
// scalar work hide under the vector work of the other waves of its SIMD?  256 scalar + 256 * K_V vector per body.
template <int K_V, int K_S>
__global__ void __launch_bounds__(256) mix(float* out, float seed, unsigned long long* cyc, unsigned useed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const float c = seed * 0.5f;
    unsigned s0 = useed, s1 = useed + 1, s2 = useed + 2, s3 = useed + 3;
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (K_V >= 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c));
            if (K_S >= 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");
            if (K_V >= 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(c));
            if (K_S >= 2) asm volatile("s_and_b32 %0, %0, 0xffffff" : "+s"(s1) : : "scc");
            if (K_V >= 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(c));
            if (K_S >= 3) asm volatile("s_add_u32 %0, %0, 5" : "+s"(s2) : : "scc");
            if (K_V >= 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(c));
            if (K_S >= 4) asm volatile("s_or_b32 %0, %0, 1" : "+s"(s3) : : "scc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) out[blockIdx.x * 256] += (float)(s0 + s1 + s2 + s3);
    if (threadIdx.x == 0) cyc[blockIdx.x] = 0;
}
template <int K_V, int K_S>
void run_mix(int waves_per_simd, float* d_out, unsigned long long* d_cyc) {
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mix<K_V, K_S>), dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, d_cyc, 7u);
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix<K_V, K_S>), dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, d_cyc, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = 16.0 * N_ITER;                       // bodies per wave
    const double waves = blocks * 4.0;
    // cycles per SIMD at 2.4 GHz per body (K_V vector + K_S scalar instructions of one wave)
    const double simd_cycles = ms * 1e-3 * 2.4e9;
    printf("mix %d vector + %d scalar per body, waves/SIMD %d: %8.3f ms  -> %.2f SIMD-cycles (at 2.4 GHz) per body per wave\n",
           K_V, K_S, waves_per_simd, ms, simd_cycles / (per_wave * waves_per_simd));
    (void)waves;
}

typedef void (*kern_t)(float*, float, unsigned long long*);
void run_gen(const char* name, kern_t kfn, int waves_per_simd, float* d_out, unsigned long long* d_cyc) {
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, d_cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double total_instr = 64.0 * N_ITER * blocks * 4;
    printf("%-14s waves/SIMD %d  %8.3f ms  %.1f Tinstr-lanes/s  (cost vs v_add_f32 at 63: %.2f)\n", name, waves_per_simd, ms,
           total_instr * 64 / (ms * 1e-3) / 1e12, 63.0 / (total_instr * 64 / (ms * 1e-3) / 1e12));
}

template <int KIND>
void run(const char* name, int waves_per_simd, float* d_out, unsigned long long* d_cyc, double flops_per_instr) {
    const int blocks = 256 * waves_per_simd;      // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, d_cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> cyc(blocks);
    hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : cyc) mean += c; mean /= blocks;
    const double instr = 64.0 * N_ITER;           // per wave
    const double total_instr = instr * blocks * 4;
    printf("%-14s waves/SIMD %d  %8.3f ms  cycles/instr/wave %.2f  => SIMD issue %.2f cyc/instr  %.1f Tinstr-lanes/s  %.1f TFLOP/s  clock %.2f GHz\n",
           name, waves_per_simd, ms, mean / instr, mean / instr / waves_per_simd, total_instr * 64 / (ms * 1e-3) / 1e12,
           total_instr * 64 * flops_per_instr / (ms * 1e-3) / 1e12, mean / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    float* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_out, 256 * 8 * 256 * 4); hipMalloc(&d_cyc, 256 * 8 * 8);
    if (argc > 1 && argv[1][0] == 's') {     // scalar / vector mixes
        for (int w : {4, 8}) {
            run_mix<4, 0>(w, d_out, d_cyc); run_mix<0, 4>(w, d_out, d_cyc); run_mix<4, 1>(w, d_out, d_cyc);
            run_mix<4, 2>(w, d_out, d_cyc); run_mix<4, 4>(w, d_out, d_cyc); run_mix<2, 4>(w, d_out, d_cyc);
        }
        return 0;
    }
    if (argc > 1) {          // round 2: the instruction kinds of the march loop's tests and selects, 8 waves per SIMD
        const int w = 8;
        run_gen("v_add_f32 (ref)", g_add_sgpr, w, d_out, d_cyc);
        run_gen("v_add_u32 lit", g_addu, w, d_out, d_cyc);
        run_gen("v_cmp_gt_u32", g_cmpu, w, d_out, d_cyc);
        run_gen("v_max3_f32", g_max3, w, d_out, d_cyc);
        run_gen("v_max3 neg", g_max3n, w, d_out, d_cyc);
        run_gen("v_min3_u32", g_min3u, w, d_out, d_cyc);
        run_gen("v_cndmask sgpr", g_cnds, w, d_out, d_cyc);
        run_gen("cmp+cnd pair/2", g_cmpcnd, w, d_out, d_cyc);
        run_gen("v_mul_legacy", g_mulleg, w, d_out, d_cyc);
        run_gen("v_max 0 lit", g_maxlit, w, d_out, d_cyc);
        run_gen("v_or_b32", g_or, w, d_out, d_cyc);
        run_gen("v_cvt_i32_f32", g_cvt, w, d_out, d_cyc);
        run_gen("v_cmp sgpr dst", g_cmps, w, d_out, d_cyc);
        run_gen("v_cmp_class", g_cmpx, w, d_out, d_cyc);
        run_gen("mul+max pair/2", g_muladd, w, d_out, d_cyc);
        run_gen("v_fmaak", g_fmak, w, d_out, d_cyc);
        run_gen("v_fmamk", g_madmk, w, d_out, d_cyc);
        run_gen("v_fma a,b,a", g_fma3, w, d_out, d_cyc);
        run_gen("v_fma sgpr", g_fmas, w, d_out, d_cyc);
        run_gen("v_max_f32", g_max, w, d_out, d_cyc);
        run_gen("v_rsq_f32", g_rsq, w, d_out, d_cyc);
        run_gen("v_floor_f32", g_floor, w, d_out, d_cyc);
        run_gen("v_fmac vop2", g_fmac, w, d_out, d_cyc);
        return 0;
    }
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_f32", w, d_out, d_cyc, 1);
        run<1>("v_pk_add_f32", w, d_out, d_cyc, 2);
        run<2>("v_fma_f32", w, d_out, d_cyc, 2);
        run<3>("v_sqrt_f32", w, d_out, d_cyc, 1);
        run<4>("add dependent", w, d_out, d_cyc, 1);
        run<5>("v_pk_mul sgpr", w, d_out, d_cyc, 2);
        run<6>("cmp+cndmask", w, d_out, d_cyc, 0);
    }
    for (int w : {1, 8}) {
        run_gen("v_mul literal", g_mul_lit, w, d_out, d_cyc);
        run_gen("v_add sgpr", g_add_sgpr, w, d_out, d_cyc);
        run_gen("v_fmac vop2", g_fmac, w, d_out, d_cyc);
        run_gen("v_fma inline", g_fma_lit, w, d_out, d_cyc);
        run_gen("v_max_f32", g_max, w, d_out, d_cyc);
        run_gen("v_med3_f32", g_med3, w, d_out, d_cyc);
        run_gen("v_cmp vcc", g_cmp32, w, d_out, d_cyc);
        run_gen("v_cndmask vcc", g_cnd32, w, d_out, d_cyc);
        run_gen("v_rsq_f32", g_rsq, w, d_out, d_cyc);
        run_gen("v_floor_f32", g_floor, w, d_out, d_cyc);
        run_gen("v_and literal", g_and, w, d_out, d_cyc);
        run_gen("v_add |x| sgpr", g_addabs, w, d_out, d_cyc);
        run_gen("v_sub_f32", g_sub, w, d_out, d_cyc);
        run_gen("v_mov_b32", g_mov, w, d_out, d_cyc);
    }
    return 0;
}
