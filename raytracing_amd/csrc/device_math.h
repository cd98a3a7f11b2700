// device_math.h - point type, scalar conventions, primitives, transforms, leaf
// normals, ray generation: everything both the interpreter build and the
// per-scene generated code share.  Compiles under hipcc and under hiprtc.
#pragma once
#include "../../include/rm_types.h"
#include "kernel_params.h"

namespace rm {

struct P3 { float x, y, z; };

// ---- scalar helpers (same arithmetic conventions as the CPU restatement) ---------------------

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

// This lane's bit of a wave-level mask held in scalar registers (the inverse of a ballot): the mask is
// used as the branch's or select's condition as it is, no vector instruction.  The compiler's builtin for
// it is missing from the hiprtc of ROCm 7.0 that a process with PyTorch loads, so the intrinsic is
// declared by name.
extern "C" __device__ bool rm_lane_of_mask(unsigned long long) __asm("llvm.amdgcn.inverse.ballot.i64");

// ---- correctly rounded sqrt and constant-divisor division in fewer instructions ---------------
//
// hipcc expands IEEE sqrtf into ~16 VALU instructions and '/' into ~12; a march step of
// chair.sce holds three sqrt and one division, 40 % of its instructions.  The sequences
// below give the SAME bits for every input, checked on the device rather than argued:
//
//  sqrt_exact: one Newton step on v_rsq_f32 with the residual taken by fma.  Compared with
//    sqrtf for every float: identical for 2^-102 <= x <= FLT_MAX (tools/microbench/
//    exact_math.hip; rm_selftest_math repeats the sweep through the C ABI, and rm_create
//    refuses a device on which two binades of it fail).  Below 2^-100 (0, denormals: the
//    residual underflows), and for inf, NaN and negatives, the wave falls back to sqrtf.
//
//  div_const(a, c, rc): Markstein's quotient from rc = RN(1/c): q = a*rc, q += (a - c*q)*rc,
//    both corrections by fma.  Every operation scales exactly with a power of two while
//    nothing leaves the normal range, so one binade of numerators proves a divisor for all
//    2^-80 <= |a| < 2^80: rm_set_primitives runs that proof on the device for every divisor
//    constant of a scene (senfina_ripeto's dot(a,a), glata's k) and hands rc = 0 for one
//    that fails or lies outside [2^-20, 2^20]; rc = 0, or a numerator outside the range,
//    takes IEEE division.
__device__ __forceinline__ float sqrt_exact(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float g = x * r, h = 0.5f * r;
    float y = __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(as_u(x) - 0x0d800000u >= 0x72000000u) != 0, 0)) {   // rare, whole wave
        float xs = x;
        asm volatile("; sqrt_exact: IEEE path" : "+v"(xs));      // keeps this a branch: the compiler would compute both and select
        y = sqrtf(xs);
    }
    return y;
}
__device__ __forceinline__ float div_const(float a, float c, float rc) {
    if (rc == 0.f) return a / c;                     // wave-uniform: rc is a scene constant
    float q = a * rc;
    q = __builtin_fmaf(__builtin_fmaf(-c, q, a), rc, q);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64((as_u(a) & 0x7fffffffu) - 0x17800000u >= 0x50000000u) != 0, 0)) {
        float as = a;
        asm volatile("; div_const: IEEE path" : "+v"(as));       // as in sqrt_exact
        q = as / c;
    }
    return q;
}
// The quotient of senfina_ripeto (:577) is only ever used as floor(q + .5f).  Below the range div_const proves
// (|a| < 2^-80, with 2^-20 <= |c|) the IEEE quotient is under 2^-60 and the short sequence's under 2^-50 (no
// operation of it can produce an infinity or a NaN from a tiny or zero numerator): either way q + .5f rounds
// to .5f and the floor is +0.  So the guard is one-sided here, one compare with a modifier instead of three
// instructions - and a lattice takes one to three of these quotients per march step.
__device__ __forceinline__ float div_const_for_floor(float a, float c, float rc) {
    if (rc == 0.f) return a / c;                     // wave-uniform: rc is a scene constant
    float q = a * rc;
    q = __builtin_fmaf(__builtin_fmaf(-c, q, a), rc, q);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(fabsf(a) < 0x1p80f)) != 0, 0)) {      // also NaN
        float as = a;
        asm volatile("; div_const_for_floor: IEEE path" : "+v"(as));
        q = as / c;
    }
    return q;
}
__device__ __forceinline__ float min2(float a, float b) { return (b < a) ? b : a; }   // std::min
__device__ __forceinline__ float max2(float a, float b) { return (a < b) ? b : a; }   // std::max
// Include/cuda_defines.cuh:49-53
__device__ __forceinline__ float length_3(float x, float y, float z) { return sqrt_exact(x * x + y * y + z * z); }
__device__ __forceinline__ float r_length_3(float x, float y, float z) { return 1.f / sqrt_exact(x * x + y * y + z * z); }
__device__ __forceinline__ float length_2(float x, float y) {
    return (float)sqrt((double)x * (double)x + (double)y * (double)y);
}
// Source/cuda_kernels.cu:38-44
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return ax * bx + ay * by + az * bz;
}
__device__ __forceinline__ float mixf(float a, float b, float x) { return a + x * (b - a); }

// ---- primitives (Source/cuda_kernels.cu:72-136), q = query point -------------

__device__ __forceinline__ float sfero_dist(P3 q, float r) { return length_3(q.x, q.y, q.z) - r; }

__device__ __forceinline__ float kubo_dist(P3 p, float bx, float by, float bz) {
    float qx = fabsf(p.x) - bx, qy = fabsf(p.y) - by, qz = fabsf(p.z) - bz;
    // The reference's max() is a compare and a select; v_max_f32 / v_max3_f32 differ from that only
    // in the sign of a zero result (and for NaN, which finite positions never produce).  The clamps
    // are squared, which drops the sign.  `inside` is used only where all three are negative: no
    // zero among them, and "all three negative" is "their maximum is negative" (-0 is not).
    // The range test of the short square root (sqrt_exact) is the inside test as well: a sum of the squared
    // clamps of at least 2^-102 has a positive clamp in it, so the point is not inside and the value is the
    // exterior one (:92); a wave with a lane inside the box (sum 0), or with a sum that small, infinite or NaN,
    // takes the branch, which is the reference's selection as it stands (:86-92).  Off the box that is the
    // maximum, a compare and a select less per evaluation (each costs 1.7 vector additions on this chip).
    const float cx = __builtin_fmaxf(qx, 0.f), cy = __builtin_fmaxf(qy, 0.f), cz = __builtin_fmaxf(qz, 0.f);
    const float s = cx * cx + cy * cy + cz * cz;
    const float r = __builtin_amdgcn_rsqf(s);
    const float g = s * r, h = 0.5f * r;
    float y = __builtin_fmaf(__builtin_fmaf(-g, g, s), h, g);
    const bool odd = as_u(s) - 0x0d800000u >= 0x72000000u;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(odd) != 0, 0)) {               // rare off the box, whole wave
        float ss = s;
        asm volatile("; kubo_dist: inside the box, or IEEE path" : "+v"(ss));       // keeps this a branch
        const float inside = __builtin_fmaxf(__builtin_fmaxf(qx, qy), qz);
        const bool in = inside < 0.f;
        if (__builtin_amdgcn_ballot_w64(odd && !in) != 0) {                        // on the surface (sqrt 0), or out of the short range
            const float e = sqrtf(ss);
            y = odd ? e : y;
        }
        y = in ? inside : y;
    }
    return y;
}

__device__ __forceinline__ float cilindro_dist(P3 p, float r, float h) {
    float qx = length_2(p.x, p.y) - r, qy = fabsf(p.z) - h;
    float inside = qx > qy ? qx : qy;
    float outside = length_2(max2(qx, 0.f), max2(qy, 0.f));
    return (qx < 0.f && qy < 0.f) ? inside : outside;
}

__device__ __forceinline__ float ebeno_dist(P3 p, float nx, float ny, float nz) {
    return dot3(nx, ny, nz, p.x, p.y, p.z);
}

// smooth union / intersection (Source/cuda_kernels.cu:356-359, :384-387)
__device__ __forceinline__ float glata(float d0, float d1, float k, float rk, bool komunajo) {
    float q = div_const(d0 - d1, k, rk);
    float h = (komunajo ? (1.f + q) : (1.f - q)) * .5f;
    float m = mixf(d0, d1, h);
    float e = k * h * (1.f - h);
    float blended = komunajo ? (m + e) : (m - e);
    return h > 1.f ? d0 : (h < 0.f ? d1 : blended);
}

// OP_CULL (scene_program.h): true if no marching lane of the wave (amask: their ballot) can be
// nearer to the group inside the sphere (centre c, k = radius + margin) than its running minimum A
__device__ __forceinline__ bool cull_all(P3 S, float A, unsigned long long amask, float cx, float cy, float cz, float k) {
    const float dx = S.x - cx, dy = S.y - cy, dz = S.z - cz;
    const float s2 = dx * dx + dy * dy + dz * dz, T = A + k;
    // a lane is near unless T < 0 or s2 > T^2 (1 + 2^-9) (near for NaN: never culls).  Two compares
    // straight into scalar masks: a ballot of a combined predicate costs two more vector instructions.
    const unsigned long long near = __builtin_amdgcn_ballot_w64(!(T < 0.f)) & __builtin_amdgcn_ballot_w64(!(s2 > T * T * 1.001953125f));
    return (near & amask) == 0;
}
// ---- transforms (Source/cuda_kernels.cu:409-413,430-432,454-456,478-480,527-530,577-581)

__device__ __forceinline__ P3 movo(P3 p, float tx, float ty, float tz) { return P3{p.x - tx, p.y - ty, p.z - tz}; }
__device__ __forceinline__ P3 rotx(P3 p, float c, float s) { return P3{p.x, c * p.y + s * p.z, -s * p.y + c * p.z}; }
__device__ __forceinline__ P3 roty(P3 p, float c, float s) { return P3{-s * p.z + c * p.x, p.y, c * p.z + s * p.x}; }
__device__ __forceinline__ P3 rotz(P3 p, float c, float s) { return P3{c * p.x + s * p.y, -s * p.x + c * p.y, p.z}; }
// p + 2 * s (:527-530, :567-569) as one fma: 2 * s is exact in binary floating point (inf stays inf, denormals
// double exactly), so fma(2, s, p) rounds the same sum once, as the reference's multiply and add do
__device__ __forceinline__ float plus_twice(float p, float s) { return __builtin_fmaf(2.f, s, p); }
__device__ __forceinline__ P3 rotq(P3 p, float xx, float xy, float xz, float yx, float yy, float yz,
                                   float zx, float zy, float zz) {
    return P3{plus_twice(p.x, xx * p.x + xy * p.y + xz * p.z),
              plus_twice(p.y, yx * p.x + yy * p.y + yz * p.z),
              plus_twice(p.z, zx * p.x + zy * p.y + zz * p.z)};
}
// ripeto_t: generated code names the components of `a` that are zero.  Such a one adds +-0 to the dot product and takes
// N * 0 = +-0 off its coordinate (positions and N are finite): without its terms a result can
// differ only in the sign of a zero, which nothing downstream tells apart (no division by, square
// root of or copysign from such a value; comparisons and sums of squares do not see it).
template <bool ZX, bool ZY, bool ZZ>
__device__ __forceinline__ float ripeto_dot(P3 p, float ax, float ay, float az) {
    if (ZX && ZY) return az * p.z;
    if (ZX && ZZ) return ay * p.y;
    if (ZY && ZZ) return ax * p.x;
    if (ZX) return ay * p.y + az * p.z;
    if (ZY) return ax * p.x + az * p.z;
    if (ZZ) return ax * p.x + ay * p.y;
    return dot3(ax, ay, az, p.x, p.y, p.z);
}
template <bool ZX, bool ZY, bool ZZ>
__device__ __forceinline__ P3 ripeto_t(P3 p, float ax, float ay, float az, float aa, float raa) {
    const float dot = ripeto_dot<ZX, ZY, ZZ>(p, ax, ay, az);
    float N = floorf(div_const_for_floor(dot, aa, raa) + .5f);
    return P3{ZX ? p.x : p.x - N * ax, ZY ? p.y : p.y - N * ay, ZZ ? p.z : p.z - N * az};
}
__device__ __forceinline__ P3 ripeto(P3 p, float ax, float ay, float az, float aa, float raa) {
    return ripeto_t<false, false, false>(p, ax, ay, az, aa, raa);
}

// ---- leaf normals (Source/cuda_kernels.cu:94-104, :120-126), unnormalised ----

__device__ __forceinline__ P3 kubo_norm(P3 p, float bx, float by, float bz) {
    float qx = fabsf(p.x) - bx, qy = fabsf(p.y) - by, qz = fabsf(p.z) - bz;
    float sx = p.x > 0.f ? 1.f : -1.f, sy = p.y > 0.f ? 1.f : -1.f, sz = p.z > 0.f ? 1.f : -1.f;
    if (qx < 0.f && qy < 0.f && qz < 0.f) {
        if (qx > qz) return (qx > qy) ? P3{sx, 0.f, 0.f} : P3{0.f, sy, 0.f};
        return (qy > qz) ? P3{0.f, sy, 0.f} : P3{0.f, 0.f, sz};
    }
    return P3{qx > 0.f ? sx : 0.f, qy > 0.f ? sy : 0.f, qz > 0.f ? sz : 0.f};
}

__device__ __forceinline__ P3 cilindro_norm(P3 p, float r, float h) {
    float qx = length_2(p.x, p.y) - r, qy = fabsf(p.z) - h;
    return (qx > qy) ? P3{p.x, p.y, 0.f} : P3{0.f, 0.f, p.z > 0.f ? 1.f : -1.f};
}

// ---- normals carried back through rotations (:443-445, :467-469, :491-493, :566-569)

__device__ __forceinline__ P3 rotx_back(P3 n, float c, float s) { return P3{n.x, c * n.y - s * n.z, s * n.y + c * n.z}; }
__device__ __forceinline__ P3 roty_back(P3 n, float c, float s) { return P3{s * n.z + c * n.x, n.y, c * n.z - s * n.x}; }
__device__ __forceinline__ P3 rotz_back(P3 n, float c, float s) { return P3{c * n.x - s * n.y, s * n.x + c * n.y, n.z}; }
__device__ __forceinline__ P3 rotq_back(P3 n, float xx, float xy, float xz, float yx, float yy, float yz,
                                        float zx, float zy, float zz) {
    return P3{plus_twice(n.x, xx * n.x + yx * n.y + zx * n.z),
              plus_twice(n.y, xy * n.x + yy * n.y + zy * n.z),
              plus_twice(n.z, xz * n.x + yz * n.y + zz * n.z)};
}
__device__ __forceinline__ P3 neg3(P3 n) { return P3{-n.x, -n.y, -n.z}; }

// blend of the two normalised child normals of a glata_* node (:369-374, :397-402)
__device__ __forceinline__ P3 glata_blend(P3 n0, P3 n1, float h) {
    float d0 = r_length_3(n0.x, n0.y, n0.z), d1 = r_length_3(n1.x, n1.y, n1.z);
    return P3{mixf(d0 * n0.x, d1 * n1.x, h), mixf(d0 * n0.y, d1 * n1.y, h), mixf(d0 * n0.z, d1 * n1.z, h)};
}
// the blend parameter h of a glata_* node (:356, :384)
__device__ __forceinline__ float glata_h(float d0, float d1, float k, float rk, bool komunajo) {
    float q = div_const(d0 - d1, k, rk);
    return (komunajo ? (1.f + q) : (1.f - q)) * .5f;
}

// kernel_SetRays, Source/cuda_kernels.cu:740-760, in two steps: the origin p and the direction before its
// normalisation (s = pos + Depth * Dir, :744-752), then d = rnorm3df(s) * s (:753-755).  The render kernel takes the
// second step only for tiles in which some ray can come near the scene (device_render.h, ray_window works on s).
__device__ __forceinline__ void make_ray_raw(const Camera& cam, uint32_t width, uint32_t height, int x, int y,
                                             P3& p, P3& s) {
    float X = .5f * (float)(2 * x - (int)width + 1);
    float Y = .5f * (float)(2 * y - (int)height + 1);
    float Z = cam.depth;
    P3 pos{X * cam.wvec[0] + Y * cam.hvec[0], X * cam.wvec[1] + Y * cam.hvec[1], X * cam.wvec[2] + Y * cam.hvec[2]};
    P3 dp{Z * cam.dir[0], Z * cam.dir[1], Z * cam.dir[2]};
    s = P3{pos.x + dp.x, pos.y + dp.y, pos.z + dp.z};
    p = P3{pos.x + cam.pos[0], pos.y + cam.pos[1], pos.z + cam.pos[2]};
}
__device__ __forceinline__ P3 ray_direction(P3 s) {
    float R_1 = r_length_3(s.x, s.y, s.z);
    return P3{R_1 * s.x, R_1 * s.y, R_1 * s.z};
}
__device__ __forceinline__ void make_ray(const Camera& cam, uint32_t width, uint32_t height, int x, int y,
                                         P3& p, P3& d) {
    P3 s;
    make_ray_raw(cam, width, height, x, y, p, s);
    d = ray_direction(s);
}

// float -> unsigned char of Source/cuda_kernels.cu:694-699, out-of-range values saturate
__device__ __forceinline__ uint32_t to_u8(float v) {
    return !(v > -1.f) ? 0u : (v >= 256.f ? 255u : (uint32_t)(int)v);
}

}  // namespace rm
