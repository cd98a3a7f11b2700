// device_render.h - the frame kernel's body, generic over the scene evaluator
// (InterpScene: distance program + tree walker; JitScene: per-scene generated
// straight-line code).  Compiles under hipcc and under hiprtc.
#pragma once
#include "device_math.h"

namespace rm {

// =============================================================================
// One fused pass replaces the reference's kernel_SetRays + kernel_Process
// (Source/cuda_kernels.cu:598-762).
//   * a wavefront owns an 8x8 pixel tile, a 256-thread workgroup a 16x16 block;
//   * every lane is a small state machine (primary march -> waiting for a
//     normal -> shadow march -> done).  Primary and shadow marches need the
//     same thing, DIST(root, point), so lanes in either phase share one
//     evaluation per loop trip instead of serialising the reference's nested
//     loops (:626, :660) lane by lane;
//   * lanes that reached a surface wait until enough of the wave needs a
//     normal, then the normal is computed for exactly those lanes.
// =============================================================================
template <class Scene>
__device__ __forceinline__ void render_body(const RenderParams& prm) {
    Scene scene;
    typename Scene::Scratch scratch;
    scene.init(prm);

    // 16x16 pixels per workgroup, one 8x8 tile per wavefront
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = (int)(blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7));
    const uint32_t ly = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);       // row in the local buffer
    uint32_t gy = ly;                                                          // row in the image
    if (prm.band_rows) gy = ((ly / prm.band_rows) * prm.nranks + prm.rank) * prm.band_rows + ly % prm.band_rows;
    const bool in_image = px < (int)prm.width && gy < prm.height && ly < prm.local_rows;

    P3 pos, rd;
    make_ray(prm.cam, prm.width, prm.height, px, (int)gy, pos, rd);
    const P3 light{prm.cam.light[0], prm.cam.light[1], prm.cam.light[2]};

    float t = 0.f, cd = 0.f, n_l = 0.f, shadow = 1.f;
    int mode = in_image ? M_PRIMARY : M_DONE, steps = 0;
    bool need = false;                       // waiting for NORM(root, pos)
    uint32_t pixel = 0xff000000u, hit = 0;   // Source/cuda_kernels.cu:622

    // the two places a normal is consumed (Source/cuda_kernels.cu:640-658 and :673-691)
    auto advance_primary = [&]() {                                             // :707-714
        pos.x += cd * rd.x; pos.y += cd * rd.y; pos.z += cd * rd.z;
        t += cd;
        ++steps;
        if (t >= RM_MAX_DIST || steps >= RM_PRIMARY_STEPS) mode = M_DONE;
    };
    auto finish = [&]() {                                                      // :693-701
        float lum = 255.f * (RM_MIN_LUM + (RM_MAX_LUM - RM_MIN_LUM) * (n_l > 0.f ? n_l : 0.f) * shadow);
        float L = (float)to_u8(lum);
        pixel = to_u8(L * RM_MATERIAL_R) | (to_u8(L * RM_MATERIAL_G) << 8) | (to_u8(L * RM_MATERIAL_B) << 16) | 0xff000000u;
        hit = 1;
        mode = M_DONE;
    };
    auto shadow_tail = [&]() {                                                 // :679-690
        shadow = min2(shadow, RM_HARDNESS * cd / t);
        if (shadow < 0.01f) { finish(); return; }
        pos.x += cd * light.x; pos.y += cd * light.y; pos.z += cd * light.z;
        t += cd;
        ++steps;
        if (t >= RM_MAX_DIST || steps >= RM_SHADOW_STEPS) finish();
    };
    auto primary_normal = [&](P3 n) {                                          // :640-658
        if (dot3(n.x, n.y, n.z, rd.x, rd.y, rd.z) < 0.f) {
            n_l = dot3(n.x, n.y, n.z, light.x, light.y, light.z);
            shadow = 1.f;
            const float k = (1 + RM_DELTA) * RM_MIN_DIST;
            pos.x += k * light.x; pos.y += k * light.y; pos.z += k * light.z;
            t = k;
            mode = M_SHADOW;
            steps = 0;
        } else {
            advance_primary();
        }
    };
    auto shadow_normal = [&](P3 n) {                                           // :673-676
        if (dot3(n.x, n.y, n.z, rd.x, rd.y, rd.z) < RM_MIN_DIST) { shadow = 0.f; finish(); }
        else shadow_tail();
    };

    // optional per-wave diagnostics (rm_debug_wave_stats): trips, normal passes, cycles
    uint32_t st_evals = 0, st_norms = 0, st_norm_cycles = 0;
    const uint64_t st_t0 = prm.wave_stats ? __builtin_amdgcn_s_memtime() : 0;

    for (;;) {
        const bool active = (mode != M_DONE) && !need;
        const unsigned long long amask = __builtin_amdgcn_ballot_w64(active);
        if (amask != 0) {
            ++st_evals;
            // DIST(root, pos) for the whole wave (:627 and :661 share this)
            float d = scene.dist(pos, scratch);
            if (active) {
                cd = d;
                if (mode == M_PRIMARY) {
                    if (cd < RM_MIN_DIST) {                                    // :629
                        if (cd < -0.f) primary_normal(P3{-rd.x, -rd.y, -rd.z});  // :630-633
                        else need = true;
                    } else {
                        advance_primary();
                    }
                } else {
                    if (8 * cd < RM_MIN_DIST) {                                // :662
                        if (cd < -0.f) shadow_normal(P3{-light.x, -light.y, -light.z});   // :663-666
                        else need = true;
                    } else {
                        shadow_tail();
                    }
                }
            }
        }
        const unsigned long long nmask = __builtin_amdgcn_ballot_w64(need);
        const unsigned long long run = __builtin_amdgcn_ballot_w64((mode != M_DONE) && !need);
        if (nmask == 0) { if (run == 0) break; continue; }
        if (run != 0 && (uint32_t)__builtin_popcountll(nmask) < prm.norm_batch) continue;
        const uint64_t st_n0 = prm.wave_stats ? __builtin_amdgcn_s_memtime() : 0;
        ++st_norms;
        if (need) {                                                            // :635-637, :668-670
            P3 N = scene.norm(pos, scratch);
            float R_1 = r_length_3(N.x, N.y, N.z);
            P3 n{R_1 * N.x, R_1 * N.y, R_1 * N.z};
            if (mode == M_PRIMARY) primary_normal(n); else shadow_normal(n);
            need = false;
        }
        if (prm.wave_stats) st_norm_cycles += (uint32_t)(__builtin_amdgcn_s_memtime() - st_n0);
    }

    if (prm.wave_stats && lane == 0) {
        uint32_t* o = prm.wave_stats + 4 * (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave);
        o[0] = st_evals; o[1] = st_norms; o[2] = (uint32_t)(__builtin_amdgcn_s_memtime() - st_t0); o[3] = st_norm_cycles;
    }

    if (in_image) {                                                            // :717
        *reinterpret_cast<uint32_t*>(prm.rgba + (size_t)ly * prm.pitch + (size_t)px * 4) = pixel;
        if (prm.hit) prm.hit[(size_t)ly * prm.hit_pitch + (size_t)px] = (uint8_t)hit;
    }
}

}  // namespace rm
