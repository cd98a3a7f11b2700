// device_render.h - the frame kernel's body, generic over the scene evaluator
// (InterpScene: distance program + tree walker; JitScene: per-scene generated
// straight-line code).  Compiles under hipcc and under hiprtc.
#pragma once
#include "device_math.h"

namespace rm {

// =============================================================================
// One fused pass replaces the reference's kernel_SetRays + kernel_Process
// (Source/cuda_kernels.cu:598-762).
//
// Persistent wavefronts with lane refill.  The grid is sized to what the chip
// holds at once; every wavefront pulls 8x8 pixel tiles from a queue until the
// queue is dry:
//   * every lane is a small state machine (primary march -> waiting for a
//     normal -> shadow march -> done).  Primary and shadow marches need the
//     same thing, DIST(root, point), so lanes in either phase share one
//     evaluation per loop trip instead of serialising the reference's nested
//     loops (:626, :660) lane by lane;
//   * a lane whose pixel is finished stores it and takes the next pixel of the
//     wave's current tile, so a ray of 800 steps costs one lane, not a wave,
//     and no wavefront idles while tiles are left (ray lengths differ by
//     two orders of magnitude inside one frame);
//   * lanes that reached a surface wait until enough of the wave needs a
//     normal, then the normal is computed for exactly those lanes.
//
// Tile queue: RM_QUEUE_HEADS counters; head h hands out tiles h, h+8, h+16...
// A wave starts on the head of its workgroup (workgroups are dealt round-robin
// over the XCDs, so a head is mostly pulled from one XCD) and moves on to the
// next head when its own is dry; after all heads it stops pulling.
//
// Heavy tiles first.  A frame is as long as its longest ray chain: one ray of
// 800 steps is ~0.3 ms of dependent work however idle the chip is, so such rays
// must start when the frame starts.  A pixel that took prm.heavy_cost steps
// or more stamps its tile with the frame number and appends it to this frame's
// heavy list; the NEXT frame hands out that list before the natural order and
// skips stamped tiles there.  Only the order of the work depends on the previous
// frame, never a pixel: with a stale or empty list the frame is just slower.
//
// The queue runs two tiles ahead of the wave (counter value -> tile id ->
// stamp / list entry), so neither the atomic nor the dependent load is waited for.
// =============================================================================
template <class Scene>
__device__ __forceinline__ void render_body(const RenderParams& prm) {
    Scene scene;
    typename Scene::Scratch scratch;
    scene.init(prm);

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const P3 light{prm.cam.light[0], prm.cam.light[1], prm.cam.light[2]};

    // this frame found its scheduler set at zero; leave the next frame's at zero too
    if (blockIdx.x == 0 && threadIdx.x < 16) prm.set_next[threadIdx.x * RM_QUEUE_STRIDE] = 0;

    const uint32_t tiles_x = (prm.width + 7) >> 3, tiles_y = (prm.local_rows + 7) >> 3;
    const uint32_t ntiles = tiles_x * tiles_y;
    const uint32_t n_heavy = min(prm.set_prev[RM_SET_HEAVY_COUNT], prm.heavy_cap);

    // ---- tile queue (wave-uniform; values that come from memory sit in lane 0 until they are needed)
    auto uniform = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    // One lane pulls for the wave.  Two things about this build keep that honest:
    //  * the lane index is taken by an asm the compiler cannot see through, afresh for every
    //    pull.  With a plain `if (lane == 0)` it threads the branch through the cross-lane
    //    reads that follow and splits the wave into a lane-0 path and a path of the other
    //    lanes, each with a queue state of its own (and the other lanes' value must not be a
    //    constant, or the cross-lane read of it is folded away);
    //  * the atomic optimizer is switched off (Makefile, jit.cpp): it would wait for the
    //    counter value right behind the atomic, and the point is to have it in flight while
    //    the wave marches.
    auto pull = [&](uint32_t counter) -> uint32_t {
        uint32_t below;                              // lanes below this one: 0 in lane 0
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(below));
        uint32_t v = below;
        if (below == 0) v = __hip_atomic_fetch_add(prm.set_cur + counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" : "+v"(v));                  // the branch joins here, in a block of its own
        return v;
    };
    auto load0 = [&](const uint32_t* p) -> uint32_t { return __builtin_nontemporal_load(p); };
    // stage A: a counter value on its way
    uint32_t a_ticket = 0, a_head = blockIdx.x & (RM_QUEUE_HEADS - 1), a_tried = 0;
    bool a_heavy = n_heavy != 0, a_done = false;
    auto issue_a = [&]() { a_ticket = pull(a_heavy ? RM_SET_HEAVY_HEAD : a_head * RM_QUEUE_STRIDE); };
    // stage S: a tile whose stamp (natural order) or whose id (heavy list) is on its way
    uint32_t s_val = 0, s_tile = 0;
    bool s_heavy = false, s_valid = false;
    auto stage_s = [&]() {
        s_valid = false;
        while (!a_done) {
            const uint32_t c = uniform(a_ticket);
            if (a_heavy) {
                if (c < n_heavy) { s_heavy = true; s_val = load0(prm.heavy_prev + c); s_valid = true; issue_a(); return; }
                a_heavy = false;                    // list handed out: on to the natural order
                issue_a();
                continue;
            }
            const uint32_t tile = a_head + RM_QUEUE_HEADS * c;
            if (tile < ntiles) { s_heavy = false; s_tile = tile; s_val = load0(prm.stamp + tile); s_valid = true; issue_a(); return; }
            if (++a_tried == RM_QUEUE_HEADS) { a_done = true; return; }
            a_head = (a_head + 1) & (RM_QUEUE_HEADS - 1);
            issue_a();
        }
    };
    bool q_dry = false;
    uint32_t tile_px0 = 0, tile_ly0 = 0, taken = RM_TILE_PIXELS;      // nothing in hand yet
    auto next_tile = [&]() {
        for (;;) {
            if (!s_valid) { q_dry = true; return; }
            const uint32_t v = uniform(s_val);
            const bool heavy = s_heavy;
            const uint32_t tile = heavy ? v : s_tile;
            stage_s();
            // natural order: skip what the heavy list has handed out (stamped by the previous
            // frame) or is stamped already by this one (then it came from the list, too)
            if (heavy ? tile >= ntiles : (v == prm.seq || v == prm.seq - 1)) continue;
            const uint32_t ty = tile / tiles_x;
            tile_px0 = (tile - ty * tiles_x) * 8;
            tile_ly0 = ty * 8;
            taken = 0;
            return;
        }
    };
    issue_a();
    stage_s();

    // ---- per-lane ray state
    P3 pos{0.f, 0.f, 0.f}, rd{0.f, 0.f, 0.f};
    float t = 0.f, cd = 0.f, n_l = 0.f, shadow = 1.f;
    int mode = M_DONE, steps = 0, psteps = 0;    // psteps: primary steps once the shadow march has begun
    bool need = false;                       // waiting for NORM(root, pos)
    bool live = false;                       // holds a finished or running pixel that is not stored yet
    uint32_t px = 0, ly = 0;                 // that pixel: column, row in the local buffer
    uint32_t born = 0;                       // loop trip at which the lane took its pixel
    bool boosted = false;
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
            psteps = steps;
            steps = 0;
        } else {
            advance_primary();
        }
    };
    auto shadow_normal = [&](P3 n) {                                           // :673-676
        if (dot3(n.x, n.y, n.z, rd.x, rd.y, rd.z) < RM_MIN_DIST) { shadow = 0.f; finish(); }
        else shadow_tail();
    };
    auto store_pixel = [&]() {                                                 // :717
        *reinterpret_cast<uint32_t*>(prm.rgba + (size_t)ly * prm.pitch + (size_t)px * 4) = pixel;
        if (prm.hit) prm.hit[(size_t)ly * prm.hit_pitch + (size_t)px] = (uint8_t)hit;
        if ((uint32_t)(steps + psteps) >= prm.heavy_cost) {                    // a long ray: its tile goes first next frame
            const uint32_t tile = (ly >> 3) * tiles_x + (px >> 3);
            if (__hip_atomic_exchange(prm.stamp + tile, prm.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != prm.seq) {
                const uint32_t at = __hip_atomic_fetch_add(prm.set_cur + RM_SET_HEAVY_COUNT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (at < prm.heavy_cap) prm.heavy_cur[at] = tile;
            }
        }
    };

    // optional per-wave diagnostics (rm_debug_wave_stats): trips, normal passes, cycles
    uint32_t st_evals = 0, st_norms = 0, st_norm_cycles = 0;
    const uint64_t st_t0 = prm.wave_stats ? __builtin_amdgcn_s_memtime() : 0;

    // Safety net for the persistent loop: every wave leaves after 2^36 shader cycles (about half
    // a minute) whatever state it is in, so a defect in the queue logic cannot hang the device.
    const uint64_t t_enter = __builtin_amdgcn_s_memtime();
    uint32_t trip = 0;

    for (;;) {
        if ((++trip & 0x3ff) == 0 && __builtin_amdgcn_s_memtime() - t_enter > (1ull << 36)) break;

        // ---- refill: finished lanes store their pixel and take the next ones of the tile in hand
        const bool idle = mode == M_DONE;
        const unsigned long long imask = __builtin_amdgcn_ballot_w64(idle);
        const uint32_t nidle = (uint32_t)__builtin_popcountll(imask);
        if (!q_dry && nidle >= prm.refill_lanes) {
            if (taken == RM_TILE_PIXELS) next_tile();
            if (!q_dry) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(imask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)imask, 0u));
                const uint32_t i = taken + rank;                               // pixel of the tile, row-major
                if (idle && i < RM_TILE_PIXELS) {
                    if (live) store_pixel();
                    px = tile_px0 + (i & 7);
                    ly = tile_ly0 + (i >> 3);
                    uint32_t gy = ly;                                          // row in the image
                    if (prm.band_rows) gy = ((ly / prm.band_rows) * prm.nranks + prm.rank) * prm.band_rows + ly % prm.band_rows;
                    live = px < prm.width && gy < prm.height && ly < prm.local_rows;
                    make_ray(prm.cam, prm.width, prm.height, (int)px, (int)gy, pos, rd);
                    t = 0.f; steps = 0; psteps = 0;
                    born = st_evals;
                    pixel = 0xff000000u; hit = 0;
                    mode = live ? M_PRIMARY : M_DONE;
                }
                taken = (taken + nidle < RM_TILE_PIXELS) ? taken + nidle : RM_TILE_PIXELS;
            }
        }

        // a wave that carries an old ray goes first on its SIMD: long rays set the
        // length of the frame, they must not queue behind six other waves per step
        {
            const bool want = __builtin_amdgcn_ballot_w64(mode != M_DONE && st_evals - born > prm.boost_age) != 0;
            if (want != boosted) {
                if (want) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
                boosted = want;
            }
        }

        // ---- one march step for every lane that has a ray in flight
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

        // ---- normals, for the lanes that wait for one
        const unsigned long long nmask = __builtin_amdgcn_ballot_w64(need);
        const unsigned long long run = __builtin_amdgcn_ballot_w64((mode != M_DONE) && !need);
        if (nmask == 0) {
            if (run == 0 && q_dry) break;        // nothing in flight, nothing left to pull
            continue;
        }
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

    if (live) store_pixel();

    if (prm.wave_stats && lane == 0) {
        uint32_t* o = prm.wave_stats + 4 * ((size_t)blockIdx.x * 4 + wave);
        o[0] = st_evals; o[1] = st_norms; o[2] = (uint32_t)(__builtin_amdgcn_s_memtime() - st_t0); o[3] = st_norm_cycles;
    }
}

}  // namespace rm
