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
// Persistent wavefronts.  The grid is sized to what the chip holds at once;
// every wavefront pulls 8x8 pixel tiles (one pixel per lane) from a queue until
// the queue is dry.  Inside a tile every lane is a small state machine (primary
// march -> waiting for a normal -> shadow march -> done).  Primary and shadow
// marches need the same thing, DIST(root, point), so lanes in either phase share
// one evaluation per loop trip instead of serialising the reference's nested
// loops (:626, :660) lane by lane; lanes that reached a surface get their normal
// in a pass of their own.
//
// Longest tiles first.  Ray lengths differ by two orders of magnitude inside a
// frame, and a ray of 800 steps is ~0.3 ms of dependent work however idle the
// chip is: a frame is as long as its longest tile plus the time that tile waited
// to start.  So every wave files each tile it finishes under its cost (loop
// trips) into one of RM_COST_BUCKETS lists, and the NEXT frame hands the lists
// out from the costliest bucket down (longest-processing-time-first).  Only the
// order of the work depends on the previous frame, never a pixel: without
// history (first frame, new tile numbering) tiles go out in natural order.  A
// wave whose tile has run long also raises its issue priority, so the tiles
// that bound the frame do not queue behind six other waves on their SIMD.
//
// Queue layout: RM_QUEUE_HEADS heads, each with its own pull counter and its own
// bucket lists; a wave works on the head of its workgroup (workgroups are dealt
// round-robin over the XCDs, so a head's counters stay in one XCD's L2 for the
// most part), files its tiles there, and moves on to the next head when its own
// is dry.  The queue runs two tiles ahead of the wave (counter value -> list
// entry -> tile), so neither the atomic nor the dependent load is waited for.
// =============================================================================
// The launch parameters, read from the kernarg segment through a pointer the compiler cannot see
// through: every call site loads what it needs there and then (scalar loads, once per tile), so
// the two dozen pointers, pitches and sizes used between tiles do not sit in SGPRs during the
// march - the scalar registers, not the vector ones, decide how many wavefronts a SIMD holds.
// Only for kernels whose first argument is the RenderParams (the two render kernels).
typedef const __attribute__((address_space(4))) RenderParams KernargParams;
__device__ __forceinline__ KernargParams* kernarg_params() {
    KernargParams* p = (KernargParams*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// Ray windows (include/raymarching_amd.h; no counterpart in the reference, which marches every ray until it
// has travelled RM_MAX_DIST, :707-714).  The host derives from the node tree a sphere, or a cylinder around a
// lattice's axis, of radius r with DIST(root, x) >= dist(x, centre or axis) - r everywhere (scene_bound.cpp).
// A primary ray can find a surface only while it is inside that volume grown by RM_WINDOW_MARGIN: outside of it
// DIST is more than the margin less what separates the march from the ideal ray (rm_types.h), hundreds of times
// RM_MIN_DIST.  So a ray that never enters the grown volume is the black pixel of a miss (:622) without a step,
// and one that does is marched no farther than where it leaves it for good: `t_end` takes RM_MAX_DIST's place in
// the test of :712.  The steps that ARE made are the reference's, from the ray's first on (the march is chaotic:
// where a ray lands depends on every step before), so pixels and hit masks do not change by a bit.
// The intersection is worked out in fp32 with its own rounding priced in: `mag` bounds every magnitude that
// enters, and the nearest approach |w| of the ray to the centre / axis is off by less than e_w, its place along
// the ray by less than e_s0 (5e-6 mag is 80 half-ulps of mag: the sums below are a dozen operations long, and
// for a cylinder the division by sin^2 of the angle between ray and axis stretches them by 1/sin^2 and 1/sin).
// The volume is grown by e_w on top of the margin, and the other error goes into the slack.
// It takes the ray's direction `v` BEFORE its normalisation (make_ray_raw): the test is homogeneous in it, and a
// tile none of whose rays can come near the volume never pays for the exactly rounded 1/|v| (a quarter of the ray
// set-up).  Distances along the ray are in units of |v| until the last line.
// Returns false for a ray that can be dropped.  t_end <= RM_MAX_DIST.
__device__ __forceinline__ bool ray_window(KernargParams* kp, P3 o, P3 v, float& t_end) {
    t_end = RM_MAX_DIST;
    const uint32_t kind = kp->bound_kind;
    if (kind == RM_BOUND_NONE) return true;                        // wave-uniform
    const float R = kp->bound_R;
    const float ocx = o.x - kp->bound_c[0], ocy = o.y - kp->bound_c[1], ocz = o.z - kp->bound_c[2];
    // ray origins far out get no window (false for NaN as well)
    bool usable = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) <= RM_WINDOW_MAX_COORD;
    float px = ocx, py = ocy, pz = ocz, qx = v.x, qy = v.y, qz = v.z;       // origin and direction, across the axis for a cylinder
    const float vv = v.x * v.x + v.y * v.y + v.z * v.z;
    float a = vv;                                                  // |q|^2 = |v|^2 sin^2 of the angle between ray and axis
    if (kind == RM_BOUND_CYLINDER) {
        const float ux = kp->bound_u[0], uy = kp->bound_u[1], uz = kp->bound_u[2];
        const float ou = ocx * ux + ocy * uy + ocz * uz, vu = v.x * ux + v.y * uy + v.z * uz;
        px -= ou * ux; py -= ou * uy; pz -= ou * uz;
        qx -= vu * ux; qy -= vu * uy; qz -= vu * uz;
        a = qx * qx + qy * qy + qz * qz;
        usable = usable && a >= RM_WINDOW_MIN_SIN2 * vv;
    }
    const float rs = __builtin_amdgcn_rsqf(a), inv_a = rs * rs;    // (v_rsq_f32 and v_sqrt_f32 are good to an ulp: inside the allowances below)
    const float len = __builtin_amdgcn_sqrtf(vv), si = len * rs;   // |v|; 1 / sin (1 for a sphere)
    const float s0 = -(px * qx + py * qy + pz * qz) * inv_a;       // where the ray is nearest to the centre / axis
    const float wx = px + s0 * qx, wy = py + s0 * qy, wz = pz + s0 * qz;
    const float p2 = wx * wx + wy * wy + wz * wz;                  // and how near, squared
    const float mag = fabsf(ocx) + fabsf(ocy) + fabsf(ocz) + R;
    const float Re = R + 5e-6f * mag * si;
    const float gap = Re * Re - p2;                                // >= 0: the ray passes through the grown volume
    // a tile none of whose rays comes near the volume (half of chair.sce's at 4K) is done here: where its rays would
    // leave the volume is not worked out (wave-uniform branch; the asm keeps the rest behind it)
    if (__builtin_amdgcn_ballot_w64(!(usable && gap < 0.f)) == 0) return false;
    asm volatile("" ::: "memory");
    const float s_leave = s0 + __builtin_amdgcn_sqrtf(fmaxf(gap, 0.f) * inv_a) * 1.00001f;
    // in units of travelled distance: the march moves by cd * (v / |v|), rounded, and adds cd up
    const float leave = s_leave * len + (RM_WINDOW_SLACK + 5e-6f * mag * (si * si) + 2e-6f * fabsf(s0 * len));
    if (usable && leave < RM_MAX_DIST) t_end = leave;              // (false for NaN)
    return !(usable && (gap < 0.f || leave < 0.f));
}

// The same volume ends a SHADOW march early (round 3; :660-691 marches until ray_dist >= RAYS_MAX_DIST, 300 steps or
// darkness).  A shadow ray starts on the surface, inside the volume, at o with ray_dist = t0, and moves by cd * l while
// ray_dist grows by cd (l: LightSource as given, any length): ray_dist = t0 + s at o + s l.  Let it leave the grown
// volume for good at s_leave.  From there on no sample can be near a surface (DIST > margin - drift > 4.9 > RAYS_MIN_DIST
// / 8: :662 never fires), and :679's quotient 128 * DIST / ray_dist stays at 1 or above, so min() keeps SHADOW (<= 1) bit
// for bit, provided that at the exit (a) ray_dist / 128 <= 4.5 - DIST is at least (distance to the axis) - r - drift = 4.9
// there - and (b) the distance to the axis grows by at least 1.02 / 128 per unit of ray_dist - it is convex in s, so it
// keeps growing at least that fast, and DIST with it.  Then nothing the rest of the march does can change SHADOW or end
// it otherwise than by :689 or the step cap: the lane is shaded when ray_dist reaches t0 + s_leave (plus the window's
// slack) exactly as it would be thousands of units further on.  s_end = RM_MAX_DIST where no such point is proved.
__device__ __forceinline__ float shadow_window(KernargParams* kp, P3 o, P3 l, float t0) {
    const uint32_t kind = kp->bound_kind;
    if (kind == RM_BOUND_NONE) return RM_MAX_DIST;                 // wave-uniform
    const float R = kp->bound_R;
    const float ocx = o.x - kp->bound_c[0], ocy = o.y - kp->bound_c[1], ocz = o.z - kp->bound_c[2];
    float px = ocx, py = ocy, pz = ocz, qx = l.x, qy = l.y, qz = l.z;
    const float vv = l.x * l.x + l.y * l.y + l.z * l.z;
    bool usable = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) <= RM_WINDOW_MAX_COORD && vv >= 0x1p-20f && vv <= 0x1p20f;
    float a = vv;
    if (kind == RM_BOUND_CYLINDER) {
        const float ux = kp->bound_u[0], uy = kp->bound_u[1], uz = kp->bound_u[2];
        const float ou = ocx * ux + ocy * uy + ocz * uz, vu = l.x * ux + l.y * uy + l.z * uz;
        px -= ou * ux; py -= ou * uy; pz -= ou * uz;
        qx -= vu * ux; qy -= vu * uy; qz -= vu * uz;
        a = qx * qx + qy * qy + qz * qz;
        usable = usable && a >= RM_WINDOW_MIN_SIN2 * vv;
    }
    const float rs = __builtin_amdgcn_rsqf(a), inv_a = rs * rs;
    const float len = __builtin_amdgcn_sqrtf(vv), si = len * rs, rlen = __builtin_amdgcn_rcpf(len);
    const float s0 = -(px * qx + py * qy + pz * qz) * inv_a;
    const float wx = px + s0 * qx, wy = py + s0 * qy, wz = pz + s0 * qz;
    const float p2 = wx * wx + wy * wy + wz * wz;
    const float mag = fabsf(ocx) + fabsf(ocy) + fabsf(ocz) + R;
    const float Re = R + 5e-6f * mag * si;
    const float gap = Re * Re - p2;
    const float s_leave = s0 + __builtin_amdgcn_sqrtf(fmaxf(gap, 0.f) * inv_a) * 1.00001f;
    // in units of ray_dist (= of s): the window's spatial allowances over |l|
    const float t_end = t0 + s_leave + (RM_WINDOW_SLACK + 5e-6f * mag * (si * si) + 2e-6f * fabsf(s0 * len)) * rlen * 1.00001f;
    const bool proved = usable && gap > 0.f && s_leave > 0.f
                        && t_end * (1.f / RM_HARDNESS) <= 4.5f                                    // (a)
                        && a * gap >= Re * Re * (1.02f * 1.02f / (RM_HARDNESS * RM_HARDNESS));    // (b): d|w|/ds = sqrt(a gap) / Re at the exit
    return proved ? t_end : RM_MAX_DIST;                           // (false for NaN)
}

#ifndef RM_NARROW_EXEC
#define RM_NARROW_EXEC 1
#endif

template <class Scene>
__device__ __forceinline__ void render_body(const RenderParams& prm) {
    Scene scene;
    typename Scene::Scratch scratch;
    scene.init(prm);

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // this frame found its scheduler set at zero; leave the next frame's at zero too
    if (blockIdx.x == 0 && threadIdx.x < RM_SET_LINES) prm.set_next[threadIdx.x * RM_QUEUE_STRIDE] = 0;
    if (blockIdx.x == 0 && threadIdx.x < RM_STATUS_COUNTERS) {      // the trips and longest-tile slots of the launch after this one
        prm.status[threadIdx.x * RM_STATUS_STRIDE + RM_STATUS_TRIPS + (prm.stat_slot ^ 1u)] = 0;
        prm.status[threadIdx.x * RM_STATUS_STRIDE + RM_STATUS_LONGEST + (prm.stat_slot ^ 1u)] = 0;
    }

    const uint32_t tiles_x = (prm.width + 7) >> 3, tiles_y = (prm.local_rows + 7) >> 3;
    const uint32_t ntiles_frame = tiles_x * tiles_y;
    const uint32_t ntiles = ntiles_frame * prm.nframes;      // a launch of several frames (rm_set_batch) is one pool of tiles
    const uint32_t home = blockIdx.x & (RM_QUEUE_HEADS - 1);
    const bool ordered = prm.ordered != 0;
    // a launch of more frames than the one before it: the lists hold the tiles of that one's frames, the rest
    // (ids hist_tiles..ntiles) go out behind them in natural order, from a pseudo-head of their own
    const uint32_t hist_tiles = ordered ? (prm.hist_tiles < ntiles ? prm.hist_tiles : ntiles) : ntiles;
    // (with list_reps > 1 the lists stand for every tile of the launch: no tail)
    const uint32_t last_cursor = ordered ? ((hist_tiles < ntiles && prm.list_reps <= 1) ? RM_QUEUE_HEADS + 1 : RM_QUEUE_HEADS) : RM_QUEUE_HEADS - 1;

    // ---- tile queue (wave-uniform; values that come from memory sit in a VGPR until they are needed)
    auto uniform = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    // One lane bumps a counter for the wave and the wave gets the old value.  Written as one asm
    // block that narrows EXEC to lane 0 around the atomic: as C++ (`if (lane == 0) v = atomic...`
    // followed by a cross-lane read) the compiler threads that branch through the queue's loops and
    // splits the wave into a lane-0 path and a path of the other lanes, each with a queue state of
    // its own - twice a hang on the device, in two different shapes of this code.  The block waits
    // for the value itself (the compiler does not track memory operations inside an asm, so the
    // result cannot be left in flight); the other wavefronts of the SIMD fill the microsecond, and
    // cheap tiles take their places RM_QUEUE_CHUNK at a time.
    auto bump = [&](uint32_t counter, uint32_t by) -> uint32_t {
        const uint32_t* base = kernarg_params()->set_cur;
        uint32_t ret, off = counter * 4u, add = by;
        unsigned long long save;
        asm volatile("s_mov_b64 %[save], exec\n\t"
                     "s_mov_b64 exec, 1\n\t"
                     "global_atomic_add %[ret], %[off], %[add], %[base] sc0\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "s_mov_b64 exec, %[save]"
                     : [ret] "=&v"(ret), [save] "=&s"(save)
                     : [off] "v"(off), [add] "v"(add), [base] "s"(base)
                     : "memory");
        return uniform(ret);
    };

    // stage A: the head being pulled from, and a counter value on its way
    // a_ticket: first place of the run of a_run places this wave holds in the head's sequence (one
    // place among the costly tiles, which must spread over the waves; RM_QUEUE_CHUNK among the cheap
    // ones at its end); a_pos: how many of them it has used
    // Cheap tiles go in runs and share one unordered list because the counters' throughput binds: a word
    // takes ~88 atomics per microsecond, and a 4K frame at a place per atomic is 16 200 pulls on each of the 8
    // heads - 0.18 ms of a frame that the arithmetic renders in 0.23 (measured per frame on chair 4K: 0.335 ->
    // 0.284 ms at 8 frames per launch in round 1; one frame per launch 0.42 -> 0.35 ms and 8K 1.52 -> 1.09 in
    // round 2).  Without history (natural order) the costly tiles are anywhere, so places go one at a time.
    const uint32_t pull_chunk = ordered ? RM_QUEUE_CHUNK : 1;
    uint32_t a_cursor = 0, a_head = home, a_total = 0, a_ticket = 0, a_run = 1, a_pos = 0;   // a_cursor: heads tried so far
    uint32_t a_cum[RM_COST_BUCKETS];                                   // running list lengths of a_head
    bool a_done = false;
    auto open_head = [&]() {
        // (a_cursor RM_QUEUE_HEADS: the overflow list; RM_QUEUE_HEADS + 1: the tail of tiles the lists do not hold)
        a_head = a_cursor < RM_QUEUE_HEADS ? ((home + a_cursor) & (RM_QUEUE_HEADS - 1)) : a_cursor;
        if (ordered && a_head > RM_QUEUE_HEADS) {
#pragma unroll
            for (int b = 0; b < RM_COST_BUCKETS; ++b) a_cum[b] = 0;
            a_total = ntiles - hist_tiles;
        } else if (ordered) {
            uint32_t sum = 0;
            KernargParams* kp = kernarg_params();
            const uint32_t* set_prev = kp->set_prev;
            const uint32_t list_cap = kp->list_cap, overflow_cap = kp->overflow_cap, reps = kp->list_reps;
#pragma unroll
            for (int b = 0; b < RM_COST_BUCKETS; ++b) {
                uint32_t n = 0;
                if (a_head < RM_QUEUE_HEADS) n = min(set_prev[RM_SET_COUNT(a_head, b)], list_cap);
                else if (b == 0)             n = min(set_prev[RM_SET_OVERFLOW], overflow_cap);
                sum += n * reps;             // places: every entry list_reps times
                a_cum[b] = sum;
            }
            a_total = sum;
        } else {        // natural order: head h owns tiles h, h+8, h+16, ...
#pragma unroll
            for (int b = 0; b < RM_COST_BUCKETS; ++b) a_cum[b] = 0;
            a_total = (a_head < RM_QUEUE_HEADS && ntiles > a_head) ? (ntiles - a_head + RM_QUEUE_HEADS - 1) / RM_QUEUE_HEADS : 0;
        }
        // A head that is already dry costs a load, not an atomic: at the end of a frame every wave
        // walks all the heads, and atomics on one word go through L2 one at a time (~88 per us).
        // The counter only grows, so a stale value can only send a wave to the atomic needlessly.
        const uint32_t seen = uniform(__hip_atomic_load(kernarg_params()->set_cur + RM_SET_PULL(a_head), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        a_pos = 0;
        a_run = seen >= a_cum[RM_CHEAP_BUCKET - 1] ? pull_chunk : 1;
        if (seen >= a_total) a_ticket = a_total;
        else a_ticket = bump(RM_SET_PULL(a_head), a_run);
    };
    // stage S: a tile id on its way (list entry), or already known (natural order)
    uint32_t s_val = 0, s_bucket = RM_COST_BUCKETS - 1;      // s_bucket: the cost bucket the previous launch filed the tile under
    bool s_valid = false;
    auto stage_s = [&]() {
        s_valid = false;
        while (!a_done) {
            const uint32_t c = uniform(a_ticket) + a_pos;
            if (c < a_total) {
                if (ordered && a_head > RM_QUEUE_HEADS) {
                    s_val = hist_tiles + c;
                    s_bucket = RM_COST_BUCKETS - 1;
                } else if (ordered) {
                    uint32_t b = 0, base = 0;
#pragma unroll
                    for (int k = 0; k + 1 < RM_COST_BUCKETS; ++k)
                        if (c >= a_cum[k]) { b = k + 1; base = a_cum[k]; }
                    KernargParams* kp = kernarg_params();
                    const uint32_t* list = kp->lists_prev + (size_t)(a_head * RM_COST_BUCKETS + b) * kp->list_cap;
                    const uint32_t reps = kp->list_reps;
                    uint32_t at = c - base, rep = 0;
                    if (reps > 1) {              // place -> (entry, which of the tiles it stands for): at / reps, at % reps
                        uint32_t q = (uint32_t)(((unsigned long long)at * kp->list_reps_magic) >> 32);      // floor or one less
                        rep = at - q * reps;
                        if (rep >= reps) { ++q; rep -= reps; }
                        at = q;
                    }
                    uint32_t entry = __builtin_nontemporal_load(list + at);
                    // (an unfilled place reads 0xffffffff and must stay out of range whatever is added to it)
                    if (reps > 1) entry = entry < hist_tiles ? entry + rep * hist_tiles : 0xffffffffu;
                    s_val = entry;
                    s_bucket = a_head < RM_QUEUE_HEADS ? b : RM_COST_BUCKETS - 1;
                } else {
                    s_val = a_head + RM_QUEUE_HEADS * c;
                }
                s_valid = true;
                if (++a_pos == a_run) {              // the run is used up: the next one is on its way while this tile marches
                    a_pos = 0;
                    a_run = c + 1 >= a_cum[RM_CHEAP_BUCKET - 1] ? pull_chunk : 1;
                    a_ticket = bump(RM_SET_PULL(a_head), a_run);
                }
                return;
            }
            // this head is dry: on to the next one, the overflow list last (it exists only with history)
            if (++a_cursor > last_cursor) { a_done = true; return; }
            open_head();
        }
    };
    open_head();
    stage_s();

    // optional per-wave diagnostics (rm_debug_wave_stats): trips, normal passes, cycles
    uint32_t st_evals = 0, st_norms = 0, st_norm_cycles = 0, st_longest = 0;
    uint32_t r_base = 0, r_left = 0;         // places this wave holds in its head's cheap list (filing, below)
    uint32_t tiles_done = 0;                 // tiles this wave rendered to the end (RM_STATUS_COUNTERS)
    const uint64_t st_t0 = prm.wave_stats ? __builtin_amdgcn_s_memtime() : 0;
    const uint64_t st_r0 = prm.wave_stats ? __builtin_amdgcn_s_memrealtime() : 0;        // the 100 MHz clock all XCDs share

    // Safety net for the persistent loop: every wave leaves after prm.watchdog ticks of the 100 MHz clock that
    // all XCDs share (s_memrealtime; 2^36 ticks are eleven minutes, unless a test lowered it), compared as a signed
    // difference.  (Until round 3 it read s_memtime, the shader-cycle counter: that one belongs to an XCD, and a
    // wavefront that is preempted - two processes sharing the GPU - may resume where the counter reads less than
    // it did at t_enter; the unsigned difference then looked like an expired watchdog.)  Whatever state it is in, so a defect in the
    // queue logic cannot hang the device.  It leaves a flag behind: the host reports the launch as
    // incomplete (RM_ERR_INCOMPLETE) instead of passing its frames on.
#ifndef RM_WATCHDOG_SHADER_CLOCK          // 1: the clock and the unsigned compare of rounds 1-2, to reproduce what they did
#define RM_WATCHDOG_SHADER_CLOCK 0       // (tools/two_process_stress.py, through RM_JIT_FLAGS); never in the product
#endif
#if RM_WATCHDOG_SHADER_CLOCK
    const uint64_t t_enter = __builtin_amdgcn_s_memtime();
#else
    const uint64_t t_enter = __builtin_amdgcn_s_memrealtime();
#endif
    bool expired = false;

    while (s_valid && !expired) {
        // ---- the next tile
        const uint32_t tile_id = uniform(s_val);
        const uint32_t tile_bucket = s_bucket;
        stage_s();
        if (tile_id >= ntiles) continue;             // cannot happen with a sound list; never index past the image
        uint32_t frame = 0, tile = tile_id;
        KernargParams* tp = kernarg_params();           // this tile's view of the launch parameters
        if (tp->nframes > 1) { frame = tile_id / ntiles_frame; tile = tile_id - frame * ntiles_frame; }
        Camera cam;                                      // 16 dwords of scalar loads
        {
            const __attribute__((address_space(4))) float* src = reinterpret_cast<const __attribute__((address_space(4))) float*>(&tp->cams[frame]);
            float* dst = reinterpret_cast<float*>(&cam);
#pragma unroll
            for (int k = 0; k < (int)(sizeof(Camera) / sizeof(float)); ++k) dst[k] = src[k];
        }
        const P3 light{cam.light[0], cam.light[1], cam.light[2]};
        const uint32_t tile_y = tile / tiles_x;
        const uint32_t px = (tile - tile_y * tiles_x) * 8 + (lane & 7);      // column
        const uint32_t ly = tile_y * 8 + (lane >> 3);                        // row in the local buffer
        uint32_t gy = ly;                                                    // row in the image
        const uint32_t band_rows = tp->band_rows, width = tp->width, height = tp->height;
        if (band_rows) {
            // bands are multiples of 8 rows (rm_set_bands) and a tile is 8 rows high: a tile lies in one band, and
            // which one is the same for all its lanes - a scalar division, not one per lane
            // (the context's band-th band is the image's: run * band_period + band_slot[k], rm_set_band_pattern)
            const uint32_t ly0 = tile_y * 8, shift = tp->band_shift, own = tp->band_own;
            const uint32_t band = shift != ~0u ? tile_y >> shift : ly0 / band_rows;
            uint32_t run = band, k = 0;
            if (own > 1) { run = (uint32_t)(((uint64_t)band * tp->band_own_magic) >> 32); k = band - run * own; }
            gy = (run * tp->band_period + tp->band_slot[k]) * band_rows + (ly0 - band * band_rows) + (lane >> 3);
        }
        const bool in_image = px < width && gy < height && ly < tp->local_rows;

        P3 pos, rd;                                      // (rd: the direction before its normalisation until a lane needs it)
        make_ray_raw(cam, width, height, (int)px, (int)gy, pos, rd);
        // The travelled distance at which a primary lane's march ends (ray_window: RM_MAX_DIST of :712 or less)
        // lives in the register that holds the lane's shadow term once it has turned to the light: a lane needs
        // one or the other, and the march loop has no vector register to spare (64 for 8 wavefronts per SIMD).
        float shadow;
        const bool may_hit = ray_window(tp, pos, rd, shadow);
        float& t_lim = shadow;                   // primary lanes only
        float t = 0.f, cd = 0.f, n_l = 0.f;
        // Which lanes march, and in which phase, is two wave-level masks in scalar registers, not a mode
        // per lane: the tests on them, their updates and the branches they feed are scalar instructions,
        // which issue beside the vector ones, and a compare or select costs 1.7 vector additions here
        // (tools/microbench/valu_peak.hip).  m_p: lanes on their primary march (:626); m_s: lanes on
        // their shadow march (:660).  A lane in neither is done.
        unsigned long long m_p = __builtin_amdgcn_ballot_w64(in_image && may_hit), m_s = 0;
        if (m_p) rd = ray_direction(rd);                 // :753-755, only in a tile that has a ray to march
        // The step caps (:626 I < 500, :660 J < 300) need no counter per lane either: a lane moves every
        // trip it is in a phase, every lane starts its primary march at trip 1 (so its step count is the
        // wave's trip count), and a lane that turns to the light at trip s has made J steps at trip s + J.
        uint32_t deadline = 0;                   // shadow lanes: the trip that is their step RM_SHADOW_STEPS
        // ...and a trip no later than any shadow lane's deadline, wave-level (every lane that turns at trip s gets
        // s + RM_SHADOW_STEPS): the per-lane compare (1.7 vector additions) is made by the general tail only, from that
        // trip on, and the loop of the shadow-only trips runs up to it without one (stop_s, below)
        uint32_t dl_min = 0xffffffffu;
        // what a shaded lane carries out of the loop: its luminance byte L (:694) under a flag bit; the pixel
        // is a function of L alone and is put together once, after the loop (one loop-carried value, not two)
        uint32_t shaded = 0;
        uint32_t trips = 0;                      // loop trips of this tile: its cost
        // A tile that runs long goes ahead of the others on its SIMD, a step at a time (s_setprio 1, 2, 3
        // after boost_age, RM_PRIO_STEP2 x and RM_PRIO_STEP3 x boost_age trips): the longest tiles set the
        // length of a frame, and where several long ones share a SIMD the longest should win.  Measured in
        // four interleaved sweeps (tools/prio_probe.py) against one step and against every long tile
        // straight at priority 3: chair 4K 0.286 / 0.284 -> 0.276 ms per frame, infinity 0.894 / 0.875 -> 0.821.
        uint32_t level = 0;
#if RM_PRIO_HISTORY
        // a tile the previous launch found long starts where it would have risen to: its first trips count as much as its last
        if (ordered && tile_bucket < 2) { level = 3 - tile_bucket; if (level == 3) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
#else
        (void)tile_bucket;
#endif
        const uint32_t boost_age = tp->boost_age;
        const bool boost_on = boost_age < (1u << 24);
        auto tier = [&](uint32_t k) { return k == 0 ? boost_age : k == 1 ? RM_PRIO_STEP2 * boost_age : RM_PRIO_STEP3 * boost_age; };
        uint32_t next_check = boost_age < 1023 ? boost_age + 1 : 1024;
        auto lanes = [](unsigned long long m) { return rm_lane_of_mask(m); };   // this lane's bit of a mask
        // a compare as a lane mask that stays one: masks or-ed together in C++ are merged by the optimiser into one
        // ballot of an or of booleans, which it then builds with a select and a second compare (3.4 vector additions)
        auto mask_of = [](bool b) { unsigned long long m = __builtin_amdgcn_ballot_w64(b); asm("" : "+s"(m)); return m; };
        // :662's `8 * cd < RAYS_MIN_DIST` without the product: scaling by 8 is exact in both directions (an overflowing
        // 8 * cd is +-inf and compares as the huge cd does; NaN is false both ways), so the test is cd < RAYS_MIN_DIST / 8
        constexpr float kShadowNear = RM_MIN_DIST * 0.125f;
        static_assert(kShadowNear * 8.f == RM_MIN_DIST, "RM_MIN_DIST / 8 must be exact");
        // shadow = min(shadow, (128*cd)/t) needs its division only where the quotient can be below `shadow`.  A shadow
        // lane has shadow >= .01 and t >= .04, so st = RN(shadow*t) > 0 is normal; n_lo = RN(cd * 128(1 - 2^-21)) > st then
        // gives 128*cd*(1 - 2^-21)(1 + 2^-24) > shadow*t*(1 - 2^-24), i.e. the exact quotient exceeds `shadow`, its
        // rounding is >= shadow and min() keeps `shadow` bit for bit (a tiny or negative cd fails the test: division).
        constexpr float kHardnessLow = RM_HARDNESS * (1.f - 0x1p-21f);

        // The scalar unit is shared by the four SIMDs of a CU and takes 4.3 cycles per instruction of a SIMD
        // (tools/microbench/valu_peak.hip, "mix"): a trip of ~70 vector instructions may hold ~35 scalar ones
        // before they, not the vector ones, set its length.  So the common trip - primary lanes only, none at a
        // surface, nothing due - has a tail of its own with a dozen scalar instructions, and everything else
        // (shadow lanes, normals, the step cap, priority and watchdog) goes through the general tail.
        // `due`: the trip at which the general tail must run whatever the lanes do (next_check; for primary
        // lanes also the step cap of :626).
        uint32_t due = next_check < (uint32_t)RM_PRIMARY_STEPS ? next_check : (uint32_t)RM_PRIMARY_STEPS;
        uint32_t stop_s = next_check;            // shadow-only trips: min(next_check, dl_min)
#ifdef RM_EXPERIMENT_NO_MARCH                    // measurements only (RM_JIT_FLAGS): what a launch costs without its marches
        m_p = 0;
#endif
        while (m_p | m_s) {
            P3 at;
            typename Scene::Keep keep;
            // The trips of a tile whose marching lanes are all in one phase and none at a surface run in a loop of
            // their phase's own (two loops, each with the scene's distance code in it: in one loop with a branch on
            // the phase the compiler carries the shadow phase's position, distance and shadow term in registers of
            // their own and copies them back and forth every trip, ten moves of 120 instructions).  A trip that is
            // anything else leaves its loop evaluated (trips, at, cd, keep) and the general tail finishes it.
            if (m_s == 0) {
                // ---- primary lanes only (:707-714)
                if constexpr (RM_NARROW_EXEC && Scene::kNarrow) {
                // The loop runs with the marching lanes as its execution mask: a compare inside it is a ballot over those
                // lanes as it stands, and a test on one is a branch on VCC - one scalar instruction where masking the
                // ballot with m_p, comparing and branching took three.  The CU's scalar unit serves four SIMDs and was as
                // busy as the vector units (infinity.sce: 7.3 G scalar instructions and branches a launch at one per cycle
                // and CU are the launch's 11.4 ms).  Every exit is wave-level.  `t >= t_lim` (:712) is looked at before
                // the next evaluation instead of behind the step: the same lanes leave with the same state, and the exit
                // says by itself whether a trip is in flight (no lane past its limit) or not.
                // (The trip count stays in a scalar register inside the region and leaves it through a vector register,
                // read back from a lane that ran the loop: a scalar the region changed is, to the compiler, a value per
                // lane behind it, and it would count in a vector register in the loop as well.)
                uint32_t trips_out = 0;
                const uint32_t due_s = uniform(due);
                if (lanes(m_p)) {
                    uint32_t n = trips;
                    for (;;) {
                        if (__builtin_amdgcn_ballot_w64(t >= t_lim)) break;                    // between trips: lanes leave
                        asm("s_add_u32 %0, %0, 1" : "+s"(n) : : "scc");                        // ++trips
                        at = pos;
                        cd = scene.dist_keep(at, scratch, ~0ull, keep);                        // DIST(root, pos) (:627)
                        if (__builtin_amdgcn_ballot_w64(cd < RM_MIN_DIST)) break;              // at a surface (:629); trip in flight
                        asm volatile("" ::: "memory");
                        if (n >= due_s) break;                                                 // trip in flight
                        pos.x += cd * rd.x; pos.y += cd * rd.y; pos.z += cd * rd.z;
                        t += cd;
                    }
                    asm("v_mov_b32 %0, %1" : "=v"(trips_out) : "s"(n));
                }
                trips = (uint32_t)__builtin_amdgcn_readlane((int)trips_out, (int)__builtin_ctzll(m_p));
                {
                    const unsigned long long left = __builtin_amdgcn_ballot_w64(t >= t_lim) & m_p;
                    if (left) { m_p &= ~left; continue; }
                }
                } else {
                for (;;) {
                    ++trips;
                    at = pos;
                    cd = scene.dist_keep(at, scratch, m_p, keep);             // DIST(root, pos) for the whole wave (:627): nothing reads a resting lane's value
                    if (__builtin_amdgcn_ballot_w64(cd < RM_MIN_DIST) & m_p) break;        // at a surface (:629)
                    asm volatile("" ::: "memory");         // two scalar branches: merged, the compiler evaluates the pair as lane masks
                    if (trips >= due) break;
                    if (lanes(m_p)) {
                        pos.x += cd * rd.x; pos.y += cd * rd.y; pos.z += cd * rd.z;
                        t += cd;
                    }
                    m_p &= ~__builtin_amdgcn_ballot_w64(t >= t_lim);           // a resting lane's stale t clears a bit that is clear
                    if (m_p == 0) break;
                }
                }
            } else if (m_p == 0) {
                // ---- shadow lanes only (:679-690)
                if constexpr (RM_NARROW_EXEC && Scene::kNarrow) {
                // (as above: the marching lanes are the execution mask, every exit is wave-level, and the ends of :680-690
                // are looked at between trips - before the next evaluation, or straight behind the quotient that darkened a lane)
                uint32_t trips_out = 0;
                const uint32_t stop_u = uniform(stop_s);
                if (lanes(m_s)) {
                    uint32_t n = trips;
                    for (;;) {
                        if (__builtin_amdgcn_ballot_w64(t >= as_f(shaded))) break;             // between trips: :689, or the shadow window's end
                        asm("s_add_u32 %0, %0, 1" : "+s"(n) : : "scc");
                        at = pos;
                        cd = scene.dist_keep(at, scratch, ~0ull, keep);                        // :661
                        if (__builtin_amdgcn_ballot_w64(cd < kShadowNear)) break;              // at a surface (:662); trip in flight
                        asm volatile("" ::: "memory");
                        if (n >= stop_u) break;                                                // trip in flight
                        bool dark = false;
                        if (__builtin_amdgcn_ballot_w64(!(cd * kHardnessLow > shadow * t))) {
                            float ns = RM_HARDNESS * cd;
                            asm volatile("; shadow quotient" : "+v"(ns));       // keeps the division behind the branch
                            shadow = min2(shadow, ns / t);
                            dark = __builtin_amdgcn_ballot_w64(shadow < 0.01f) != 0;           // :680
                        }
                        pos.x += cd * light.x; pos.y += cd * light.y; pos.z += cd * light.z;
                        t += cd;
                        if (dark) break;                                                       // between trips
                    }
                    asm("v_mov_b32 %0, %1" : "=v"(trips_out) : "s"(n));
                }
                trips = (uint32_t)__builtin_amdgcn_readlane((int)trips_out, (int)__builtin_ctzll(m_s));
                {
                    // (a marching lane's shadow is >= .01 until a quotient lowers it, and no lane is at its step cap before trip stop_s)
                    const unsigned long long left = (mask_of(t >= as_f(shaded)) | mask_of(shadow < 0.01f)) & m_s;
                    if (left) {                                                                // :693-701
                        if (lanes(left)) {
                            float lum = 255.f * (RM_MIN_LUM + (RM_MAX_LUM - RM_MIN_LUM) * (n_l > 0.f ? n_l : 0.f) * shadow);
                            shaded = 0x100u | to_u8(lum);
                        }
                        m_s &= ~left;
                        continue;
                    }
                }
                } else {
                for (;;) {
                    ++trips;
                    at = pos;
                    cd = scene.dist_keep(at, scratch, m_s, keep);             // :661
                    if (__builtin_amdgcn_ballot_w64(cd < kShadowNear) & m_s) break;        // at a surface (:662)
                    asm volatile("" ::: "memory");
                    if (trips >= stop_s) break;
                    // shadow = min(shadow, (128*cd)/t), the division only in a trip in which it can lower some
                    // lane's shadow, and then for all of them (see the general tail)
                    // (:680-690) the ends as compares straight into scalar masks (mask_of).  `shadow < .01` (:680) can
                    // only become true where the quotient was taken: a marching lane's shadow was >= .01 before
                    unsigned long long left = 0;
                    if (__builtin_amdgcn_ballot_w64(!(cd * kHardnessLow > shadow * t)) & m_s) {
                        if (lanes(m_s)) {
                            float ns = RM_HARDNESS * cd;
                            asm volatile("; shadow quotient" : "+v"(ns));       // keeps the division behind the branch
                            shadow = min2(shadow, ns / t);
                        }
                        left = mask_of(shadow < 0.01f);
                    }
                    if (lanes(m_s)) {
                        pos.x += cd * light.x; pos.y += cd * light.y; pos.z += cd * light.z;
                        t += cd;
                    }
                    left = (left | mask_of(t >= as_f(shaded))) & m_s;      // :689, or the shadow window's end; (no lane is at its step cap before trip stop_s)
                    if (left) {                                                        // :693-701
                        if (lanes(left)) {
                            float lum = 255.f * (RM_MIN_LUM + (RM_MAX_LUM - RM_MIN_LUM) * (n_l > 0.f ? n_l : 0.f) * shadow);
                            shaded = 0x100u | to_u8(lum);
                        }
                        m_s &= ~left;
                        if (m_s == 0) break;
                    }
                }
                }
            } else {
                // ---- lanes in both phases: one evaluation serves them all (:627 and :661)
                ++trips;
                at = pos;
                cd = scene.dist_keep(at, scratch, m_p | m_s, keep);
            }
            if ((m_p | m_s) == 0) break;

            // ---- the general tail
#ifdef RM_STATS_GENERAL            // diagnostics (RM_JIT_FLAGS): rm_debug_wave_stats' fourth word counts the trips that end here
            ++st_norm_cycles;
#endif
            // a tile that runs long goes first on its SIMD from trip boost_age + 1 on (the long tiles set the
            // length of the frame), and the watchdog looks at the clock every 1024 trips
            if (trips >= next_check) {
                if (boost_on && level < RM_PRIO_LEVELS && trips > tier(level)) {
                    do ++level; while (level < RM_PRIO_LEVELS && trips > tier(level));
                    if (level == 1) __builtin_amdgcn_s_setprio(1);
                    else if (level == 2) __builtin_amdgcn_s_setprio(2);
                    else __builtin_amdgcn_s_setprio(3);
                }
#if RM_WATCHDOG_SHADER_CLOCK
                if (__builtin_amdgcn_s_memtime() - t_enter > kernarg_params()->watchdog) { expired = true; break; }
#else
                if ((long long)(__builtin_amdgcn_s_memrealtime() - t_enter) > (long long)kernarg_params()->watchdog) { expired = true; break; }
#endif
                next_check = trips + 1024;
                if (boost_on && level < RM_PRIO_LEVELS && tier(level) >= trips && tier(level) - trips < 1024) next_check = tier(level) + 1;
                due = next_check < (uint32_t)RM_PRIMARY_STEPS ? next_check : (uint32_t)RM_PRIMARY_STEPS;
                if (trips >= (uint32_t)RM_PRIMARY_STEPS) due = trips + 1;      // past the primary cap every trip takes this tail (m_p is empty by then)
                next_check = uniform(next_check); due = uniform(due);          // (the compiler would keep them in vector registers)
            }
            unsigned long long near = __builtin_amdgcn_ballot_w64(cd < RM_MIN_DIST) & m_p;
            if (m_s) near |= __builtin_amdgcn_ballot_w64(cd < kShadowNear) & m_s;

            // ---- rare: some lane is at a surface.  It takes its normal in this trip, at the point the distance
            // was just evaluated at: the normal's transforms and child distances are the ones computed above
            // (Scene::Keep).  Then it either turns (primary -> shadow, shadow -> done) or marches on with the others.
            unsigned long long turn = 0, occluded = 0;          // primary lanes that turn to the light; shadow lanes behind an occluder
            unsigned long long unlit = 0;                       // primary lanes at a surface that faces away from the light: shaded at once
            if (near) {
                const uint64_t st_n0 = prm.wave_stats ? __builtin_amdgcn_s_memtime() : 0;
                ++st_norms;
                bool turns = false, hidden = false, dark = false;
                if (lanes(near)) {
                    const bool in_shadow = lanes(m_s);
                    // inside the surface the march direction stands in for the normal (:630-633, :663-666)
                    P3 n = in_shadow ? P3{-light.x, -light.y, -light.z} : P3{-rd.x, -rd.y, -rd.z};
                    if (!(cd < -0.f)) {                                                // :635-637, :668-670
                        P3 N = scene.norm_keep(at, scratch, keep);
                        float R_1 = r_length_3(N.x, N.y, N.z);
                        n = P3{R_1 * N.x, R_1 * N.y, R_1 * N.z};
                    }
                    const float facing = dot3(n.x, n.y, n.z, rd.x, rd.y, rd.z);
                    if (in_shadow) {                                                   // :673-676
                        if (facing < RM_MIN_DIST) { shadow = 0.f; hidden = true; }
                    } else if (facing < 0.f) {                                         // :640-658
                        n_l = dot3(n.x, n.y, n.z, light.x, light.y, light.z);
                        if (n_l > 0.f) {
                            shadow = 1.f;
                            const float k = (1 + RM_DELTA) * RM_MIN_DIST;
                            // (what depends on the light alone - k * light, the light's part of the shadow window - is worked
                            // out here, not in front of the loop, where the compiler would put it for every tile: most
                            // tiles never get here.  chair.sce 4K +1.8 %, the 8K sweep +16 %: four times the tiles.)
                            // (Not for the compiled distance programs: csg1024 4K lost 6 % with it, its kernel being
                            // short of registers around its normals' pass.)
                            P3 lw = light;
                            if constexpr (Scene::kNarrow) asm volatile("" : "+v"(lw.x), "+v"(lw.y), "+v"(lw.z));
                            pos.x += k * lw.x; pos.y += k * lw.y; pos.z += k * lw.z;
                            t = k;
                            deadline = trips + RM_SHADOW_STEPS;
                            turns = true;
                            // where this lane's shadow march may end (shadow_window), kept in the register that will hold
                            // its luminance once it is shaded: until then a marching lane's `shaded` says nothing
                            shaded = as_u(shadow_window(kernarg_params(), pos, lw, k));
                        } else {
                            // A surface that faces away from the light (N_L <= 0, or NaN): :694 multiplies SHADOW by
                            // (N_L > 0 ? N_L : 0) = 0, and SHADOW is finite whatever the shadow march does with it
                            // (:644 sets 1, :679 takes minima with finite quotients, :674 sets 0), so the product is a
                            // zero and LIGHT is 0xff * (MIN_LUM + 0) whichever way that march goes: the lane is shaded
                            // now and never marches it (chair.sce at 4K: a fifth of the hit pixels, 18 % of the shadow
                            // evaluations; the reference's march makes them for nothing).  Same pixel, same hit flag.
                            shaded = 0x100u | to_u8(255.f * (RM_MIN_LUM + (RM_MAX_LUM - RM_MIN_LUM) * 0.f * 1.f));
                            dark = true;
                        }
                    }
                }
                turn = __builtin_amdgcn_ballot_w64(turns);
                unlit = __builtin_amdgcn_ballot_w64(dark);
                if (turn && trips + RM_SHADOW_STEPS < dl_min) dl_min = trips + RM_SHADOW_STEPS;
                occluded = __builtin_amdgcn_ballot_w64(hidden);
#ifndef RM_STATS_GENERAL
                if (prm.wave_stats) st_norm_cycles += (uint32_t)(__builtin_amdgcn_s_memtime() - st_n0);
#endif
            }

            // ---- primary lanes march on (:707-714), those that did not just turn to the light
            const unsigned long long go_p = m_p & ~(turn | unlit);
            unsigned long long left_p = 0;
            if (go_p) {
                if (lanes(go_p)) {
                    pos.x += cd * rd.x; pos.y += cd * rd.y; pos.z += cd * rd.z;
                    t += cd;
                }
                left_p = trips >= RM_PRIMARY_STEPS ? go_p : (__builtin_amdgcn_ballot_w64(t >= t_lim) & go_p);
            }

            // ---- shadow lanes march on (:679-690), those not found behind an occluder
            const unsigned long long go_s = m_s & ~occluded;
            unsigned long long left_s = 0;
            if (go_s) {
                // shadow = min(shadow, (128*cd)/t).  The quotient matters only when it is below `shadow`, and
                // most steps it is not: n > shadow*t, by a margin that covers the product's rounding, means
                // n/t > shadow in exact arithmetic, so the rounded quotient is >= shadow and min() keeps
                // `shadow` bit for bit.  The division (12 instructions) runs only in a trip in which some
                // shadow lane needs it, and then for all of them (for the others min() changes nothing):
                // one scalar branch, no divergence.
                if (__builtin_amdgcn_ballot_w64(!(cd * kHardnessLow > shadow * t)) & go_s) {
                    if (lanes(go_s)) {
                        float ns = RM_HARDNESS * cd;
                        asm volatile("; shadow quotient" : "+v"(ns));           // keeps the division behind the branch
                        shadow = min2(shadow, ns / t);
                    }
                    left_s = mask_of(shadow < 0.01f);        // (only a quotient can bring a marching lane's shadow below .01)
                }
                // (:680-690) a lane whose shadow fell below .01 stops before the step, one that marches past
                // RM_MAX_DIST or has made RM_SHADOW_STEPS steps after it: either way it is shaded now and its
                // position is dead, so every shadow lane takes the step and the three ends are one mask
                if (lanes(go_s)) {
                    pos.x += cd * light.x; pos.y += cd * light.y; pos.z += cd * light.z;
                    t += cd;
                }
                left_s |= mask_of(t >= as_f(shaded));                // :689, or the shadow window's end
                if (trips >= dl_min) {
                    left_s |= mask_of(trips >= deadline);
                    // nobody due within 16 trips: look again then; else at every trip (through this tail) until they are
                    dl_min = (mask_of(deadline <= trips + 16) & go_s) ? trips + 1 : trips + 17;
                }
                left_s &= go_s;
            }

            // ---- lanes whose shadow march ended are shaded (:693-701)
            const unsigned long long fin = occluded | left_s;
            if (fin) {
                if (lanes(fin)) {
                    float lum = 255.f * (RM_MIN_LUM + (RM_MAX_LUM - RM_MIN_LUM) * (n_l > 0.f ? n_l : 0.f) * shadow);
                    shaded = 0x100u | to_u8(lum);
                }
            }
            m_p = go_p & ~left_p;
            m_s = (go_s & ~left_s) | turn;
            stop_s = next_check < dl_min ? next_check : dl_min;
        }
        st_evals += trips;
        st_longest = trips > st_longest ? trips : st_longest;
        if (level) __builtin_amdgcn_s_setprio(0);
        if (expired) break;

        if (in_image) {                                                            // :717
            const uint32_t hit = shaded >> 8;
            uint32_t pixel = 0xff000000u;                                          // :622
            // (most tiles hold no shaded pixel at all: one scalar test, no arithmetic)
            if (__builtin_amdgcn_ballot_w64(hit != 0) != 0 && hit) {               // :695-699
                // L is a whole number in [0, 255] and the material factors lie in (0, 1]: the casts of :697-699
                // need no saturation here, and the blue channel is L itself
                static_assert(RM_MATERIAL_R > 0.f && RM_MATERIAL_R <= 1.f && RM_MATERIAL_G > 0.f && RM_MATERIAL_G <= 1.f, "");
                const uint32_t Lu = shaded & 0xffu;
                const float L = (float)Lu;
                pixel = (uint32_t)(int)(L * RM_MATERIAL_R) | ((uint32_t)(int)(L * RM_MATERIAL_G) << 8) | (Lu << 16) | 0xff000000u;
            }
            KernargParams* ep = kernarg_params();
            *reinterpret_cast<uint32_t*>(ep->rgba + frame * ep->frame_stride + (size_t)ly * ep->pitch + (size_t)px * 4) = pixel;
            if (ep->hit) ep->hit[frame * ep->hit_frame_stride + (size_t)ly * ep->hit_pitch + (size_t)px] = (uint8_t)hit;
            // the luminance byte L of :694 is the blue channel (RM_MATERIAL_B is 1): a pixel is a function of L alone
            static_assert(RM_MATERIAL_B == 1.f, "the luminance plane reads L back from the blue channel");
            if (ep->lum) ep->lum[frame * ep->lum_frame_stride + (size_t)ly * ep->lum_pitch + (size_t)px] = (uint8_t)(pixel >> 16);
        }

        ++tiles_done;

        // ---- file the tile under its cost for the next frame
        KernargParams* fp = kernarg_params();
        if (fp->feedback) {
            const uint32_t b = trips >= 384 ? 0 : trips >= 192 ? 1 : trips >= 96 ? 2 : trips >= 48 ? 3 : trips >= 24 ? 4
                             : RM_CHEAP_BUCKET;
            uint32_t at;
            if (b == RM_CHEAP_BUCKET) {   // places in the cheap list are taken RM_QUEUE_CHUNK at a time
                if (r_left == 0) { r_base = uniform(bump(RM_SET_COUNT(home, RM_CHEAP_BUCKET), RM_QUEUE_CHUNK)); r_left = RM_QUEUE_CHUNK; }
                at = r_base + (RM_QUEUE_CHUNK - r_left);
                --r_left;
            } else {
                at = uniform(bump(RM_SET_COUNT(home, b), 1));
            }
            if (at < fp->list_cap) {
                fp->lists_cur[(size_t)(home * RM_COST_BUCKETS + b) * fp->list_cap + at] = tile_id;    // every lane, same word
            } else {                                 // list full: the overflow list takes any tile
                const uint32_t at2 = uniform(bump(RM_SET_OVERFLOW, 1));
                if (at2 < fp->overflow_cap)
                    fp->lists_cur[(size_t)RM_QUEUE_HEADS * RM_COST_BUCKETS * fp->list_cap + at2] = tile_id;
            }
        }
    }

    // places taken in the cheap list and not filled: marked, so that the next launch skips them
    if (!expired) {
        KernargParams* fp = kernarg_params();
        for (; r_left; --r_left) {
            const uint32_t at = r_base + (RM_QUEUE_CHUNK - r_left);
            if (at < fp->list_cap) fp->lists_cur[(size_t)(home * RM_COST_BUCKETS + RM_CHEAP_BUCKET) * fp->list_cap + at] = 0xffffffffu;
        }
    }

    // what the host checks a launch against: tiles rendered, and whether the watchdog ended anything
    if (lane == 0) {
        uint32_t* status = kernarg_params()->status;
        uint32_t* line = status + (blockIdx.x & (RM_STATUS_COUNTERS - 1)) * RM_STATUS_STRIDE;
        if (tiles_done) {
            __hip_atomic_fetch_add(line, tiles_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(line + RM_STATUS_TRIPS + kernarg_params()->stat_slot, st_evals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(line + RM_STATUS_LONGEST + kernarg_params()->stat_slot, st_longest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (expired) __hip_atomic_fetch_or(status + RM_STATUS_FLAG, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    if (prm.wave_stats && lane == 0) {
        uint32_t* o = prm.wave_stats + 8 * ((size_t)blockIdx.x * 4 + wave);
        const uint64_t st_t1 = __builtin_amdgcn_s_memtime();
        o[0] = st_evals; o[1] = st_norms; o[2] = (uint32_t)(st_t1 - st_t0); o[3] = st_norm_cycles;
        o[4] = (uint32_t)st_r0; o[5] = (uint32_t)__builtin_amdgcn_s_memrealtime(); o[6] = tiles_done; o[7] = st_longest;
    }
}

}  // namespace rm
