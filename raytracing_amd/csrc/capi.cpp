// capi.cpp - the C ABI declared in include/raymarching_amd.h: a thin host layer
// over HIP that takes the place of the reference's CUDA driver glue
// (Source/cuda_rays.cu).  No CPU rendering path exists here: without a HIP
// device every call that would render fails with the HIP error.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "../../include/raymarching_amd.h"
#include <cstdlib>

#include "jit.h"
#include "kernel_params.h"
#include "scene_build.h"

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return static_cast<int>(e_); } while (0)

struct rm_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t width = 0, height = 0;

    uint8_t* fb = nullptr;            // own framebuffer, width*4 pitch, `height` rows
    uint8_t* out = nullptr;           // where rm_process writes (fb or caller's buffer)
    size_t out_pitch = 0;
    uint8_t* hit = nullptr;           // optional hit mask, width pitch
    uint8_t* lum = nullptr;           // optional caller-owned luminance plane
    size_t lum_pitch = 0;

    rm::SceneBuild scene;
    rm::Slot* d_program = nullptr;
    rm::TreeNode* d_tree = nullptr;
    uint32_t n_nodes = 0;
    bool have_scene = false, have_info = false;
    rm::Camera cams[RM_MAX_BATCH];    // cams[f]: camera of frame f of a launch
    size_t batch = 1;                 // frames a launch may hold (rm_set_batch): what the planes and lists are sized for
    size_t nframes = 1;               // frames the next launch renders (rm_set_info: batch; rm_set_info_batch(n): n)
    size_t out_frame_stride = 0, lum_frame_stride = 0;   // 0: planes packed (pitch x local rows)
    size_t fb_bytes = 0, hit_bytes = 0;

    // row bands (rm_set_band_pattern): owner[k] renders the bands b with b % period == k
    size_t band_rows = 0; int rank = 0, nranks = 1;
    size_t band_period = 1;
    uint8_t band_owner[RM_MAX_BAND_PERIOD] = {0};
    size_t local_rows = 0;
    uint32_t evaluator = 0;
    size_t lds_bytes = 0;

    // tile scheduler (device_render.h): three counter sets used by launches seq, seq+1, seq+2 in turn
    // (a launch zeroes the next one's set itself, so no memset sits between launches) and two groups of
    // tile lists (the one the previous launch wrote, the one this launch writes).
    uint32_t* sched = nullptr;        // [3 sets][2 x ([8 heads][8 buckets][list_cap] + [overflow_cap])][status]
    uint32_t seq = 0;                 // launch sequence number
    uint32_t list_cap = 0, overflow_cap = 0;
    bool adaptive_grid = true;        // the grid follows the previous launch's work and longest tile (render_blocks)
    // wavefronts per (work / longest tile).  Round 3, after the trips got cheaper: chair.sce 4K, one frame per launch, 0.313 ms
    // at 1 (2048 wavefronts), 0.248-0.250 at 1.5-2.5 (4096-5120), 0.293 at 3; csg1024 4K 5.2 -> 4.8 ms at 2; the other
    // scenes and sizes within a few per cent either way (RM_GRID_FACTOR, read by rm_create)
    double grid_factor = 2.0;
    size_t tiles_per_wave = 8;        // grid: at least this many tiles per wavefront (RM_TILES_PER_WAVE, read by rm_create)
    size_t blocks_per_cu_forced = 0;  // experiments (RM_BLOCKS_PER_CU, read by rm_create): 0 = what the kernel's occupancy allows
    bool feedback = true;             // launches file their tiles by cost for the next launch
    bool history = false;             // the previous launch did, with the same tile numbering...
    size_t history_frames = 0;        // ...for this many frames (a launch of more frames: fill_params, hist_tiles / list_reps)
    // completion accounting (kernel_params.h, RM_STATUS_*): tiles launched so far against tiles the
    // wavefronts report finished, both modulo 2^32
    uint32_t* status_host = nullptr;  // pinned copy of the device's status block
    uint32_t expected_done = 0;
    uint32_t check_done = 0, check_expected = 0, check_flag = 0;   // what the last completion check compared (rm_debug_last_check)
    // what the last checked launches measured (judge_status), for the next launch's grid (render_blocks)
    uint32_t launches_unchecked = 0;  // launches since the last check
    size_t last_tiles = 0;            // tiles of the last launch
    double trips_per_tile = 0;        // mean of the last launch checked (0: nothing measured yet)
    uint32_t longest_tile = 0;        // trips of the longest tile of the last launch checked
    uint32_t last_slot = 0;           // RM_STATUS_LONGEST slot of the last launch
    uint32_t boost_age = 48;
    unsigned long long watchdog = 1ull << 36;
    bool cull = true;                 // bounding-sphere tests in union chains (scene_build.cpp)
    rm_scene_bound bound = {};        // the current scene's volume (scene_bound.cpp); kind RM_BOUND_NONE: none
    bool ray_windows = true;          // primary rays are marched only where they can be near it (rm_set_ray_windows)
    bool short_math = true;           // scenes may use div_const for divisors the device has proved
    uint32_t exact_divisors = 0;      // how many of the current scene's divisor constants were
    int cus = 1, blocks_per_cu = 1;   // of the device / of the ahead-of-time kernel at lds_bytes
    int vgprs = 0, scratch = 0;       // of the ahead-of-time (interpreter) kernel

    int jit_mode = 1;                 // 0 never, 1 when the scene is small enough, 2 required
    bool jit_active = false;          // the current scene runs on generated code
    rm::JitKernels jit;
    std::string jit_log;
};

namespace {

constexpr size_t kLdsLimit = 160 * 1024 - 1024;   // LDS per CU on gfx950, less a margin

int sqrt_sweep(rm_ctx* c, uint32_t first, unsigned long long count, unsigned long long* bad) {
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof *d));
    hipError_t e = hipMemsetAsync(d, 0, sizeof *d, c->stream);
    if (e == hipSuccess) e = rm::launch_verify_sqrt(first, count, d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(bad, d, sizeof *d, hipMemcpyDeviceToHost, c->stream);
    hipError_t s = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = s;
    (void)hipFree(d);
    return static_cast<int>(e);
}

size_t list_group_dwords(uint32_t list_cap, uint32_t overflow_cap) {
    return static_cast<size_t>(RM_QUEUE_HEADS) * RM_COST_BUCKETS * list_cap + overflow_cap;
}
size_t list_group_dwords(const rm_ctx* c) { return list_group_dwords(c->list_cap, c->overflow_cap); }

// dwords of one scheduler block: three counter sets, two list groups, the status block
size_t sched_dwords(uint32_t list_cap, uint32_t overflow_cap) {
    return 3 * RM_SET_DWORDS + 2 * list_group_dwords(list_cap, overflow_cap) + RM_STATUS_DWORDS;
}
// list capacities for `tiles` tiles per launch: a head sees about an eighth of the tiles, so a quarter
// per list leaves room for imbalance; what does not fit goes to the overflow list, which holds every tile
void list_caps(size_t tiles, uint32_t* list_cap, uint32_t* overflow_cap) {
    *overflow_cap = static_cast<uint32_t>(tiles);
    *list_cap = static_cast<uint32_t>(tiles / 4 + 16);
}

size_t count_local_rows(const rm_ctx* c, int rank) {
    if (!c->band_rows) return c->height;
    size_t rows = 0;
    for (size_t b = 0, y = 0; y < c->height; ++b, y += c->band_rows)
        if (c->band_owner[b % c->band_period] == rank)
            rows += (c->height - y < c->band_rows) ? c->height - y : c->band_rows;
    return rows;
}

// where launch `seq` finds its counter sets, list groups and status block inside a scheduler block
void point_at_schedule(rm::RenderParams& p, uint32_t* sched, uint32_t seq, uint32_t list_cap, uint32_t overflow_cap) {
    const uint32_t cur = seq % 3, prev = (seq + 2) % 3, next = (seq + 1) % 3;
    const size_t group = list_group_dwords(list_cap, overflow_cap);
    uint32_t* lists = sched + 3 * RM_SET_DWORDS;
    p.set_cur = sched + cur * RM_SET_DWORDS;
    p.set_prev = sched + prev * RM_SET_DWORDS;
    p.set_next = sched + next * RM_SET_DWORDS;
    p.lists_cur = lists + (seq & 1u) * group;
    p.lists_prev = lists + ((seq + 1) & 1u) * group;
    p.list_cap = list_cap;
    p.overflow_cap = overflow_cap;
    p.status = lists + 2 * group;
    p.stat_slot = seq & 1u;
}

// tiles of one frame of this context (its local rows)
size_t local_tiles(const rm_ctx* c) { return ((c->width + 7) / 8) * ((c->local_rows + 7) / 8); }

void fill_params(const rm_ctx* c, rm::RenderParams& p) {
    std::memset(&p, 0, sizeof p);
    p.program = c->d_program;
    p.tree = c->d_tree;
    p.n_nodes = c->n_nodes;
    p.width = static_cast<uint32_t>(c->width);
    p.height = static_cast<uint32_t>(c->height);
    p.local_rows = static_cast<uint32_t>(c->local_rows);
    p.band_rows = static_cast<uint32_t>(c->band_rows);
    p.band_period = static_cast<uint32_t>(c->band_period);
    p.band_own = 0;
    for (uint32_t k = 0; k < RM_MAX_BAND_OWN; ++k) p.band_slot[k] = 0;
    if (c->band_rows)
        for (size_t k = 0; k < c->band_period; ++k)
            if (c->band_owner[k] == c->rank && p.band_own < RM_MAX_BAND_OWN) p.band_slot[p.band_own++] = static_cast<uint32_t>(k);
    const size_t tile_rows = c->band_rows / 8;
    p.band_shift = ~0u;
    if (tile_rows && (tile_rows & (tile_rows - 1)) == 0)
        for (p.band_shift = 0; (size_t{1} << p.band_shift) < tile_rows;) ++p.band_shift;
    p.band_own_magic = p.band_own > 1 ? static_cast<uint32_t>(((uint64_t{1} << 32) + p.band_own - 1) / p.band_own) : 0u;
    p.evaluator = c->evaluator;
    p.rgba = c->out; p.pitch = c->out_pitch;
    p.hit = c->hit;  p.hit_pitch = c->width;
    p.lum = c->lum;  p.lum_pitch = c->lum_pitch;
    p.nframes = static_cast<uint32_t>(c->nframes);
    p.frame_stride = c->out_frame_stride ? c->out_frame_stride : c->out_pitch * c->local_rows;
    p.hit_frame_stride = c->width * c->local_rows;
    p.lum_frame_stride = c->lum_frame_stride ? c->lum_frame_stride : c->lum_pitch * c->local_rows;
    for (size_t f = 0; f < RM_MAX_BATCH; ++f) p.cams[f] = c->cams[f];
    p.boost_age = c->boost_age;
    p.watchdog = c->watchdog;
    if (c->ray_windows && c->bound.kind != RM_BOUND_NONE) {
        p.bound_kind = c->bound.kind;
        for (int k = 0; k < 3; ++k) { p.bound_c[k] = c->bound.c[k]; p.bound_u[k] = c->bound.u[k]; }
        p.bound_R = c->bound.r + RM_WINDOW_MARGIN;
    }
    point_at_schedule(p, c->sched, c->seq, c->list_cap, c->overflow_cap);
    p.feedback = c->feedback ? 1u : 0u;
    // the lists of the previous launch hold the tiles of its frames: a launch of fewer frames skips the
    // entries beyond its own, a launch of more frames takes the tiles of the frames beyond them in natural
    // order behind the lists (hist_tiles: the ids the lists hold)
    p.ordered = (c->feedback && c->history) ? 1u : 0u;
    p.hist_tiles = static_cast<uint32_t>(local_tiles(c) * std::min(c->nframes, c->history_frames));
    // ...or, when it holds a whole multiple of them, every entry stands for the same tile of the frames further on
    // too (list_reps): the frames of a camera path cost nearly the same tile by tile, so one frame's order serves a
    // launch of sixteen (hip::raymarching::ProcessBatch starts a path with a launch of one frame for that)
    p.list_reps = 1; p.list_reps_magic = 0;
    if (p.ordered && c->history_frames > 0 && c->nframes > c->history_frames && c->nframes % c->history_frames == 0) {
        const size_t reps = c->nframes / c->history_frames;
        p.list_reps = static_cast<uint32_t>(reps);
        p.list_reps_magic = static_cast<uint32_t>((uint64_t{1} << 32) / reps);       // reps >= 2
    }
}

size_t sched_bytes(const rm_ctx* c) { return sched_dwords(c->list_cap, c->overflow_cap) * sizeof(uint32_t); }
uint32_t* status_block(const rm_ctx* c) { return c->sched + 3 * RM_SET_DWORDS + 2 * list_group_dwords(c); }

// forget the previous frame's tile lists (the tile numbering changed, or feedback was switched)
hipError_t reset_schedule(rm_ctx* c) {
    c->history = false;
    return hipMemsetAsync(c->sched, 0, 3 * RM_SET_DWORDS * sizeof(uint32_t), c->stream);
}


// A launch is as long as its longest tile (one wavefront, a chain of dependent steps) or as its work spread
// over the chip, whichever is more, and a wavefront is the faster the fewer share its SIMD: with `trips`
// march-loop trips in all and `longest` in the costliest tile, more than trips / longest wavefronts only slow
// that tile down (chair 4K, one frame per launch: 8192 wavefronts 0.47 ms, 4096 0.35 ms; a launch of 8 frames,
// or infinity.sce, has work for every wavefront the chip holds).  Both figures are the previous launch's.
size_t wanted_waves(const rm_ctx* c, size_t tiles) {
    if (c->trips_per_tile <= 0 || c->longest_tile == 0) return ~size_t(0);
    return static_cast<size_t>(c->grid_factor * c->trips_per_tile * static_cast<double>(tiles) / c->longest_tile) + 1;
}

unsigned render_blocks(const rm_ctx* c, bool jit, size_t frames) {
    const size_t tiles = local_tiles(c) * frames;
    // Four wavefronts per workgroup, and enough tiles per wavefront that starting it and having it
    // find every queue head dry at the end is small beside its work: a share of a frame (row bands
    // on 8 ranks) must not launch a wavefront per two tiles, and a smaller grid leaves room for the
    // kernels of other frames in flight.
    const size_t tiles_per_wave = c->tiles_per_wave;
    const size_t want = (tiles + 4 * tiles_per_wave - 1) / (4 * tiles_per_wave);
    size_t per_cu = static_cast<size_t>(jit ? c->jit.blocks_per_cu : c->blocks_per_cu);
    if (c->blocks_per_cu_forced) per_cu = c->blocks_per_cu_forced;
    size_t resident = static_cast<size_t>(c->cus) * per_cu;
    if (!c->blocks_per_cu_forced && c->adaptive_grid) {
        const size_t waves = wanted_waves(c, tiles);                     // ~0 when nothing has been measured yet
        size_t by_work = waves > 4 * resident ? resident : (waves + 3) / 4;
        // the same number of workgroups on every CU: a wavefront is the slower the more share its SIMD, and the long
        // tiles should all run at one speed
        by_work = (by_work + c->cus - 1) / c->cus * c->cus;
        const size_t floor = static_cast<size_t>(c->cus) * 2;            // never below two workgroups per CU
        if (by_work < resident) resident = by_work > floor ? by_work : floor;
    }
    return static_cast<unsigned>(want < resident ? want : resident);
}

// one frame on the context's stream; *waves (optional) receives the wavefronts launched
int launch_frame(rm_ctx* c, uint32_t* wave_stats, size_t* waves) {
    rm::RenderParams p;
    fill_params(c, p);
    p.wave_stats = wave_stats;
    const bool jit = c->jit_active && c->evaluator == 0;
    const unsigned blocks = render_blocks(c, jit, c->nframes);
    if (waves) *waves = static_cast<size_t>(blocks) * 4;
    hipError_t e = jit ? rm::jit_launch_render(c->jit, p, blocks, c->stream)
                       : rm::launch_render(p, blocks, c->lds_bytes, c->stream);
    if (e == hipSuccess) {
        c->history = c->feedback; c->history_frames = c->nframes;
        c->expected_done += static_cast<uint32_t>(local_tiles(c) * c->nframes);
        ++c->launches_unchecked; c->last_tiles = local_tiles(c) * c->nframes;
        c->last_slot = c->seq & 1u;
        ++c->seq;
    }
    return static_cast<int>(e);
}

// Completion check, after a wait on the context's stream that followed `fetch_status`: every tile
// launched so far must have been reported finished and no wavefront may have left on the watchdog.
// Otherwise some frame since the last check is incomplete: the caller is told, and the next launch
// starts from a clean schedule in natural order (a lost tile is in no list).
hipError_t fetch_status(rm_ctx* c) {
    return hipMemcpyAsync(c->status_host, status_block(c), RM_STATUS_DWORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
}
int judge_status(rm_ctx* c) {
    uint32_t done = 0, longest = 0;
    uint64_t trips = 0;               // of the last launch alone (its slot): no running sum that could wrap between checks
    for (int k = 0; k < RM_STATUS_COUNTERS; ++k) {
        const uint32_t* line = c->status_host + k * RM_STATUS_STRIDE;
        done += line[0];
        trips += line[RM_STATUS_TRIPS + c->last_slot];
        if (line[RM_STATUS_LONGEST + c->last_slot] > longest) longest = line[RM_STATUS_LONGEST + c->last_slot];
    }
    if (c->launches_unchecked) {
        c->trips_per_tile = c->last_tiles ? static_cast<double>(trips) / static_cast<double>(c->last_tiles) : 0;
        c->longest_tile = longest;
        c->launches_unchecked = 0;
    }
    c->check_done = done; c->check_expected = c->expected_done; c->check_flag = c->status_host[RM_STATUS_FLAG];
    if (done == c->expected_done && c->status_host[RM_STATUS_FLAG] == 0) return RM_OK;
    c->trips_per_tile = 0; c->longest_tile = 0;
    c->expected_done = done;
    c->history = false;
    (void)hipMemsetAsync(c->sched, 0, 3 * RM_SET_DWORDS * sizeof(uint32_t), c->stream);
    (void)hipMemsetAsync(status_block(c) + RM_STATUS_FLAG, 0, sizeof(uint32_t), c->stream);
    (void)hipStreamSynchronize(c->stream);
    return RM_ERR_INCOMPLETE;
}

// Runs `count` numerators from bit pattern `first` through div_const for every divisor;
// mismatches[i] receives the count for c[i].  Synchronous.
hipError_t check_divisors(rm_ctx* c, const std::vector<float>& cs, uint32_t first, unsigned long long count,
                          std::vector<unsigned long long>& mismatches) {
    const size_t n = cs.size();
    mismatches.assign(n, 0);
    if (!n) return hipSuccess;
    std::vector<float> rcs(n);
    for (size_t i = 0; i < n; ++i) rcs[i] = 1.0f / cs[i];
    float* d_c = nullptr; unsigned long long* d_bad = nullptr;
    hipError_t e = hipMalloc(&d_c, 2 * n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d_bad, n * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpyAsync(d_c, cs.data(), n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_c + n, rcs.data(), n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, n * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = rm::launch_verify_div(d_c, d_c + n, static_cast<uint32_t>(n), first, count, d_bad, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(mismatches.data(), d_bad, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    hipError_t s = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = s;
    (void)hipFree(d_c); (void)hipFree(d_bad);
    return e;
}

// The divisor constants of a scene for which the device proves div_const exact: finite,
// 2^-20 <= |c| <= 2^20 (so that quotient and residual of a guarded numerator stay normal), and
// no mismatch over the binade [1,2) of numerators.
hipError_t exact_divisors(rm_ctx* c, const rm_node* nodes, size_t n, rm::DivisorSet& out) {
    out.clear();
    std::vector<float> cand;
    for (float v : rm::scene_divisors(nodes, n)) {
        const float m = v < 0 ? -v : v;
        if (m >= 0x1p-20f && m <= 0x1p20f) cand.push_back(v);      // false for NaN
    }
    std::vector<unsigned long long> bad;
    hipError_t e = check_divisors(c, cand, 0x3f800000u, 1u << 23, bad);
    if (e != hipSuccess) return e;
    for (size_t i = 0; i < cand.size(); ++i)
        if (bad[i] == 0) { uint32_t u; std::memcpy(&u, &cand[i], 4); out.insert(u); }
    return hipSuccess;
}

// Generated source for a flattened scene and the dynamic LDS its kernels need; false if the
// scene is too large for code generation (jit.h).
bool jit_source(const rm_node* nodes, size_t n, const rm::SceneBuild& built, const rm::DivisorSet* exact,
                bool force_program, std::string* src, size_t* lds_bytes) {
    if (built.reachable <= rm::kJitMaxNodes && !force_program) {
        *src = rm::generate_scene_source(nodes, n, built.seen.data(), exact,
                                         built.program.size() <= rm::kJitMaxSlots ? &built.program : nullptr, built.value_slots);
        *lds_bytes = 0;
        return true;
    }
    const size_t tree_bytes = built.tree.size() * sizeof(rm::TreeNode);
    // the tree table is only staged for the walker's normals; a union of primitives takes them from its own culled pass
    if (built.program.size() > rm::kJitMaxSlots || (built.union_rank.empty() && tree_bytes > rm::kJitMaxLds)) return false;
    *src = rm::generate_program_source(built.program, built.value_slots, built.slot_node, built.union_rank);
    *lds_bytes = built.union_rank.empty() ? tree_bytes : 0;
    return true;
}

int upload_scene(rm_ctx* c, const rm_node* nodes, size_t n) {
    HIP_TRY(hipSetDevice(c->device));
    rm::DivisorSet exact;
    if (c->short_math) HIP_TRY(exact_divisors(c, nodes, n, exact));
    rm::SceneBuild built;
    int rc = rm::build_scene(nodes, n, built, &exact, c->cull);
    if (rc != RM_OK) return rc;
    c->exact_divisors = static_cast<uint32_t>(exact.size());
    const size_t lds = built.tree.size() * sizeof(rm::TreeNode) +
                       static_cast<size_t>(built.value_slots) * RM_BLOCK_THREADS * sizeof(float);
    if (lds > kLdsLimit) return RM_ERR_TOO_LARGE;
    if (rm::configure_lds(lds) != hipSuccess) return RM_ERR_TOO_LARGE;

    HIP_TRY(hipSetDevice(c->device));
    rm::Slot* d_prog = nullptr; rm::TreeNode* d_tree = nullptr;
    HIP_TRY(hipMalloc(&d_prog, built.program.size() * sizeof(rm::Slot)));
    hipError_t e = hipMalloc(&d_tree, built.tree.size() * sizeof(rm::TreeNode));
    if (e != hipSuccess) { (void)hipFree(d_prog); return static_cast<int>(e); }
    e = hipMemcpyAsync(d_prog, built.program.data(), built.program.size() * sizeof(rm::Slot), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_tree, built.tree.data(), built.tree.size() * sizeof(rm::TreeNode), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(d_prog); (void)hipFree(d_tree); return static_cast<int>(e); }

    // per-scene generated code (jit.h); the generic program and tree stay uploaded
    // either way: known-answer hooks and rm_set_evaluator use them
    bool jit_ok = false;
    rm::JitKernels jit;
    c->jit_log.clear();
    std::string src;
    size_t jit_lds = 0;
    if (c->jit_mode != 0 && jit_source(nodes, n, built, &exact, c->jit_mode == 3, &src, &jit_lds)) {
        hipError_t je = rm::jit_compile(c->device, src, jit_lds, &jit, &c->jit_log);
        jit_ok = je == hipSuccess;
        if (!jit_ok && c->jit_mode >= 2) { (void)hipFree(d_prog); (void)hipFree(d_tree); return RM_ERR_JIT; }
    } else if (c->jit_mode >= 2) {
        (void)hipFree(d_prog); (void)hipFree(d_tree);
        return RM_ERR_TOO_LARGE;
    }
    if (c->jit.module) {                              // the previous scene's kernels: nothing of them in flight, then let go
        (void)hipStreamSynchronize(c->stream);
        rm::jit_release(c->jit);
    }
    c->jit_active = jit_ok;
    c->jit = jit_ok ? jit : rm::JitKernels();

    if (c->d_program) (void)hipFree(c->d_program);
    if (c->d_tree) (void)hipFree(c->d_tree);
    c->d_program = d_prog; c->d_tree = d_tree;
    c->n_nodes = static_cast<uint32_t>(built.tree.size());
    c->lds_bytes = lds;
    int occ = 0;
    c->blocks_per_cu = (rm::render_occupancy(lds, &occ) == hipSuccess && occ > 0) ? occ : 1;
    c->scene = std::move(built);
    rm::scene_bound(nodes, n, &c->bound);
    c->have_scene = true;
    c->trips_per_tile = 0; c->longest_tile = 0;        // measured on another scene
    return RM_OK;
}

// the context's own framebuffer, grown (never shrunk) to hold `frames` frames; the stream must be idle
int grow_framebuffer(rm_ctx* c, size_t frames) {
    const size_t need = c->width * c->height * 4 * frames;
    if (need <= c->fb_bytes) return RM_OK;
    uint8_t* fb = nullptr;
    HIP_TRY(hipMalloc(&fb, need));
    hipError_t e = hipMemset(fb, 0, need);
    if (e != hipSuccess) { (void)hipFree(fb); return static_cast<int>(e); }
    if (c->out == c->fb) c->out = fb;
    (void)hipFree(c->fb);
    c->fb = fb; c->fb_bytes = need;
    return RM_OK;
}

// ---- rm_process_static: the scheduler block in the library's file-scope device array ----------
// One per device (raymarch_kernels.hip: static_scratch_address).  Every call on a device shares that
// device's block, so the state that says what the block holds is per device too, and the calls of
// one process on one device take their turn under that device's lock (calls on different devices do
// not wait for each other; g_static_mutex only guards the map itself).
struct StaticSched {
    std::mutex mu;                    // held for a whole rm_process_static call on this device
    uint32_t* ptr = nullptr;          // device address of this device's copy of the array
    uint32_t seq = 0;                 // launches so far (the counter sets rotate as a context's do)
    bool history = false;             // the lists of launch seq-1 hold every tile of a frame of...
    size_t width = 0, local_rows = 0; // ...this tile numbering
    uint32_t expected_done = 0;       // tiles launched so far, modulo 2^32
};
std::mutex g_static_mutex;
std::map<int, StaticSched> g_static;

}  // namespace

extern "C" {

int rm_decode_bazo(const void* records, size_t n, int layout, rm_node* out) {
    return rm::decode_bazo(records, n, layout, out);
}

int rm_load_sce(const char* path, rm_node* out, size_t cap, size_t* n) {
    if (!path || !n) return RM_ERR_INVALID_ARG;
    std::vector<rm_node> nodes;
    int rc = rm::load_sce(path, nodes);
    if (rc != RM_OK) return rc;
    *n = nodes.size();
    if (out) std::memcpy(out, nodes.data(), sizeof(rm_node) * (nodes.size() < cap ? nodes.size() : cap));
    return RM_OK;
}

int rm_save_sce(const char* path, const rm_node* nodes, size_t n) { return rm::save_sce(path, nodes, n); }

int rm_create(int device, size_t width, size_t height, rm_ctx** out) {
    if (!out || width == 0 || height == 0 || width > (1u << 20) || height > (1u << 20) ||
        width * height > (size_t(1) << 30)) return RM_ERR_INVALID_ARG;       // tile and pixel indices are 32-bit
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    rm_ctx* c = new (std::nothrow) rm_ctx;
    if (!c) return static_cast<int>(hipErrorOutOfMemory);
    c->device = device; c->width = width; c->height = height;
    // experiment knobs, read here once: nothing on the launch path looks at the environment
    if (const char* env = std::getenv("RM_ADAPTIVE_GRID")) c->adaptive_grid = std::atoi(env) != 0;
    if (const char* env = std::getenv("RM_GRID_FACTOR")) if (std::atof(env) > 0) c->grid_factor = std::atof(env);
    if (const char* env = std::getenv("RM_TILES_PER_WAVE")) if (std::atoi(env) > 0) c->tiles_per_wave = static_cast<size_t>(std::atoi(env));
    if (const char* env = std::getenv("RM_BLOCKS_PER_CU")) if (std::atoi(env) > 0) c->blocks_per_cu_forced = static_cast<size_t>(std::atoi(env));
    if (const char* env = std::getenv("RM_RAY_WINDOWS")) c->ray_windows = std::atoi(env) != 0;          // experiments only
    if (const char* env = std::getenv("RM_JIT")) c->jit_mode = std::atoi(env) < 0 ? 0 : std::atoi(env) > 3 ? 3 : std::atoi(env);
    hipError_t e = hipStreamCreate(&c->own_stream);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    c->fb_bytes = width * height * 4;
    if (e == hipSuccess) e = hipMalloc(&c->fb, c->fb_bytes);
    if (e == hipSuccess) e = hipMemset(c->fb, 0, c->fb_bytes);
    list_caps(((width + 7) / 8) * ((height + 7) / 8), &c->list_cap, &c->overflow_cap);
    if (e == hipSuccess) e = hipMalloc(&c->sched, sched_bytes(c));
    if (e == hipSuccess) e = hipMemset(c->sched, 0, sched_bytes(c));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->status_host), RM_STATUS_DWORDS * sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device);
    if (e == hipSuccess) e = rm::kernel_attributes(&c->vgprs, &c->scratch);
    if (e != hipSuccess) { rm_destroy(c); return e == hipErrorInvalidDeviceFunction ? RM_ERR_NO_KERNEL : static_cast<int>(e); }
    c->stream = c->own_stream;
    // sqrt_exact must be IEEE sqrtf on this device: two binades cover every mantissa and both
    // exponent parities (the sequence scales exactly by powers of four).  Once per device.
    {
        static std::mutex mu; static std::set<int> proved;
        std::lock_guard<std::mutex> lock(mu);
        if (!proved.count(device)) {
            unsigned long long bad = 0;
            int rc = sqrt_sweep(c, 0x3f800000u, 1ull << 24, &bad);
            if (rc != RM_OK || bad != 0) { rm_destroy(c); return rc != RM_OK ? rc : RM_ERR_ARITHMETIC; }
            proved.insert(device);
        }
    }
    // rm_process_static's scheduler block: its address on this device, looked up once, here at set-up
    {
        StaticSched* found = nullptr;
        { std::lock_guard<std::mutex> lock(g_static_mutex); found = &g_static[device]; }      // (map nodes never move)
        StaticSched& st = *found;
        std::lock_guard<std::mutex> lock(st.mu);
        if (!st.ptr) {
            hipError_t se = rm::static_scratch_address(device, &st.ptr);
            if (se == hipSuccess) se = hipMemset(st.ptr, 0, (3 * RM_SET_DWORDS) * sizeof(uint32_t));
            if (se == hipSuccess) se = hipMemset(st.ptr + RM_STATIC_DWORDS - RM_STATUS_DWORDS, 0, RM_STATUS_DWORDS * sizeof(uint32_t));
            if (se != hipSuccess) { st.ptr = nullptr; rm_destroy(c); return static_cast<int>(se); }
        }
    }
    c->out = c->fb; c->out_pitch = width * 4;
    c->local_rows = height;
    *out = c;
    return RM_OK;
}

int rm_destroy(rm_ctx* c) {
    if (!c) return RM_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->stream && c->stream != c->own_stream) (void)hipStreamSynchronize(c->stream);
    rm::jit_release(c->jit);
    if (c->d_program) (void)hipFree(c->d_program);
    if (c->d_tree) (void)hipFree(c->d_tree);
    if (c->fb) (void)hipFree(c->fb);
    if (c->hit) (void)hipFree(c->hit);
    if (c->sched) (void)hipFree(c->sched);
    if (c->status_host) (void)hipHostFree(c->status_host);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return RM_OK;
}

int rm_set_stream(rm_ctx* c, void* hip_stream) {
    if (!c) return RM_ERR_INVALID_ARG;
    hipStream_t to = static_cast<hipStream_t>(hip_stream);
    if (to == c->stream) return RM_OK;
    // Completion accounting and the scheduler's counter sets follow stream order: launches still in flight on the
    // old stream are waited for (and checked) before anything is queued on the new one.
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(fetch_status(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int rc = judge_status(c);
    c->stream = to;
    return rc;
}

int rm_set_output(rm_ctx* c, void* device_rgba, size_t pitch_bytes) {
    if (!c) return RM_ERR_INVALID_ARG;
    if (!device_rgba) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        int rc = grow_framebuffer(c, c->batch);          // room for every frame of a launch (rm_set_batch)
        if (rc != RM_OK) return rc;
        c->out = c->fb; c->out_pitch = c->width * 4;
        return RM_OK;
    }
    if (pitch_bytes < c->width * 4 || (pitch_bytes & 3) || (reinterpret_cast<uintptr_t>(device_rgba) & 3))
        return RM_ERR_INVALID_ARG;
    c->out = static_cast<uint8_t*>(device_rgba); c->out_pitch = pitch_bytes;
    return RM_OK;
}

int rm_set_luminance_output(rm_ctx* c, void* device_l8, size_t pitch_bytes) {
    if (!c) return RM_ERR_INVALID_ARG;
    if (!device_l8) { c->lum = nullptr; c->lum_pitch = 0; return RM_OK; }
    if (pitch_bytes < c->width) return RM_ERR_INVALID_ARG;
    c->lum = static_cast<uint8_t*>(device_l8); c->lum_pitch = pitch_bytes;
    return RM_OK;
}

int rm_assemble_luminance_batch(rm_ctx* c, const void* device_parts, size_t rank_stride_bytes, size_t part_frame_stride_bytes,
                                size_t part_pitch_bytes, size_t frames, void* device_rgba, size_t pitch_bytes,
                                size_t rgba_frame_stride_bytes, void* hip_stream) {
    if (!c || !device_parts || !device_rgba || frames == 0 || frames > RM_MAX_BATCH) return RM_ERR_INVALID_ARG;
    if (part_pitch_bytes < c->width || pitch_bytes < c->width * 4 || (pitch_bytes & 3) || (rgba_frame_stride_bytes & 3) ||
        (reinterpret_cast<uintptr_t>(device_rgba) & 3))
        return RM_ERR_INVALID_ARG;
    rm::AssembleParams p{};
    size_t rows_max = c->height;
    if (c->band_rows) {
        // every rank's part must hold that rank's rows, every frame's plane the largest share
        rows_max = 0;
        for (size_t k = 0; k < c->band_period; ++k) {
            const uint8_t r = c->band_owner[k];
            p.owner[k] = r;
            p.index[k] = p.own[r]++;
        }
        for (int r = 0; r < c->nranks; ++r) rows_max = std::max(rows_max, count_local_rows(c, r));
        if (c->nranks > 1 && rank_stride_bytes < rows_max * part_pitch_bytes) return RM_ERR_INVALID_ARG;
    }
    if (frames > 1 && (part_frame_stride_bytes < rows_max * part_pitch_bytes || rgba_frame_stride_bytes < c->height * pitch_bytes))
        return RM_ERR_INVALID_ARG;
    p.parts = static_cast<const uint8_t*>(device_parts);
    p.rank_stride = rank_stride_bytes; p.part_frame_stride = part_frame_stride_bytes; p.part_pitch = part_pitch_bytes;
    p.rgba = static_cast<uint8_t*>(device_rgba);
    p.pitch = pitch_bytes; p.rgba_frame_stride = rgba_frame_stride_bytes;
    p.width = static_cast<uint32_t>(c->width); p.height = static_cast<uint32_t>(c->height);
    p.band_rows = static_cast<uint32_t>(c->band_rows); p.period = static_cast<uint32_t>(c->band_period);
    p.vector = ((reinterpret_cast<uintptr_t>(device_parts) | rank_stride_bytes | part_frame_stride_bytes | part_pitch_bytes) & 3) == 0 &&
               ((reinterpret_cast<uintptr_t>(device_rgba) | pitch_bytes | rgba_frame_stride_bytes) & 15) == 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(rm::launch_assemble_luminance(p, static_cast<uint32_t>(frames), static_cast<hipStream_t>(hip_stream)));
    return RM_OK;
}

int rm_assemble_luminance(rm_ctx* c, const void* device_parts, size_t rank_stride_bytes, size_t part_pitch_bytes,
                          void* device_rgba, size_t pitch_bytes, void* hip_stream) {
    return rm_assemble_luminance_batch(c, device_parts, rank_stride_bytes, 0, part_pitch_bytes, 1, device_rgba, pitch_bytes, 0, hip_stream);
}

int rm_set_primitives(rm_ctx* c, const void* bazo_records, size_t n, int layout) {
    if (!c || !bazo_records || n == 0) return RM_ERR_INVALID_ARG;
    if (n > RM_MAX_NODES) return RM_ERR_TOO_LARGE;
    std::vector<rm_node> nodes(n);
    int rc = rm::decode_bazo(bazo_records, n, layout, nodes.data());
    if (rc != RM_OK) return rc;
    return upload_scene(c, nodes.data(), n);
}

int rm_set_nodes(rm_ctx* c, const rm_node* nodes, size_t n) {
    if (!c || !nodes || n == 0) return RM_ERR_INVALID_ARG;
    return upload_scene(c, nodes, n);
}

namespace {
void camera_from(const rm_rays_info* info, rm::Camera& k) {
    k.depth = static_cast<float>(info->Depth);           // Source/cuda_kernels.cu:744
    const rm_point* src[5] = { &info->LightSource, &info->StartPos, &info->StartDir, &info->StartWVec, &info->StartHVec };
    float* dst[5] = { k.light, k.pos, k.dir, k.wvec, k.hvec };
    for (int i = 0; i < 5; ++i) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
}
}  // namespace

int rm_set_info(rm_ctx* c, const rm_rays_info* info) {
    if (!c || !info) return RM_ERR_INVALID_ARG;
    camera_from(info, c->cams[0]);
    for (size_t f = 1; f < RM_MAX_BATCH; ++f) c->cams[f] = c->cams[0];      // every frame of a batch, until told otherwise
    c->nframes = c->batch;
    c->have_info = true;
    return RM_OK;
}

int rm_set_info_batch(rm_ctx* c, const rm_rays_info* infos, size_t n) {
    if (!c || !infos || n == 0 || n > c->batch) return RM_ERR_INVALID_ARG;
    for (size_t f = 0; f < n; ++f) camera_from(infos + f, c->cams[f]);
    for (size_t f = n; f < RM_MAX_BATCH; ++f) c->cams[f] = c->cams[n - 1];
    c->nframes = n;            // a launch that is not full: no reallocation, and the tile order learnt so far stays
    c->have_info = true;
    return RM_OK;
}

int rm_set_batch(rm_ctx* c, size_t frames, size_t rgba_frame_stride_bytes, size_t luminance_frame_stride_bytes) {
    if (!c || frames == 0 || frames > RM_MAX_BATCH) return RM_ERR_INVALID_ARG;
    const size_t tiles = ((c->width + 7) / 8) * ((c->height + 7) / 8);
    if (tiles * frames > (size_t(1) << 30)) return RM_ERR_INVALID_ARG;       // tile ids are 32-bit
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // The scheduler's lists hold a launch's tiles and the context's own planes hold all its frames: both
    // only ever grow, so going back to fewer frames (or launching fewer with rm_set_info_batch) allocates
    // nothing and keeps the tile order learnt so far.
    uint32_t list_cap = 0, overflow_cap = 0;
    list_caps(tiles * frames, &list_cap, &overflow_cap);
    if (sched_dwords(list_cap, overflow_cap) > sched_dwords(c->list_cap, c->overflow_cap)) {
        const size_t bytes = sched_dwords(list_cap, overflow_cap) * sizeof(uint32_t);
        uint32_t* sched = nullptr;
        HIP_TRY(hipMalloc(&sched, bytes));
        hipError_t e = hipMemset(sched, 0, bytes);
        if (e != hipSuccess) { (void)hipFree(sched); return static_cast<int>(e); }
        (void)hipFree(c->sched);
        c->sched = sched; c->overflow_cap = overflow_cap; c->list_cap = list_cap;
        c->history = false; c->seq = 0; c->expected_done = 0;
        c->launches_unchecked = 0; c->last_tiles = 0; c->trips_per_tile = 0; c->longest_tile = 0;
    }
    if (c->out == c->fb) {               // a caller-owned output needs no planes of the context's own
        int rc = grow_framebuffer(c, frames);
        if (rc != RM_OK) return rc;
    }
    if (c->hit && c->width * c->height * frames > c->hit_bytes) {
        uint8_t* hit = nullptr;
        HIP_TRY(hipMalloc(&hit, c->width * c->height * frames));
        hipError_t e = hipMemset(hit, 0, c->width * c->height * frames);
        if (e != hipSuccess) { (void)hipFree(hit); return static_cast<int>(e); }
        (void)hipFree(c->hit);
        c->hit = hit; c->hit_bytes = c->width * c->height * frames;
    }
    c->batch = frames;
    c->nframes = frames;
    c->out_frame_stride = rgba_frame_stride_bytes;
    c->lum_frame_stride = luminance_frame_stride_bytes;
    return RM_OK;
}

int rm_set_band_pattern(rm_ctx* c, size_t band_rows, const unsigned char* owner, size_t period, int rank, int nranks) {
    if (!c) return RM_ERR_INVALID_ARG;
    if (band_rows != 0) {
        if ((band_rows % 8) || !owner || period < 1 || period > RM_MAX_BAND_PERIOD || nranks < 1 || nranks > RM_MAX_BAND_PERIOD ||
            rank < 0 || rank >= nranks) return RM_ERR_INVALID_ARG;
        size_t mine = 0;
        for (size_t k = 0; k < period; ++k) {
            if (owner[k] >= nranks) return RM_ERR_INVALID_ARG;
            mine += owner[k] == rank;
        }
        if (mine > RM_MAX_BAND_OWN) return RM_ERR_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(c->device));
    if (band_rows == 0) { c->band_rows = 0; c->rank = 0; c->nranks = 1; c->band_period = 1; c->band_owner[0] = 0; c->local_rows = c->height; }
    else {
        c->band_rows = band_rows; c->rank = rank; c->nranks = nranks; c->band_period = period;
        for (size_t k = 0; k < period; ++k) c->band_owner[k] = owner[k];
        c->local_rows = count_local_rows(c, rank);
    }
    HIP_TRY(reset_schedule(c));          // tiles are numbered over the local rows
    return RM_OK;
}

int rm_set_bands(rm_ctx* c, size_t band_rows, int rank, int nranks) {
    if (band_rows == 0) return rm_set_band_pattern(c, 0, nullptr, 0, 0, 1);
    if (nranks < 1 || nranks > RM_MAX_BAND_PERIOD) return RM_ERR_INVALID_ARG;
    unsigned char owner[RM_MAX_BAND_PERIOD];
    for (int k = 0; k < nranks; ++k) owner[k] = static_cast<unsigned char>(k);
    return rm_set_band_pattern(c, band_rows, owner, static_cast<size_t>(nranks), rank, nranks);
}

int rm_local_rows(rm_ctx* c, size_t* rows) {
    if (!c || !rows) return RM_ERR_INVALID_ARG;
    *rows = c->local_rows;
    return RM_OK;
}

int rm_enable_hit_mask(rm_ctx* c, int enable) {
    if (!c) return RM_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    if (enable && !c->hit) {
        c->hit_bytes = c->width * c->height * c->batch;
        HIP_TRY(hipMalloc(&c->hit, c->hit_bytes));
        HIP_TRY(hipMemset(c->hit, 0, c->hit_bytes));       // a pixel no tile wrote reads as a miss, not as what the memory held
    }
    if (!enable && c->hit) { HIP_TRY(hipStreamSynchronize(c->stream)); (void)hipFree(c->hit); c->hit = nullptr; }
    return RM_OK;
}

int rm_process_async(rm_ctx* c) {
    if (!c) return RM_ERR_INVALID_ARG;
    if (!c->have_scene || !c->have_info) return RM_ERR_NOT_READY;
    if (c->local_rows == 0) return RM_OK;
    // launch-shape checks: everything the kernel indexes must match what was allocated
    if (!c->out || c->out_pitch < c->width * 4 || !c->d_program || !c->d_tree ||
        c->n_nodes != c->scene.tree.size() || c->local_rows > c->height) return RM_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    return launch_frame(c, nullptr, nullptr);
}

int rm_synchronize(rm_ctx* c) {
    if (!c) return RM_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(fetch_status(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return judge_status(c);
}

int rm_process(rm_ctx* c, float* elapsed_ms) {
    if (!c) return RM_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    int rc = rm_process_async(c);
    if (rc != RM_OK) return rc;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(fetch_status(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (elapsed_ms) HIP_TRY(hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
    return judge_status(c);
}

int rm_process_static(rm_ctx* c, const rm_rays_info* info, void* device_rgba, size_t pitch_bytes, float* elapsed_ms) {
    if (!c || !info || !device_rgba) return RM_ERR_INVALID_ARG;
    if (pitch_bytes < c->width * 4 || (pitch_bytes & 3) || (reinterpret_cast<uintptr_t>(device_rgba) & 3)) return RM_ERR_INVALID_ARG;
    if (!c->have_scene) return RM_ERR_NOT_READY;
    if (!c->d_program || !c->d_tree || c->n_nodes != c->scene.tree.size() || c->local_rows > c->height) return RM_ERR_INVALID_ARG;
    const size_t tiles = local_tiles(c);
    if (tiles > RM_STATIC_MAX_TILES) return RM_ERR_TOO_LARGE;        // the array holds lists for that many tiles, no more
    if (c->local_rows == 0) return RM_OK;
    HIP_TRY(hipSetDevice(c->device));
    StaticSched* found = nullptr;
    { std::lock_guard<std::mutex> lock(g_static_mutex); found = &g_static[c->device]; }
    StaticSched& st = *found;
    std::lock_guard<std::mutex> lock(st.mu);                         // this device's block is ours until the frame is in
    if (!st.ptr) return RM_ERR_NOT_READY;                            // rm_create looked it up: cannot happen
    rm::RenderParams p;
    fill_params(c, p);
    p.rgba = static_cast<uint8_t*>(device_rgba); p.pitch = pitch_bytes;
    p.hit = nullptr; p.lum = nullptr; p.wave_stats = nullptr;
    p.nframes = 1;
    camera_from(info, p.cams[0]);
    point_at_schedule(p, st.ptr, st.seq, RM_STATIC_LIST_CAP, RM_STATIC_MAX_TILES);
    p.ordered = (c->feedback && st.history && st.width == c->width && st.local_rows == c->local_rows) ? 1u : 0u;
    p.hist_tiles = static_cast<uint32_t>(tiles);
    p.list_reps = 1; p.list_reps_magic = 0;
    const bool jit = c->jit_active && c->evaluator == 0;
    const unsigned blocks = render_blocks(c, jit, 1);
    // The block keeps what the last call on this device left in it.  This launch's own counter set must
    // start at zero (a launch zeroes the set of the launch after it, but the call before this one may
    // have failed half-way); the lists and the previous set are read only when `ordered` says they are sound.
    HIP_TRY(hipMemsetAsync(p.set_cur, 0, RM_SET_DWORDS * sizeof(uint32_t), c->stream));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(jit ? rm::jit_launch_render(c->jit, p, blocks, c->stream) : rm::launch_render(p, blocks, c->lds_bytes, c->stream));
    // From here on a kernel is using the shared block: whatever fails below, the call does not return (and let
    // the next call on this device in) before the stream has been waited for, and a failure leaves the block
    // marked as holding nothing the next call may trust.
    ++st.seq;
    st.history = c->feedback; st.width = c->width; st.local_rows = c->local_rows;
    st.expected_done += static_cast<uint32_t>(tiles);
    hipError_t e = hipEventRecord(c->ev1, c->stream);
    hipError_t e2 = hipMemcpyAsync(c->status_host, p.status, RM_STATUS_DWORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = e2;
    e2 = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = e2;
    if (e == hipSuccess && elapsed_ms) e = hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1);
    if (e != hipSuccess) {
        // the status copy cannot be relied on: take the device's count as it stands for the next call to build on
        st.history = false;
        uint32_t done = 0;
        if (hipMemcpy(c->status_host, p.status, RM_STATUS_DWORDS * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess)
            for (int k = 0; k < RM_STATUS_COUNTERS; ++k) done += c->status_host[k * RM_STATUS_STRIDE];
        st.expected_done = done;
        return static_cast<int>(e);
    }
    uint32_t done = 0;
    for (int k = 0; k < RM_STATUS_COUNTERS; ++k) done += c->status_host[k * RM_STATUS_STRIDE];
    if (done == st.expected_done && c->status_host[RM_STATUS_FLAG] == 0) return RM_OK;
    st.expected_done = done;
    st.history = false;
    (void)hipMemsetAsync(p.status + RM_STATUS_FLAG, 0, sizeof(uint32_t), c->stream);
    (void)hipStreamSynchronize(c->stream);
    return RM_ERR_INCOMPLETE;
}

// Device rows to host memory.  Rows that are contiguous on both sides go as one linear copy: into pageable
// memory the runtime's 2-D path moves 1.9 GB/s on the GPU box, the linear one pipelines through its pinned staging.
static int read_rows(rm_ctx* c, void* host, size_t host_pitch, const void* dev, size_t dev_pitch, size_t row_bytes) {
    // The frame is waited for first and the copy is the blocking kind: the destination may be pageable memory,
    // and the call must not return before the last byte is in it.
    HIP_TRY(fetch_status(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int complete = judge_status(c);
    if (complete != RM_OK) return complete;          // some frame since the last check lacks tiles: nothing is copied
    if (host_pitch == row_bytes && dev_pitch == row_bytes)
        HIP_TRY(hipMemcpy(host, dev, row_bytes * c->local_rows, hipMemcpyDeviceToHost));
    else
        HIP_TRY(hipMemcpy2D(host, host_pitch, dev, dev_pitch, row_bytes, c->local_rows, hipMemcpyDeviceToHost));
    return RM_OK;
}

int rm_read_pixels(rm_ctx* c, void* host_rgba, size_t pitch_bytes) {
    if (!c || !host_rgba || pitch_bytes < c->width * 4) return RM_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    if (c->local_rows == 0) return RM_OK;
    return read_rows(c, host_rgba, pitch_bytes, c->out, c->out_pitch, c->width * 4);
}

int rm_read_pixels_of_frame(rm_ctx* c, size_t frame, void* host_rgba, size_t pitch_bytes) {
    if (!c || !host_rgba || pitch_bytes < c->width * 4 || frame >= c->batch) return RM_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    if (c->local_rows == 0) return RM_OK;
    const size_t stride = c->out_frame_stride ? c->out_frame_stride : c->out_pitch * c->local_rows;
    return read_rows(c, host_rgba, pitch_bytes, c->out + frame * stride, c->out_pitch, c->width * 4);
}

int rm_read_hit_mask(rm_ctx* c, void* host_mask, size_t pitch_bytes) {
    if (!c || !host_mask || pitch_bytes < c->width) return RM_ERR_INVALID_ARG;
    if (!c->hit) return RM_ERR_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    if (c->local_rows == 0) return RM_OK;
    return read_rows(c, host_mask, pitch_bytes, c->hit, c->width, c->width);
}

int rm_device_framebuffer(rm_ctx* c, void** device_rgba, size_t* pitch_bytes) {
    if (!c || !device_rgba) return RM_ERR_INVALID_ARG;
    *device_rgba = c->out;
    if (pitch_bytes) *pitch_bytes = c->out_pitch;
    return RM_OK;
}

int rm_get_scene_stats(rm_ctx* c, rm_scene_stats* out) {
    if (!c || !out) return RM_ERR_INVALID_ARG;
    if (!c->have_scene) return RM_ERR_NOT_READY;
    out->nodes = c->n_nodes;
    out->reachable = c->scene.reachable;
    out->program_slots = static_cast<uint32_t>(c->scene.program.size());
    out->value_slots = c->scene.value_slots;
    out->tree_depth = c->scene.tree_depth;
    out->lds_bytes = static_cast<uint32_t>(c->lds_bytes);
    out->kernel_vgprs = static_cast<uint32_t>(c->jit_active ? c->jit.vgprs : c->vgprs);
    out->kernel_scratch = static_cast<uint32_t>(c->jit_active ? c->jit.scratch : c->scratch);
    out->jit = c->jit_active ? 1u : 0u;
    out->exact_divisors = c->exact_divisors;
    return RM_OK;
}

int rm_flatten_scene(const rm_node* nodes, size_t n, rm_scene_stats* stats, char* listing, size_t cap) {
    rm::SceneBuild built;
    int rc = rm::build_scene(nodes, n, built);
    if (rc != RM_OK) return rc;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->nodes = static_cast<uint32_t>(built.tree.size());
        stats->reachable = built.reachable;
        stats->program_slots = static_cast<uint32_t>(built.program.size());
        stats->value_slots = built.value_slots;
        stats->tree_depth = built.tree_depth;
        stats->lds_bytes = static_cast<uint32_t>(built.tree.size() * sizeof(rm::TreeNode) +
                                                 static_cast<size_t>(built.value_slots) * RM_BLOCK_THREADS * sizeof(float));
    }
    if (listing && cap) {
        std::string text = rm::disassemble(built.program);
        size_t k = text.size() < cap - 1 ? text.size() : cap - 1;
        std::memcpy(listing, text.data(), k);
        listing[k] = 0;
    }
    return RM_OK;
}

int rm_debug_wave_stats(rm_ctx* c, uint32_t* host_out, size_t cap_waves, size_t* n_waves) {
    if (!c || !n_waves) return RM_ERR_INVALID_ARG;
    if (!c->have_scene || !c->have_info) return RM_ERR_NOT_READY;
    const size_t waves = static_cast<size_t>(render_blocks(c, c->jit_active && c->evaluator == 0, c->nframes)) * 4;
    *n_waves = waves;
    if (!host_out) return RM_OK;
    if (cap_waves < waves) return RM_ERR_INVALID_ARG;
    if (c->local_rows == 0) return RM_OK;
    HIP_TRY(hipSetDevice(c->device));
    uint32_t* d = nullptr;
    HIP_TRY(hipMalloc(&d, waves * 32));
    hipError_t e = hipMemsetAsync(d, 0, waves * 32, c->stream);
    if (e == hipSuccess) e = static_cast<hipError_t>(launch_frame(c, d, nullptr));
    if (e == hipSuccess) e = hipMemcpyAsync(host_out, d, waves * 32, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = fetch_status(c);
    hipError_t s2 = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = s2;
    (void)hipFree(d);
    return e == hipSuccess ? judge_status(c) : static_cast<int>(e);
}

int rm_set_jit(rm_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 3) return RM_ERR_INVALID_ARG;
    c->jit_mode = mode;
    return RM_OK;
}

int rm_jit_log(rm_ctx* c, char* buf, size_t cap) {
    if (!c || !buf || !cap) return RM_ERR_INVALID_ARG;
    size_t k = c->jit_log.size() < cap - 1 ? c->jit_log.size() : cap - 1;
    std::memcpy(buf, c->jit_log.data(), k);
    buf[k] = 0;
    return RM_OK;
}

int rm_jit_check(const rm_node* nodes, size_t n, const char* arch, int want_source, char* text, size_t cap) {
    rm::SceneBuild built;
    int rc = rm::build_scene(nodes, n, built);
    if (rc != RM_OK) return rc;
    // want_source & 2: generate as if the device had proved every divisor in range (ISA inspection only)
    rm::DivisorSet assumed;
    if (want_source & 2) {
        for (float v : rm::scene_divisors(nodes, n)) {
            const float m = v < 0 ? -v : v;
            if (m >= 0x1p-20f && m <= 0x1p20f) { uint32_t u; std::memcpy(&u, &v, 4); assumed.insert(u); }
        }
        rc = rm::build_scene(nodes, n, built, &assumed);
        if (rc != RM_OK) return rc;
    }
    std::string scene;
    size_t lds = 0;
    if (!jit_source(nodes, n, built, &assumed, (want_source & 4) != 0, &scene, &lds)) return RM_ERR_TOO_LARGE;
    std::string log;
    std::vector<char> code;
    const bool ok = rm::jit_build_code(arch ? arch : "gfx950", rm::jit_full_source(scene), &code, &log);
    if (text && cap) {
        const std::string& src = (want_source & 1) ? scene : log;
        size_t k = src.size() < cap - 1 ? src.size() : cap - 1;
        std::memcpy(text, src.data(), k);
        text[k] = 0;
    }
    return ok ? RM_OK : RM_ERR_JIT;
}

int rm_set_heavy_first(rm_ctx* c, int enable) {
    if (!c) return RM_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    c->feedback = enable != 0;
    HIP_TRY(reset_schedule(c));
    return RM_OK;
}

int rm_selftest_math(rm_ctx* c, const float* divisors, size_t n, unsigned long long* sqrt_mismatches,
                     unsigned long long* div_mismatches) {
    if (!c || !sqrt_mismatches || (n && (!divisors || !div_mismatches)) || n > 64) return RM_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(c->device));
    int rc = sqrt_sweep(c, 0u, 1ull << 32, sqrt_mismatches);
    if (rc != RM_OK) return rc;
    std::vector<float> cs(divisors, divisors + n);
    std::vector<unsigned long long> bad;
    HIP_TRY(check_divisors(c, cs, 0u, 1ull << 32, bad));
    for (size_t i = 0; i < n; ++i) div_mismatches[i] = bad[i];
    return RM_OK;
}

int rm_scene_bound_of(const rm_node* nodes, size_t n, rm_scene_bound* out) {
    if (!nodes || !n || !out) return RM_ERR_INVALID_ARG;
    rm::SceneBuild built;                        // validation: child links inside the table and forward
    const int rc = rm::build_scene(nodes, n, built, nullptr, false);
    if (rc != RM_OK) return rc;
    rm::scene_bound(nodes, n, out);
    return RM_OK;
}

int rm_set_ray_windows(rm_ctx* c, int enable) {
    if (!c) return RM_ERR_INVALID_ARG;
    c->ray_windows = enable != 0;
    c->trips_per_tile = 0; c->longest_tile = 0;       // measured under the other setting
    return RM_OK;
}

int rm_set_culling(rm_ctx* c, int enable) {
    if (!c) return RM_ERR_INVALID_ARG;
    c->cull = enable != 0;
    return RM_OK;
}

int rm_set_short_math(rm_ctx* c, int enable) {
    if (!c) return RM_ERR_INVALID_ARG;
    c->short_math = enable != 0;
    return RM_OK;
}

int rm_set_boost_age(rm_ctx* c, unsigned trips) {
    if (!c) return RM_ERR_INVALID_ARG;
    c->boost_age = trips;
    return RM_OK;
}

int rm_debug_last_check(rm_ctx* c, unsigned* tiles_done, unsigned* tiles_expected, unsigned* watchdog_flag) {
    if (!c) return RM_ERR_INVALID_ARG;
    if (tiles_done) *tiles_done = c->check_done;
    if (tiles_expected) *tiles_expected = c->check_expected;
    if (watchdog_flag) *watchdog_flag = c->check_flag;
    return RM_OK;
}

int rm_debug_set_watchdog(rm_ctx* c, unsigned long long ticks) {
    if (!c) return RM_ERR_INVALID_ARG;
    c->watchdog = ticks ? ticks : 1ull << 36;
    return RM_OK;
}

int rm_set_evaluator(rm_ctx* c, int which) {
    if (!c || (which != 0 && which != 1)) return RM_ERR_INVALID_ARG;
    c->evaluator = static_cast<uint32_t>(which);
    return RM_OK;
}

const char* rm_status_string(int status) {
    switch (status) {
    case RM_OK: return "ok";
    case RM_ERR_INVALID_ARG: return "invalid argument";
    case RM_ERR_BAD_SCENE: return "bad scene: nenio/unknown node reachable or child link out of range";
    case RM_ERR_UNSUPPORTED: return "record not representable in the given bazo layout";
    case RM_ERR_IO: return "scene file could not be read or written";
    case RM_ERR_TOO_LARGE: return "scene too large (nodes, tree depth, program slots or LDS)";
    case RM_ERR_NOT_READY: return "primitives and rays_info must be set first";
    case RM_ERR_NO_KERNEL: return "no gfx950 code object for this device";
    case RM_ERR_JIT: return "run-time compilation of the scene failed (see rm_jit_log)";
    case RM_ERR_ARITHMETIC: return "this device's short sqrt sequence is not IEEE sqrtf (device_math.h)";
    case RM_ERR_INCOMPLETE: return "a frame launched since the last check did not render all its tiles (or a wavefront left on the watchdog)";
    default: return status > 0 ? hipGetErrorString(static_cast<hipError_t>(status)) : "unknown status";
    }
}

int rm_eval_points(rm_ctx* c, int which, size_t idx, const float* xyz, size_t count, float* dist_out, float* norm_out) {
    if (!c || !xyz || !dist_out || count == 0 || count > (1u << 24) || which < 0 || which > 2) return RM_ERR_INVALID_ARG;
    if (!c->have_scene) return RM_ERR_NOT_READY;
    if (idx >= c->n_nodes || (which != 1 && idx != 0)) return RM_ERR_INVALID_ARG;
    if (which == 2 && !c->jit_active) return RM_ERR_NOT_READY;
    if (!c->scene.seen[idx]) return RM_ERR_INVALID_ARG;   // not reachable from node 0: never flattened
    HIP_TRY(hipSetDevice(c->device));
    float *d_xyz = nullptr, *d_dist = nullptr, *d_norm = nullptr;
    hipError_t e = hipMalloc(&d_xyz, count * 12);
    if (e == hipSuccess) e = hipMalloc(&d_dist, count * 4);
    if (e == hipSuccess && norm_out) e = hipMalloc(&d_norm, count * 12);
    if (e == hipSuccess) e = hipMemcpyAsync(d_xyz, xyz, count * 12, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        rm::RenderParams p;
        fill_params(c, p);
        if (which == 2)
            e = rm::jit_launch_eval_points(c->jit, p, d_xyz, static_cast<uint32_t>(count), d_dist, d_norm, c->stream);
        else
            e = rm::launch_eval_points(p, which, static_cast<uint32_t>(idx), d_xyz, static_cast<uint32_t>(count),
                                       d_dist, d_norm, c->lds_bytes, c->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(dist_out, d_dist, count * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && norm_out) e = hipMemcpyAsync(norm_out, d_norm, count * 12, hipMemcpyDeviceToHost, c->stream);
    hipError_t s = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = s;
    (void)hipFree(d_xyz); (void)hipFree(d_dist); (void)hipFree(d_norm);
    return static_cast<int>(e);
}

int rm_eval_rays(rm_ctx* c, const int32_t* xy, size_t count, float* ray_out) {
    if (!c || !xy || !ray_out || count == 0 || count > (1u << 24)) return RM_ERR_INVALID_ARG;
    if (!c->have_info) return RM_ERR_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    int* d_xy = nullptr; float* d_out = nullptr;
    hipError_t e = hipMalloc(&d_xy, count * 8);
    if (e == hipSuccess) e = hipMalloc(&d_out, count * 24);
    if (e == hipSuccess) e = hipMemcpyAsync(d_xy, xy, count * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        rm::RenderParams p;
        fill_params(c, p);
        e = rm::launch_eval_rays(p, d_xy, static_cast<uint32_t>(count), d_out, c->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ray_out, d_out, count * 24, hipMemcpyDeviceToHost, c->stream);
    hipError_t s = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = s;
    (void)hipFree(d_xy); (void)hipFree(d_out);
    return static_cast<int>(e);
}

}  // extern "C"
