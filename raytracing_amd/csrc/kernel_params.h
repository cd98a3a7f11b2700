// kernel_params.h - launch parameters shared by capi.cpp and raymarch_kernels.hip.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "scene_program.h"

#define RM_BLOCK_THREADS 256     // 4 wavefronts per workgroup
#define RM_TILE_PIXELS   64      // work unit of the tile queue: one 8x8 pixel tile, one wavefront
#define RM_QUEUE_HEADS   8       // tile queue heads (one per XCD under round-robin workgroup placement)
#define RM_COST_BUCKETS  8       // tiles are sorted by cost into this many lists per head
#define RM_QUEUE_STRIDE  32      // dwords between counters: each on a 128-byte line of its own
// One scheduler set = 96 counters (lines):
//   [0..7]          pull position of head h;  [8] pull position of the overflow list;  [9] of the tail (tiles of
//                   frames the previous launch did not have, in natural order)
//   [16 + 8*h + b]  length of the list (head h, cost bucket b) written by the frame that owns the set
//   [80]            length of the overflow list written by that frame
#define RM_SET_PULL(h)     ((h) * RM_QUEUE_STRIDE)
#define RM_SET_COUNT(h, b) ((16 + 8 * (h) + (b)) * RM_QUEUE_STRIDE)
#define RM_SET_OVERFLOW    (80 * RM_QUEUE_STRIDE)
// Tiles of fewer than 24 loop trips are the bulk of most frames and too cheap to order among
// themselves: they share cost bucket RM_CHEAP_BUCKET, and both ends of the queue handle them
// RM_QUEUE_CHUNK at a time (one atomic per eight tiles: a word takes ~88 atomics per microsecond, a wavefront
// waits a microsecond or two for each, and an 8K frame of cheap tiles is 500 k pulls and 500 k filings on
// 8 + 8 words.  Four at a time was round 1's choice; with ray windows most tiles are a few hundred instructions
// long and eight measure 12 % faster on the 8K sweep, 1 % at 4K; sixteen add another per cent at 8K).
#define RM_CHEAP_BUCKET    5
#ifndef RM_QUEUE_CHUNK
#define RM_QUEUE_CHUNK     8
#endif
#ifndef RM_PRIO_HISTORY      // 1: tiles the previous launch filed as long start at a raised priority
#define RM_PRIO_HISTORY    0
#endif
#ifndef RM_PRIO_LEVELS       // how far a long tile's priority rises (1..3), a step at a time (device_render.h)
#define RM_PRIO_LEVELS     3
#endif
#ifndef RM_PRIO_STEP2
#define RM_PRIO_STEP2      3
#endif
#ifndef RM_PRIO_STEP3
#define RM_PRIO_STEP3      6
#endif
#define RM_SET_LINES       96
#define RM_SET_DWORDS      (RM_SET_LINES * RM_QUEUE_STRIDE)
// Completion accounting (rm_process / rm_synchronize report RM_ERR_INCOMPLETE from it): every wavefront adds
// the tiles it finished to one of RM_STATUS_COUNTERS running counters when it leaves (spread over that many
// words so that the end of a launch is not 8000 atomics on one), and a wavefront that leaves on the
// watchdog sets the flag word behind them.  The host knows how many tiles it has launched.
// Beside its tile count every line holds what the host sizes the next launch's grid from: the march-loop
// trips of the line's wavefronts and the trips of their longest tile, both per launch (two slots each, used
// in turn: a launch clears the pair the next launch will use), so neither wraps however many launches go
// unchecked (a launch of 2^22 tiles of 500 trips spreads 2^31 trips over 64 words).
#define RM_STATUS_COUNTERS 64
#define RM_STATUS_TRIPS    1       // dwords 1, 2 of a line: trips of launch seq (slot seq & 1)
#define RM_STATUS_LONGEST  3       // dwords 3, 4 of a line: longest tile of launch seq (slot seq & 1)
#define RM_STATUS_STRIDE   16      // dwords between counters
#define RM_STATUS_FLAG     (RM_STATUS_COUNTERS * RM_STATUS_STRIDE)
#define RM_STATUS_DWORDS   ((RM_STATUS_COUNTERS + 1) * RM_STATUS_STRIDE)

// rm_process_static (the entry point that allocates nothing) keeps its scheduler block - three counter sets,
// two list groups, a status block, laid out as a context's - in a file-scope device array of
// raymarch_kernels.hip: room for RM_STATIC_MAX_TILES tiles (8.39 Mpixels: a 3840x2160 frame has 129 600),
// lists of an eighth of that per head and bucket plus the overflow list; 9.5 MB on every GPU the library runs on.
#define RM_STATIC_MAX_TILES 131072
#define RM_STATIC_LIST_CAP  (RM_STATIC_MAX_TILES / 8 + 16)
#define RM_STATIC_DWORDS    (3 * RM_SET_DWORDS + 2 * (RM_QUEUE_HEADS * RM_COST_BUCKETS * RM_STATIC_LIST_CAP + RM_STATIC_MAX_TILES) + RM_STATUS_DWORDS)

// frames one launch may render (rm_set_batch): their cameras travel in the kernarg segment
// (64 B each; the segment holds 4 KiB)
#define RM_MAX_BATCH 32

// row bands (rm_set_band_pattern): a run of bands is at most RM_MAX_BAND_PERIOD long (include/raymarching_amd.h),
// and one context renders at most this many bands of a run
#define RM_MAX_BAND_OWN 16

namespace rm {

// frames of the per-lane tree walker's private stack (>= scene_build.h kMaxTreeDepth)
constexpr int kWalkDepth = 64;

// rays_info (Include/types.h:15-19) as the kernel wants it: Depth already
// converted to float (Source/cuda_kernels.cu:744), passed by value in the
// kernarg segment, so SetInfo needs no device copy.
struct Camera {
    float depth;
    float light[3];
    float pos[3], dir[3], wvec[3], hvec[3];
};

struct RenderParams {
    const Slot*     program;      // distance program (global memory, scalar loads)
    const TreeNode* tree;         // tree table (global memory, staged to LDS)
    uint32_t n_nodes;
    uint32_t width, height;       // whole image
    uint32_t local_rows;          // rows this context renders
    // row bands: 0 = the whole frame; else of every run of band_period bands this context renders band_own, the
    // ones at places band_slot[0..band_own) of the run, packed top to bottom (rm_set_bands: band_period = nranks,
    // band_own = 1, band_slot[0] = rank)
    uint32_t band_rows, band_period, band_own;
    // the two divisions of a tile's band lookup, prepared by the host: band_rows / 8 as a shift when it is a power of
    // two (else ~0u: divide), and ceil(2^32 / band_own), whose high product with a band number below 2^28 is the quotient
    uint32_t band_shift, band_own_magic;
    uint32_t band_slot[RM_MAX_BAND_OWN];
    uint32_t evaluator;           // 0 distance program, 1 tree walker
    uint32_t nframes;             // frames of this launch (rm_set_batch): tile ids run over all of them
    size_t frame_stride, hit_frame_stride, lum_frame_stride;     // bytes from a frame's plane to the next frame's
    uint8_t* rgba;  size_t pitch;
    uint8_t* hit;   size_t hit_pitch;
    uint8_t* lum;   size_t lum_pitch;   // optional luminance plane, 1 B/px (rm_set_luminance_output)
    uint32_t* wave_stats;         // optional diagnostics, 8 dwords per wavefront
    uint32_t* status;             // RM_STATUS_DWORDS: tiles finished (running counters) and the watchdog flag
    uint32_t stat_slot;           // which RM_STATUS_LONGEST slot this launch fills (the launch number's parity)
    // tile scheduler (device_render.h): this frame owns set_cur (all zero when it starts), reads the
    // lists the previous frame wrote (counts in set_prev), and zeroes set_next for the frame after it
    uint32_t* set_cur;
    const uint32_t* set_prev;
    uint32_t* set_next;
    const uint32_t* lists_prev;   // [8 heads][8 buckets][list_cap] then [overflow_cap]: tiles by falling cost
    uint32_t* lists_cur;          // the same for the next frame, written by this one
    uint32_t list_cap, overflow_cap;
    uint32_t ordered;             // 1: hand out lists_prev; 0: no history, natural tile order
    // ordered: the lists hold the tile ids below hist_tiles (the previous launch's frames); a launch of more
    // frames hands out the ids from there up in natural order, behind the lists (pull counter line 9 of the set)
    uint32_t hist_tiles;
    // ...unless its frame count is a multiple of the previous launch's: then every list entry stands for list_reps
    // tiles, itself and the same tile of the frames hist_tiles, 2 hist_tiles, ... further on (consecutive frames of a
    // camera path cost nearly the same tile by tile), handed out one after the other; list_reps_magic = floor(2^32 /
    // list_reps) for the division (1 and 0: entries stand for themselves)
    uint32_t list_reps, list_reps_magic;
    uint32_t feedback;            // 1: write lists_cur
    uint32_t boost_age;           // loop trips after which a tile raises its wave's issue priority
    unsigned long long watchdog;  // ticks of the 100 MHz clock (s_memrealtime) after which a wavefront gives up (rm_debug_set_watchdog)
    // ray windows (device_render.h ray_window; rm_types.h): the scene's volume, its radius grown by the margin;
    // kind RM_BOUND_NONE when the scene offers none or windows are switched off
    uint32_t bound_kind;
    float bound_c[3], bound_u[3], bound_R;
    Camera cams[RM_MAX_BATCH];    // cams[f]: frame f of the launch
};

#ifndef __HIPCC_RTC__
// rm_assemble_luminance: where every image row lies in the ranks' parts, and the planes
struct AssembleParams {
    const uint8_t* parts;  size_t rank_stride, part_frame_stride, part_pitch;
    uint8_t* rgba;         size_t pitch, rgba_frame_stride;
    uint32_t width, height;
    uint32_t band_rows, period;   // band_rows 0: one part, rows as they are
    uint32_t vector;              // every row of every plane allows 4-byte loads and 16-byte stores
    uint8_t owner[RM_MAX_BAND_PERIOD];   // of place k of a run of bands: the rank that renders it...
    uint8_t index[RM_MAX_BAND_PERIOD];   // ...and which of that rank's bands of the run it is
    uint8_t own[RM_MAX_BAND_PERIOD];     // own[r]: bands of a run that rank r renders
};

hipError_t launch_render(const RenderParams& prm, unsigned blocks, size_t lds_bytes, hipStream_t stream);
// resident workgroups per CU of the ahead-of-time render kernel with `lds_bytes` of dynamic LDS
hipError_t render_occupancy(size_t lds_bytes, int* blocks_per_cu);
hipError_t launch_eval_points(const RenderParams& prm, int which, uint32_t idx, const float* xyz, uint32_t count,
                              float* dist_out, float* norm_out, size_t lds_bytes, hipStream_t stream);
hipError_t launch_eval_rays(const RenderParams& prm, const int* xy, uint32_t count, float* out, hipStream_t stream);
// raise the kernels' dynamic-LDS ceiling when a scene needs more than the 64 KiB default
hipError_t configure_lds(size_t lds_bytes);
hipError_t kernel_attributes(int* vgprs, int* scratch_bytes);
// device address of the static scheduler block (RM_STATIC_DWORDS dwords) on `device`, which must be current
hipError_t static_scratch_address(int device, uint32_t** ptr);
// enumeration proofs of device_math.h's short sqrt / quotient sequences (mismatch counts in *bad)
// luminance planes of the ranks' band-interleaved parts -> RGBA8 frames (rm_assemble_luminance[_batch])
hipError_t launch_assemble_luminance(const AssembleParams& prm, uint32_t frames, hipStream_t stream);
hipError_t launch_verify_sqrt(uint32_t first, unsigned long long count, unsigned long long* bad, hipStream_t stream);
hipError_t launch_verify_div(const float* c, const float* rc, uint32_t n, uint32_t first, unsigned long long count,
                             unsigned long long* bad, hipStream_t stream);
#endif

}  // namespace rm
