/*
 * raymarching_amd.h - C ABI of the MI355X raymarch backend.
 *
 * This is the whole host<->device boundary of the per-pixel hot path.  It takes
 * the place of the reference's CUDA driver glue (Include/cuda_rays.cuh:14-46,
 * Include/cuda_defines.cuh, Source/cuda_rays.cu) behind the reference's abstract
 * backend interface `null::raymarching` (Include/rays.h:15-69).  Each entry
 * point cites the reference interface it replaces; paths are relative to the
 * reference checkout.  Plain pointers and sizes only.
 *
 * Conventions (Source/cuda_rays.cu:3-5): every call returns 0 on success.  A
 * positive value is a hipError_t, a negative value one of enum rm_status.
 * Unlike the reference, an error is never latched and silently carried on:
 * every call that waits for the device (rm_process, rm_synchronize, rm_read_*)
 * also checks that every tile of every frame launched since the last check was
 * rendered, and returns RM_ERR_INCOMPLETE otherwise (the kernel is a set of
 * persistent wavefronts pulling tiles from a queue; a wavefront that its watchdog
 * stopped, or a tile that went missing, must not pass for a finished frame).
 * All calls are synchronous on the context's stream unless stated otherwise.
 * A context is not thread-safe; use one per host thread / per GPU.
 *
 * Call protocol (Source/main.cpp:67-138,269-270,292-293):
 *   rm_create -> [rm_set_output] -> rm_set_primitives|rm_set_nodes -> rm_set_info
 *   -> per frame { rm_set_info, rm_process } -> rm_read_pixels -> rm_destroy
 */
#ifndef RAYMARCHING_AMD_H
#define RAYMARCHING_AMD_H

#include "rm_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rm_ctx rm_ctx;

enum rm_status {
    RM_OK              =  0,
    RM_ERR_INVALID_ARG = -1,  /* null pointer, zero size, bad enum value            */
    RM_ERR_BAD_SCENE   = -2,  /* unknown/nenio node reachable, link out of range     */
    RM_ERR_UNSUPPORTED = -3,  /* record not representable in the given layout        */
    RM_ERR_IO          = -4,  /* file could not be read / written / wrong size       */
    RM_ERR_TOO_LARGE   = -5,  /* scene exceeds RM_MAX_NODES or the LDS budget        */
    RM_ERR_NOT_READY   = -6,  /* rm_process before primitives and info were set      */
    RM_ERR_NO_KERNEL   = -7,  /* the HIP code object for gfx950 could not be loaded  */
    RM_ERR_JIT         = -8,  /* run-time compilation of the scene failed            */
    RM_ERR_ARITHMETIC  = -9,  /* the device's short sqrt sequence is not IEEE sqrtf  */
    RM_ERR_INCOMPLETE  = -10  /* a launched frame did not render all its tiles       */
};

#define RM_MAX_NODES 2048     /* reference: 256 (Source/cuda_kernels.cu:607) */

/* ---- scene tables ------------------------------------------------------- */

/* Decode `n` 48-byte `primitives::bazo` records (Include/types.h:56-61) given
 * in `layout` (enum rm_bazo_layout) into canonical nodes.  The dist/norm
 * function-pointer fields are ignored, as kernel_SetPrimitives
 * (Source/cuda_kernels.cu:721-728) would overwrite them. */
int rm_decode_bazo(const void* records, size_t n, int layout, rm_node* out);

/* Read a `.sce` scene table (Scenes/NAME.sce; the file named by `--input`,
 * Source/main.cpp:29-31).  Writes at most `cap` nodes; *n receives the number
 * of records in the file. */
int rm_load_sce(const char* path, rm_node* out, size_t cap, size_t* n);

/* Write canonical nodes as a `.sce` table (4-byte counters). */
int rm_save_sce(const char* path, const rm_node* nodes, size_t n);

/* ---- context ------------------------------------------------------------ */

/* Init (Include/rays.h:29; Source/cuda_rays.cu:17-199): bind `device`, create
 * the stream and timing events, allocate the width x height RGBA8 framebuffer.
 * No ray buffer is allocated: ray generation is fused into the march kernel. */
int rm_create(int device, size_t width, size_t height, rm_ctx** out);

/* UnsetTexture + Quit (Include/rays.h:41,47; Source/cuda_rays.cu:218-227,256-264). */
int rm_destroy(rm_ctx* ctx);

/* Launch on a caller-owned hipStream_t instead of the context's own stream
 * (replaces the fixed `_default_stream`, Include/cuda_defines.cuh:9-10).
 * NULL selects the null stream. */
int rm_set_stream(rm_ctx* ctx, void* hip_stream);

/* SetTexture (Include/rays.h:46; Source/cuda_rays.cu:241-254): render into a
 * caller-owned DEVICE buffer of RGBA8 rows `pitch_bytes` apart instead of the
 * context's framebuffer.  Row 0 is the top row and pixel x sits at byte x*4,
 * as surf2Dwrite does at Source/cuda_kernels.cu:717.  NULL restores the
 * context's own framebuffer. */
int rm_set_output(rm_ctx* ctx, void* device_rgba, size_t pitch_bytes);

/* AddPrimitive... + SetPrimitives (Include/rays.h:49,57-64; Source/cuda_rays.cu:266-287):
 * copy the host table, validate it, flatten it into the device scene program
 * and upload.  kernel_SetPrimitives' function-pointer binding has no
 * counterpart: node types are resolved while flattening. */
int rm_set_primitives(rm_ctx* ctx, const void* bazo_records, size_t n, int layout);
int rm_set_nodes(rm_ctx* ctx, const rm_node* nodes, size_t n);

/* SetInfo (Include/rays.h:44; Source/cuda_rays.cu:230-239).  The struct is
 * copied before the call returns.  kernel_SetRays (Source/cuda_kernels.cu:730-762)
 * is not launched: rm_process derives each pixel's ray from this block. */
int rm_set_info(rm_ctx* ctx, const rm_rays_info* info);

/* Several frames per launch (no counterpart: the reference renders one frame per
 * Process()).  rm_set_batch(ctx, n, ...) makes room for launches of up to n frames
 * (1..RM_MAX_BATCH): every rm_process / rm_process_async then renders its frames in
 * ONE kernel launch, the tiles of all of them in one pool for the persistent
 * wavefronts, so the long tiles of one frame run beside the short ones of the
 * others.  A rank's share of a frame on 8 GPUs is too little work to fill the chip
 * for as long as its longest ray takes; a batch is.
 * How many frames the next launch holds is set with the cameras: rm_set_info gives
 * all n frames the same block; rm_set_info_batch(ctx, infos, m) gives frame f the
 * f-th block and makes the launch m <= n frames long (a last launch that is not
 * full costs no reallocation, no wait and keeps the tile order learnt so far; a
 * launch of more frames than the one before it takes that one's order for as many
 * frames - for every group of them when it is a whole multiple, which is why a
 * camera path is best started with a launch of one frame - and the rest in natural order).
 * Frame f writes its planes `*_frame_stride_bytes` after frame f-1's: RGBA8 output,
 * luminance plane; 0 = packed (pitch x rows of this context).  With the context's
 * own framebuffer rm_read_pixels returns frame 0.  rm_set_batch waits for the
 * stream; it allocates only when n grows beyond anything set before. */
#define RM_MAX_BATCH 32
int rm_set_batch(rm_ctx* ctx, size_t frames, size_t rgba_frame_stride_bytes, size_t luminance_frame_stride_bytes);
int rm_set_info_batch(rm_ctx* ctx, const rm_rays_info* infos, size_t n);

/* Multi-GPU row bands (no counterpart: the reference is single-device,
 * Source/cuda_rays.cu:20-21).  The image is cut into bands of `band_rows` rows
 * (a multiple of 8, the tile height); this context renders bands b with b % nranks == rank,
 * packed top to bottom into its output buffer.  rm_local_rows reports how many
 * rows that is.  band_rows = 0 restores whole-frame rendering.
 * rm_set_band_pattern deals the bands unevenly: owner[k] (k < period <=
 * RM_MAX_BAND_PERIOD) is the rank that renders bands b with b % period == k, so a
 * rank that has other work - rank 0 receives every rank's rows and rebuilds the
 * frames - can be given fewer bands than the others (none at all is allowed; at
 * most 16 of a period).  rm_set_bands(ctx, rows, rank, nranks) is the pattern
 * owner[k] = k of period nranks (nranks <= RM_MAX_BAND_PERIOD). */
int rm_set_bands(rm_ctx* ctx, size_t band_rows, int rank, int nranks);
int rm_set_band_pattern(rm_ctx* ctx, size_t band_rows, const unsigned char* owner, size_t period, int rank, int nranks);
int rm_local_rows(rm_ctx* ctx, size_t* rows);

/* Optional luminance plane (no counterpart in the reference): besides the RGBA8
 * frame the kernel writes, into a caller-owned DEVICE buffer, one byte per pixel:
 * the 8-bit luminance L of Source/cuda_kernels.cu:694, 0 where nothing was hit.
 * A pixel is a function of L alone (:695-699 scale it by the material constants;
 * a miss is the black of L = 0), so a multi-GPU gather can move this plane, a
 * quarter of the bytes, and rebuild the identical RGBA8 frame on the receiving
 * device.  Rows follow the context's band layout like the RGBA8 output.  NULL
 * switches the plane off. */
int rm_set_luminance_output(rm_ctx* ctx, void* device_l8, size_t pitch_bytes);

/* Rebuilds the whole RGBA8 frame from the luminance planes of all ranks: part r
 * (rank r's plane, rows packed as rm_set_bands / rm_set_band_pattern lay them
 * out, `part_pitch_bytes` apart) starts at device_parts + r * rank_stride_bytes.
 * Uses this context's image size and band layout (band_rows = 0: one part, a
 * plain expansion).  Runs on `hip_stream` (a hipStream_t; NULL = the null
 * stream), asynchronously.
 * rm_assemble_luminance_batch rebuilds the `frames` frames of a launch in one
 * kernel: frame f's plane of rank r starts part_frame_stride_bytes * f into
 * part r, and its RGBA8 frame rgba_frame_stride_bytes * f into device_rgba (the
 * frames must not overlap).  Rows that allow it (pitches, strides and bases
 * that are multiples of 4 bytes on the luminance side and of 16 on the RGBA8
 * side) are moved as 4-byte loads and 16-byte stores. */
int rm_assemble_luminance(rm_ctx* ctx, const void* device_parts, size_t rank_stride_bytes, size_t part_pitch_bytes,
                          void* device_rgba, size_t pitch_bytes, void* hip_stream);
int rm_assemble_luminance_batch(rm_ctx* ctx, const void* device_parts, size_t rank_stride_bytes, size_t part_frame_stride_bytes,
                                size_t part_pitch_bytes, size_t frames, void* device_rgba, size_t pitch_bytes,
                                size_t rgba_frame_stride_bytes, void* hip_stream);

/* Optional explicit hit mask: one byte per
 * pixel, 1 where the pixel was shaded.  enable != 0 allocates it, zeroed. */
int rm_enable_hit_mask(rm_ctx* ctx, int enable);

/* Process (Include/rays.h:35; Source/cuda_rays.cu:201-216): render one frame
 * and wait for it.  *elapsed_ms (optional) receives the kernel time between two
 * stream events as a float; the reference returns it truncated to whole ms. */
int rm_process(rm_ctx* ctx, float* elapsed_ms);

/* rm_process for callers that cannot allocate.  It has no counterpart in the
 * reference: it is for tight serving loops, harnesses that forbid hipMalloc and
 * code that renders between two graph launches without touching the allocator.
 * (It is NOT itself capturable into a HIP graph: like rm_process it records
 * events, copies its completion counters to the host and waits for the stream,
 * none of which a capturing stream allows.)  One frame of the context's scene, camera
 * `info`, into the caller's DEVICE buffer (RGBA8 rows `pitch_bytes` apart, as
 * rm_set_output), waited for like rm_process, same pixels.  The call itself
 * allocates nothing: the tile scheduler's scratch (counter sets, tile lists,
 * completion counters), which a context otherwise gets from hipMalloc, lives in a
 * file-scope device array of the library, 9.5 MB on every GPU it runs on, whose
 * device address rm_create looks up once per device.  Limits: one frame per call,
 * at most 131 072 tiles of 8x8 pixels (8.39 Mpixels of this context's rows: a
 * 3840x2160 frame fits, 7680x4320 does not -> RM_ERR_TOO_LARGE); no hit mask, no
 * luminance plane.  All calls on one GPU share that GPU's array (the tile order
 * learnt by one call serves the next one of the same image size): calls of one
 * process on one GPU take their turn under that GPU's lock, held until the frame
 * is in (calls on different GPUs run side by side); anything else that overlaps
 * on one GPU - another process, the same array - races. */
int rm_process_static(rm_ctx* ctx, const rm_rays_info* info, void* device_rgba, size_t pitch_bytes, float* elapsed_ms);

/* Same launch without events and without waiting: for back-to-back frames
 * timed by the caller on the context's stream. */
int rm_process_async(rm_ctx* ctx);
int rm_synchronize(rm_ctx* ctx);

/* Copy the local rows of the framebuffer (or hit mask) to host memory, waiting
 * for the frame first.  pitch_bytes >= width*4 (width for the mask).  Rows with
 * no padding on either side travel as one linear copy: 55 GB/s into pinned
 * host memory, 17 GB/s into pageable memory on the MI355X box. */
int rm_read_pixels(rm_ctx* ctx, void* host_rgba, size_t pitch_bytes);
int rm_read_hit_mask(rm_ctx* ctx, void* host_mask, size_t pitch_bytes);
/* rm_read_pixels for frame `frame` (< the rm_set_batch count) of the last launch: the plane that frame's camera
 * block rendered into (no counterpart: the reference has one frame per Process(), Source/cuda_rays.cu:201-216). */
int rm_read_pixels_of_frame(rm_ctx* ctx, size_t frame, void* host_rgba, size_t pitch_bytes);

/* Device address and pitch of the buffer rm_process writes. */
int rm_device_framebuffer(rm_ctx* ctx, void** device_rgba, size_t* pitch_bytes);

/* ---- introspection ------------------------------------------------------ */

typedef struct rm_scene_stats {
    uint32_t nodes;            /* records in the node table                    */
    uint32_t reachable;        /* nodes reachable from node 0                  */
    uint32_t program_slots;    /* 32-byte slots of the wave-uniform program    */
    uint32_t value_slots;      /* per-lane LDS dwords used by the program      */
    uint32_t tree_depth;       /* longest root-to-leaf path                    */
    uint32_t lds_bytes;        /* dynamic LDS per 256-thread workgroup         */
    uint32_t kernel_vgprs;     /* register footprint reported by the runtime   */
    uint32_t kernel_scratch;   /* private (scratch) bytes per lane             */
    uint32_t jit;              /* 1: the scene runs on per-scene generated code */
    uint32_t exact_divisors;   /* divisor constants proved for the short quotient */
} rm_scene_stats;
int rm_get_scene_stats(rm_ctx* ctx, rm_scene_stats* out);

/* Ray windows (no counterpart in the reference, which marches every ray to RM_MAX_DIST,
 * Source/cuda_kernels.cu:707-714).  From the node tree the library derives a volume - a
 * sphere, or a cylinder around the axis of a senfina_ripeto - with
 *     DIST(root, x) >= distance(x, centre or axis) - r      at every point x,
 * so a primary ray is marched only over the part of it that can come near that volume:
 * a ray that never does is the black pixel of a miss at once, one that has left it
 * for good stops there.  Pixels are the reference's, byte for byte (outside the
 * volume DIST is hundreds of times RM_MIN_DIST, so no step there finds a surface);
 * on chair.sce at 3840x2160 half the march-loop trips of a frame are such steps.
 * kind RM_BOUND_NONE: the tree offers no volume (a komplemento or an ebeno at the
 * top, lattices of lattices) and every ray is marched to the end.
 * rm_scene_bound and the window's constants: rm_types.h. */
/* Host-only: the volume of a node table. */
int rm_scene_bound_of(const rm_node* nodes, size_t n, rm_scene_bound* out);
/* enable = 0: every primary ray is marched to RM_MAX_DIST as the reference marches it
 * (diagnostic; identical results).  Takes effect at the next launch. */
int rm_set_ray_windows(rm_ctx* ctx, int enable);

/* Host-only: validate and flatten a node table without touching a device.
 * stats (optional) receives the host-side fields; listing (optional) receives a
 * NUL-terminated text listing of the distance program, truncated to cap. */
int rm_flatten_scene(const rm_node* nodes, size_t n, rm_scene_stats* stats, char* listing, size_t cap);

/* Per-scene code generation (the counterpart of the reference's NVRTC step,
 * Source/cuda_rays.cu:59-153: there the fixed kernel source is compiled at
 * start-up; here the SCENE is compiled, each node becoming straight-line code).
 * mode 0: never, every scene runs on the generic distance program;
 * mode 1 (default): scenes are compiled with hiprtc when rm_set_primitives /
 *         rm_set_nodes is called: up to 64 reachable nodes as one distance and one
 *         normal function per node; larger ones (up to 4096 program slots, 64 KiB
 *         of tree table) as their flattened distance program, normals staying on
 *         the tree walker.  Falls back to the generic program if the scene is
 *         larger still or compilation fails (rm_get_scene_stats().jit tells);
 * mode 2: as 1, but a scene that cannot be compiled is an error;
 * mode 3: as 2, and small scenes take the large-scene form too (tests).
 * Takes effect at the next scene upload.  Environment RM_JIT=<mode> sets the
 * default; RM_JIT_CACHE=<dir> keeps code objects on disk between processes. */
int rm_set_jit(rm_ctx* ctx, int mode);
int rm_jit_log(rm_ctx* ctx, char* buf, size_t cap);
/* Host-only: generate and compile the code of a scene for `arch` without a
 * device.  text receives the generated source (want_source & 1) or the log; want_source & 2 generates
 * as if every divisor constant had been proved on a device, want_source & 4 in the
 * large-scene form whatever the size (inspection only). */
int rm_jit_check(const rm_node* nodes, size_t n, const char* arch, int want_source, char* text, size_t cap);

/* Arithmetic self-test (no counterpart in the reference).  The kernels take
 * sqrt and the divisions by scene constants through sequences a third as long
 * as the compiler's IEEE expansions (csrc/device_math.h: sqrt_exact, div_const)
 * and must return the same bits.  This runs them against IEEE sqrtf / division
 * on the device for all 2^32 float bit patterns (sqrt: the argument; division:
 * the numerator, once per divisor, quotient by RN(1/divisor)) and returns the
 * mismatch counts: all must be 0.  n <= 64.  Takes about a second. */
int rm_selftest_math(rm_ctx* ctx, const float* divisors, size_t n, unsigned long long* sqrt_mismatches,
                     unsigned long long* div_mismatches);

/* enable = 0: scenes uploaded from now on are flattened without bounding-sphere
 * tests in front of the sub-trees of unions (diagnostic; identical results). */
int rm_set_culling(rm_ctx* ctx, int enable);

/* enable = 0: scenes uploaded from now on divide by their constants with IEEE
 * division only (diagnostic; identical results). */
int rm_set_short_math(rm_ctx* ctx, int enable);

/* Tuning: frame-to-frame scheduling feedback (on by default).  Every frame files
 * its 8x8 tiles by cost (march-loop trips); the next frame renders them from the
 * costliest down, because the longest tiles bound the frame time.  Only the
 * order of the work depends on the previous frame, never a pixel value.
 * enable = 0: always natural tile order.  Forgets the current history. */
int rm_set_heavy_first(rm_ctx* ctx, int enable);

/* Tuning: a wavefront whose tile has run for more than `trips` march-loop trips
 * raises its issue priority (long tiles bound the frame time).  0xffffffff
 * disables it.  Results do not depend on it. */
int rm_set_boost_age(rm_ctx* ctx, unsigned trips);

/* Which evaluator computes root distances: 0 = wave-uniform scene program
 * (default), 1 = per-lane tree walker (diagnostic; identical results). */
int rm_set_evaluator(rm_ctx* ctx, int which);

/* Diagnostics: render one launch and return, per (persistent) wavefront, 8
 * dwords: march loop trips, normal passes, clock ticks in the kernel, ticks in
 * normal passes, the low words of the 100 MHz clock all XCDs share (s_memrealtime)
 * when the wavefront started and when it left, tiles rendered, trips of its longest tile.
 * With host_out NULL only *n_waves is set. */
int rm_debug_wave_stats(rm_ctx* ctx, uint32_t* host_out, size_t cap_waves, size_t* n_waves);

/* Diagnostics: what the last completion check of the context compared (rm_process,
 * rm_synchronize, rm_read_*, rm_set_stream): tiles the wavefronts reported finished
 * and tiles launched, both modulo 2^32 since the context was made, and the flag a
 * wavefront sets when its watchdog stops it.  After RM_ERR_INCOMPLETE it tells a
 * missing tile (counts differ) from a stopped wavefront (flag set). */
int rm_debug_last_check(rm_ctx* ctx, unsigned* tiles_done, unsigned* tiles_expected, unsigned* watchdog_flag);

/* Diagnostics: the persistent wavefronts' watchdog, in ticks of the 100 MHz clock all
 * XCDs share (s_memrealtime; 0 restores the default, 2^36: eleven minutes).  A wavefront
 * that has been in the kernel for longer gives up
 * at its next look at the clock (every 1024 march-loop trips, and where a tile's
 * priority steps up: rm_set_boost_age) and the launch is reported as
 * RM_ERR_INCOMPLETE.  Tests lower it to see that error path work. */
int rm_debug_set_watchdog(rm_ctx* ctx, unsigned long long ticks);

const char* rm_status_string(int status);

/* ---- node-level evaluation on the device (known-answer tests) ----------- */

/* Evaluate DIST/NORM of node `idx` of the context's scene at `count` points
 * (host arrays of xyz triples) with the device evaluators.  which: 0 = program
 * (idx must be 0), 1 = tree walker, 2 = generated code (idx must be 0).
 * norm_out may be NULL. */
int rm_eval_points(rm_ctx* ctx, int which, size_t idx, const float* xyz, size_t count,
                   float* dist_out, float* norm_out);

/* kernel_SetRays known-answer hook: the rays the march kernel derives for
 * `count` pixels (xy pairs), as 6 floats p.xyz d.xyz each. */
int rm_eval_rays(rm_ctx* ctx, const int32_t* xy, size_t count, float* ray_out);

#ifdef __cplusplus
}
#endif
#endif /* RAYMARCHING_AMD_H */
