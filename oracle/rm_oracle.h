/*
 * rm_oracle.h - CPU restatement of the reference's per-pixel raymarch path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under raytracing_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the CPU
 * baseline, never as the thing shipped.
 *
 * Parity pin: the reference ships no tests, golden images or known-answer
 * vectors.  The reference's own kernel source cannot be
 * built in this image without hand-written stand-ins for the CUDA toolkit
 * headers and for the CMake-generated Include/config.h, so no oracle/_ref
 * build exists.  This restatement is pinned instead by the outputs the
 * reference's march produced when it was run host-side during the survey and
 * which BASELINE.md section 2 records: the sha256 of the 640x480 frames of
 * chair.sce and infinity.sce, the hit-pixel counts at 640x480 and 1920x1080,
 * and the mean root-evaluation / node-call counts per pixel.  Node types that
 * occur in neither stock scene (portanta_sfero, cilindro, ebeno, *_3, *_4,
 * glata_*, rotacioX/Y/Z) are restated from the source text alone: for those,
 * parity is unpinned.
 *
 * Arithmetic conventions (all fp32, no contraction: build with
 * -ffp-contract=off):
 *   length_3(x,y,z)   = sqrtf(x*x + y*y + z*z)            (CUDA norm3df)
 *   r_length_3(x,y,z) = 1.f / sqrtf(x*x + y*y + z*z)      (CUDA rnorm3df)
 *   length_2(x,y)     = (float)sqrt((double)x*x + (double)y*y)
 *                       == glibc 2.35 hypotf bit-for-bit (CUDA hypotf)
 *   min(a,b) = (b<a)?b:a ; max(a,b) = (a<b)?b:a
 * Source/cuda_kernels.cu itself is JIT-compiled with -use_fast_math
 * (Source/cuda_rays.cu:76,88), so its CUDA output is not bit-defined; the
 * conventions above are the ones under which BASELINE.md's hashes were made.
 */
#ifndef RM_ORACLE_H
#define RM_ORACLE_H

#include "../include/rm_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RMO_NTYPES 21

typedef struct rmo_counters {
    uint64_t dist_calls[RMO_NTYPES];   /* per node type, all callers */
    uint64_t norm_calls[RMO_NTYPES];
    uint64_t root_dist_primary;        /* DIST(root) in the primary march */
    uint64_t root_dist_shadow;         /* DIST(root) in the shadow march  */
    uint64_t root_norm;                /* NORM(root) calls                */
    uint64_t hits;                     /* pixels that were shaded         */
    uint64_t lum_out_of_range;         /* quirk q7: float->u8 outside [0,255] */
    uint64_t hit_colour_disagree;      /* quirk q6: shaded pixel with RGB == 0 */
    uint64_t flops;                    /* algorithmic FLOPs, SURVEY.md appendix D */
} rmo_counters;

/* index 0..20 of a known type in declaration order, -1 for nenio/unknown */
int  rmo_type_index(uint32_t type);
const char* rmo_type_name(uint32_t type);

/* Decode `n` 48-byte bazo records (layout: enum rm_bazo_layout) into canonical
 * nodes.  Returns 0, or a negative value on an unrepresentable record. */
int  rmo_decode_bazo(const void* bytes, size_t n, int layout, rm_node* out);
/* Encode canonical nodes as `.sce` records (RM_BAZO_SCE layout). */
void rmo_encode_sce(const rm_node* nodes, size_t n, void* bytes48);

/* 0 if every node reachable from node 0 has a known type, child links stay
 * inside [0,n) and point forward; negative otherwise. */
int  rmo_validate(const rm_node* nodes, size_t n);

/* Node-level evaluation (Source/cuda_kernels.cu:51-592). c may be NULL. */
float    rmo_dist(const rm_node* nodes, size_t idx, rm_point p, rmo_counters* c);
rm_point rmo_norm(const rm_node* nodes, size_t idx, rm_point p, rmo_counters* c);

/* kernel_SetRays for one pixel (Source/cuda_kernels.cu:730-762). */
rm_ray rmo_set_ray(size_t width, size_t height, const rm_rays_info* info,
                   long long x, long long y);

/* kernel_Process over rows [y0,y1) (Source/cuda_kernels.cu:598-719).
 * rgba: width*height*4 bytes, row 0 = top, byte offset x*4 (surf2Dwrite at
 * :717).  hit (optional): width*height bytes, 1 where the pixel was shaded.
 * evals (optional): width*height uint16 root DIST evaluations per pixel.
 * Single-threaded; counters (optional) are accumulated into. */
void rmo_render_rows(const rm_node* nodes, size_t n, size_t width, size_t height,
                     const rm_rays_info* info, size_t y0, size_t y1,
                     uint8_t* rgba, uint8_t* hit, uint16_t* evals, rmo_counters* c);

/* Same over the whole frame with `threads` pthreads (rows interleaved).
 * Returns wall seconds. */
double rmo_render(const rm_node* nodes, size_t n, size_t width, size_t height,
                  const rm_rays_info* info, int threads,
                  uint8_t* rgba, uint8_t* hit, uint16_t* evals, rmo_counters* c);

/* The product's ray windows restated (NOT the reference's behaviour: include/raymarching_amd.h "Ray windows";
 * device code raytracing_amd/csrc/device_render.h ray_window).  `window` is a volume with
 * DIST(root, x) >= dist(x, centre or axis) - r; a primary ray is marched only while it can be within
 * RM_WINDOW_MARGIN of it.  With a window record (of any kind, RM_BOUND_NONE included) the march is the PRODUCT's in
 * its second departure from the reference's too: a hit whose surface faces away from the light (N_L <= 0) is shaded
 * without its shadow march, whose result :694 multiplies by zero, and a shadow march ends where the rest of it
 * provably changes nothing (rmo_shadow_window).  Two uses: tests check that the product march's
 * frame equals the reference march's frame, and the counters give the FLOPs the product's kernel executes.
 * window == NULL: the reference's march. */
int rmo_ray_window(const rm_scene_bound* window, rm_ray r, float* t_end);
void rmo_render_rows_windowed(const rm_node* nodes, size_t n, size_t width, size_t height,
                              const rm_rays_info* info, const rm_scene_bound* window, size_t y0, size_t y1,
                              uint8_t* rgba, uint8_t* hit, uint16_t* evals, rmo_counters* c);
double rmo_render_windowed(const rm_node* nodes, size_t n, size_t width, size_t height,
                           const rm_rays_info* info, const rm_scene_bound* window, int threads,
                           uint8_t* rgba, uint8_t* hit, uint16_t* evals, rmo_counters* c);

/* main.cpp:127-138 default camera; light rotated `light_steps` times by +1
 * degree about z (main.cpp:261-266); right-drag of `phi_steps` pixels in x
 * (main.cpp:248-255). */
void rmo_default_info(rm_rays_info* info, int light_steps, int phi_steps);

/* The reference's event loop (main.cpp:144-270) driven by a script: `ev` holds
 * n_events int32 triples (kind, a, b).  Every RMO_EV_FRAME emits the rays_info
 * the loop would hand to SetInfo for that frame; returns the number of frames
 * (at most `cap` are stored). */
enum {
    RMO_EV_FRAME = 0,   /* end of the polling loop: light turns unless paused, render   */
    RMO_EV_LEFT = 1,    /* SDLK_LEFT: light +1 degree                                   */
    RMO_EV_RIGHT = 2,   /* SDLK_RIGHT: light -1 degree                                  */
    RMO_EV_SPACE = 3,   /* SDLK_SPACE: camera reset (scale 2^-2.1)                      */
    RMO_EV_PAUSE = 4,   /* 'p' down                                                     */
    RMO_EV_RESUME = 5,  /* 'p' up                                                       */
    RMO_EV_WHEEL = 6,   /* a = wheel.y                                                  */
    RMO_EV_PAN = 7,     /* mouse motion (a, b) pixels, middle button down               */
    RMO_EV_DRAG = 8     /* mouse motion (a, b) pixels, right button down                */
};
size_t rmo_camera_script(const int32_t* ev, size_t n_events, rm_rays_info* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
