/*
 * rm_types.h - plain-C data model of the raymarch hot path.
 *
 * These are this project's own declarations of the PODs the reference passes
 * across its host/device boundary.  Names follow the reference's (Esperanto)
 * domain vocabulary; each item cites the reference declaration it mirrors.
 * Paths are relative to the reference checkout.
 *
 *   point / ray / rays_info      Include/types.h:8-19
 *   object_type enum values      Include/types.h:29-54
 *   payload structs              Include/objects_list.h:8-30
 *   tuning constants             Include/defines.h:80-93, Source/cuda_kernels.cu:651-652,693
 */
#ifndef RM_TYPES_H
#define RM_TYPES_H

#ifdef __HIPCC_RTC__      /* run-time compilation: no system headers */
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
#else
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Include/types.h:8 `point` */
typedef struct rm_point { float x, y, z; } rm_point;

/* Include/types.h:14 `ray` : origin p, unit direction d (24 bytes). */
typedef struct rm_ray { rm_point p, d; } rm_ray;

/* Include/types.h:15-19 `rays_info`, LP64 layout: size_t Depth @0, five
 * points @8..67, tail padding to 72 bytes. */
typedef struct rm_rays_info {
    uint64_t Depth;
    rm_point LightSource;
    rm_point StartPos, StartDir, StartWVec, StartHVec;
    uint32_t _pad;
} rm_rays_info;

/* Include/types.h:29-54 `primitives::object_type` (same numeric values). */
enum rm_object_type {
    RM_TYPE_NENIO            = 0x0000,
    RM_TYPE_PORTANTA_SFERO   = 0x0001,
    RM_TYPE_SFERO            = 0x0002,
    RM_TYPE_KUBO             = 0x0003,
    RM_TYPE_CILINDRO         = 0x0004,
    RM_TYPE_EBENO            = 0x0080,
    RM_TYPE_KUNIGAJO_2       = 0x0100,
    RM_TYPE_KUNIGAJO_3       = 0x0101,
    RM_TYPE_KUNIGAJO_4       = 0x0102,
    RM_TYPE_KOMUNAJO_2       = 0x0103,
    RM_TYPE_KOMUNAJO_3       = 0x0104,
    RM_TYPE_KOMUNAJO_4       = 0x0105,
    RM_TYPE_KOMPLEMENTO      = 0x0106,
    RM_TYPE_GLATA_KUNIGAJO_2 = 0x0107,
    RM_TYPE_GLATA_KOMUNAJO_2 = 0x0108,
    RM_TYPE_MOVO             = 0x0200,
    RM_TYPE_ROTACIOX         = 0x0201,
    RM_TYPE_ROTACIOY         = 0x0202,
    RM_TYPE_ROTACIOZ         = 0x0203,
    RM_TYPE_ROTACIOQ         = 0x0204,
    RM_TYPE_SENFINA_RIPETO   = 0x0205
};

/*
 * Canonical node: one `primitives::bazo` (Include/types.h:56-61) with its
 * 24-byte payload (Include/objects_list.h:8-30) unpacked into fixed-width
 * fields, so the record no longer depends on the width of the host's `long`
 * (`counter`, Include/types.h:5) nor on the field order of the build that
 * wrote it.  Child links `o[i]` stay relative to the node itself.
 *
 *   type              o[]        f[]
 *   portanta_sfero    o          t.x t.y t.z r
 *   sfero                        r
 *   kubo                         b.x b.y b.z
 *   cilindro                     r h
 *   ebeno                        n.x n.y n.z
 *   kunigajo_N/komunajo_N  o[0..N-1]
 *   komplemento       o
 *   glata_*_2         o[0] o[1]  k
 *   movo              o          t.x t.y t.z
 *   rotacioX/Y/Z      o          cos_phi sin_phi
 *   rotacioQ          o          q_w q.x q.y q.z
 *   senfina_ripeto    o          a.x a.y a.z
 */
typedef struct rm_node {
    uint32_t type;
    int32_t  o[4];
    float    f[5];
} rm_node;

/* Byte layouts of a 48-byte `bazo` record that rm_decode_bazo understands. */
enum rm_bazo_layout {
    /* Include/types.h:56-61 compiled LP64 (Linux): type@0, data@4 with 8-byte
     * counters, dist@32, norm@40.  kunigajo_4/komunajo_4 do not fit the
     * 24-byte payload in this layout and are rejected. */
    RM_BAZO_LP64  = 0,
    /* Same field order compiled LLP64 (the reference's own Windows x64
     * target): 4-byte counters. */
    RM_BAZO_LLP64 = 1,
    /* The on-disk `.sce` record (Scenes/NAME.sce): payload@0 with 4-byte
     * counters, dist@24, norm@32, type@40, 4 pad bytes. */
    RM_BAZO_SCE   = 2
};

#define RM_BAZO_BYTES        48   /* sizeof(primitives::bazo) in all three layouts */
#define RM_PRIMITIVE_PAYLOAD 24   /* Include/defines.h:80 */

/* Include/defines.h:89-93 and Source/cuda_kernels.cu:626,651-652,660,693 */
#define RM_MAX_DIST       10000.f
#define RM_MIN_DIST       .02f
#define RM_MAX_LUM        .9f
#define RM_MIN_LUM        .1f
#define RM_DELTA          1
#define RM_HARDNESS       128.f
#define RM_PRIMARY_STEPS  500
#define RM_SHADOW_STEPS   300
#define RM_MATERIAL_R     .5f
#define RM_MATERIAL_G     .7f
#define RM_MATERIAL_B     1.f

/* Ray windows (raymarching_amd.h; no counterpart in the reference): a volume with
 *     DIST(root, x) >= distance(x, centre or axis) - r     at every point x. */
enum rm_bound_kind { RM_BOUND_NONE = 0, RM_BOUND_SPHERE = 1, RM_BOUND_CYLINDER = 2 };
typedef struct rm_scene_bound {
    uint32_t kind;             /* enum rm_bound_kind                               */
    float c[3];                /* centre of the sphere / a point of the axis       */
    float u[3];                /* unit direction of the cylinder's axis            */
    float r;                   /* radius                                           */
} rm_scene_bound;
/* A primary ray is marched while it can be within RM_WINDOW_MARGIN of the volume.  The
 * margin covers what separates the march from the ideal ray: positions and the
 * travelled distance are running float sums (up to RM_PRIMARY_STEPS steps of coordinates
 * up to RM_WINDOW_MAX_COORD + RM_MAX_DIST: 3.1 units at worst), DIST itself is rounded
 * (well under a unit at such coordinates), and a surface is RM_MIN_DIST thick. */
#define RM_WINDOW_MARGIN     8.f
#define RM_WINDOW_SLACK      1.f        /* added to the distance at which the ray leaves the volume */
#define RM_WINDOW_MAX_COORD  16384.f    /* rays starting farther out, and volumes centred or reaching farther out, get no window */
#define RM_WINDOW_MIN_SIN2   .01f       /* rays within 5.7 degrees of a cylinder's axis get no window */

/* Multi-GPU row bands (rm_set_band_pattern): bands are dealt to ranks in runs of at most this many bands, to at
 * most this many ranks */
#define RM_MAX_BAND_PERIOD   64

#ifdef __cplusplus
}
#endif
#endif /* RM_TYPES_H */
