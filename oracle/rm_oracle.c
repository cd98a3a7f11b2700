/*
 * rm_oracle.c - CPU restatement of the reference's per-pixel raymarch path.
 * TEST INFRASTRUCTURE ONLY; see rm_oracle.h for scope, pinning and the
 * arithmetic conventions.  Build: gcc -O2 -ffp-contract=off (oracle/Makefile).
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference checkout).  The reference walks the node table through per-node
 * function pointers (Include/defines.h:75-76); here the dispatch is a switch
 * on the node type and the recursion is ordinary C recursion.
 */
#define _GNU_SOURCE
#include "rm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ helpers */

static const uint32_t k_types[RMO_NTYPES] = {
    RM_TYPE_PORTANTA_SFERO, RM_TYPE_SFERO, RM_TYPE_KUBO, RM_TYPE_CILINDRO,
    RM_TYPE_EBENO,
    RM_TYPE_KUNIGAJO_2, RM_TYPE_KUNIGAJO_3, RM_TYPE_KUNIGAJO_4,
    RM_TYPE_KOMUNAJO_2, RM_TYPE_KOMUNAJO_3, RM_TYPE_KOMUNAJO_4,
    RM_TYPE_KOMPLEMENTO, RM_TYPE_GLATA_KUNIGAJO_2, RM_TYPE_GLATA_KOMUNAJO_2,
    RM_TYPE_MOVO, RM_TYPE_ROTACIOX, RM_TYPE_ROTACIOY, RM_TYPE_ROTACIOZ,
    RM_TYPE_ROTACIOQ, RM_TYPE_SENFINA_RIPETO
    /* 20 known types; slot 20 is unused padding so the arrays stay 21 wide
     * like the reference's switch (Include/defines.h:47-73, which lists 20
     * cases plus `default`). */
};
static const char* k_names[RMO_NTYPES] = {
    "portanta_sfero", "sfero", "kubo", "cilindro", "ebeno",
    "kunigajo_2", "kunigajo_3", "kunigajo_4",
    "komunajo_2", "komunajo_3", "komunajo_4",
    "komplemento", "glata_kunigajo_2", "glata_komunajo_2",
    "movo", "rotacioX", "rotacioY", "rotacioZ", "rotacioQ", "senfina_ripeto",
    "nenio"
};

int rmo_type_index(uint32_t type) {
    for (int i = 0; i < 20; ++i) if (k_types[i] == type) return i;
    return -1;
}
const char* rmo_type_name(uint32_t type) {
    int i = rmo_type_index(type);
    return k_names[i < 0 ? 20 : i];
}

/* number of child links of a node type */
static int arity(uint32_t type) {
    switch (type) {
    case RM_TYPE_KUNIGAJO_2: case RM_TYPE_KOMUNAJO_2:
    case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2: return 2;
    case RM_TYPE_KUNIGAJO_3: case RM_TYPE_KOMUNAJO_3: return 3;
    case RM_TYPE_KUNIGAJO_4: case RM_TYPE_KOMUNAJO_4: return 4;
    case RM_TYPE_PORTANTA_SFERO: case RM_TYPE_KOMPLEMENTO: case RM_TYPE_MOVO:
    case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY: case RM_TYPE_ROTACIOZ:
    case RM_TYPE_ROTACIOQ: case RM_TYPE_SENFINA_RIPETO: return 1;
    default: return 0;
    }
}

/* Include/cuda_defines.cuh:49-53 with the conventions of rm_oracle.h */
static inline float length_3(float x, float y, float z)   { return sqrtf(x * x + y * y + z * z); }
static inline float r_length_3(float x, float y, float z) { return 1.f / sqrtf(x * x + y * y + z * z); }
static inline float length_2(float x, float y) {
    return (float)sqrt((double)x * (double)x + (double)y * (double)y);
}
static inline float fmin2(float a, float b) { return (b < a) ? b : a; }
static inline float fmax2(float a, float b) { return (a < b) ? b : a; }

/* Source/cuda_kernels.cu:38-44 atomic::dot / atomic::mix */
static inline float dot3(rm_point a, rm_point b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float mixf(float a, float b, float x) { return a + x * (b - a); }

#define CNT(field, k) do { if (c) c->field += (k); } while (0)
#define CNT_DIST(type)  do { if (c) c->dist_calls[rmo_type_index(type)]++; } while (0)
#define CNT_NORM(type)  do { if (c) c->norm_calls[rmo_type_index(type)]++; } while (0)

/* ------------------------------------------------------------ record codecs */

static int32_t rd_counter(const uint8_t* p, int width) {
    if (width == 4) { int32_t v; memcpy(&v, p, 4); return v; }
    int64_t v; memcpy(&v, p, 8); return (int32_t)v;
}
static float rd_f32(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }

/* Payload structs of Include/objects_list.h:8-30 laid out with `cw`-byte
 * counters.  Floats are 4-aligned, so after k counters they start at k*cw. */
static int nfloats(uint32_t type) {
    switch (type) {
    case RM_TYPE_PORTANTA_SFERO: case RM_TYPE_ROTACIOQ: return 4;
    case RM_TYPE_KUBO: case RM_TYPE_EBENO: case RM_TYPE_MOVO: case RM_TYPE_SENFINA_RIPETO: return 3;
    case RM_TYPE_CILINDRO: case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY: case RM_TYPE_ROTACIOZ: return 2;
    case RM_TYPE_SFERO: case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2: return 1;
    default: return 0;
    }
}

static int decode_payload(uint32_t type, const uint8_t* d, int cw, rm_node* out) {
    int na = arity(type), nf = nfloats(type);
    if (na * cw + nf * 4 > RM_PRIMITIVE_PAYLOAD) return -1;   /* e.g. *_4 with 8-byte counters */
    for (int i = 0; i < na; ++i) out->o[i] = rd_counter(d + i * cw, cw);
    for (int i = 0; i < nf; ++i) out->f[i] = rd_f32(d + na * cw + i * 4);
    return 0;
}

int rmo_decode_bazo(const void* bytes, size_t n, int layout, rm_node* out) {
    const uint8_t* b = (const uint8_t*)bytes;
    for (size_t i = 0; i < n; ++i, b += RM_BAZO_BYTES) {
        uint32_t type; const uint8_t* data; int cw;
        if (layout == RM_BAZO_SCE)        { memcpy(&type, b + 40, 4); data = b;     cw = 4; }
        else if (layout == RM_BAZO_LLP64) { memcpy(&type, b, 4);      data = b + 4; cw = 4; }
        else if (layout == RM_BAZO_LP64)  { memcpy(&type, b, 4);      data = b + 4; cw = 8; }
        else return -2;
        memset(&out[i], 0, sizeof(rm_node));
        out[i].type = type;
        if (decode_payload(type, data, cw, &out[i]) != 0) return -1;
    }
    return 0;
}

void rmo_encode_sce(const rm_node* nodes, size_t n, void* bytes48) {
    uint8_t* b = (uint8_t*)bytes48;
    memset(b, 0, n * RM_BAZO_BYTES);
    for (size_t i = 0; i < n; ++i, b += RM_BAZO_BYTES) {
        int na = arity(nodes[i].type), nf = nfloats(nodes[i].type);
        memcpy(b, nodes[i].o, (size_t)na * 4);
        memcpy(b + na * 4, nodes[i].f, (size_t)nf * 4);
        memcpy(b + 40, &nodes[i].type, 4);
    }
}

static int validate_rec(const rm_node* nodes, size_t n, size_t idx, size_t depth) {
    if (idx >= n || depth > n) return -1;
    const rm_node* nd = &nodes[idx];
    if (rmo_type_index(nd->type) < 0) return -2;
    int na = arity(nd->type);
    for (int i = 0; i < na; ++i) {
        if (nd->o[i] <= 0) return -3;                 /* forward links only */
        int r = validate_rec(nodes, n, idx + (size_t)nd->o[i], depth + 1);
        if (r) return r;
    }
    return 0;
}
int rmo_validate(const rm_node* nodes, size_t n) {
    if (!nodes || n == 0) return -1;
    return validate_rec(nodes, n, 0, 0);
}

/* --------------------------------------------------- node library: distance */

/* rotacioQ's Q matrix, Source/cuda_kernels.cu:504-525 (identical at :538-559) */
typedef struct { float xx, xy, xz, yx, yy, yz, zx, zy, zz; } qmat;
static qmat build_q(const rm_node* nd) {
    float qw = nd->f[0], qx = nd->f[1], qy = nd->f[2], qz = nd->f[3];
    qmat Q; float temp;
    Q.xx = qx * qx; Q.yy = qy * qy; Q.zz = qz * qz;
    temp = Q.xx + Q.yy + Q.zz;
    Q.xx -= temp; Q.yy -= temp; Q.zz -= temp;
    Q.xy = qx * qy; temp = qz * qw; Q.yx = Q.xy + temp; Q.xy -= temp;
    Q.yz = qy * qz; temp = qx * qw; Q.zy = Q.yz + temp; Q.yz -= temp;
    Q.zx = qz * qx; temp = qy * qw; Q.xz = Q.zx + temp; Q.zx -= temp;
    return Q;
}

/* point transforms shared by *_dist and *_norm of the six transform types */
static rm_point xf_movo(const rm_node* nd, rm_point p) {          /* :409-413 */
    rm_point P = { p.x - nd->f[0], p.y - nd->f[1], p.z - nd->f[2] };
    return P;
}
static rm_point xf_rotx(const rm_node* nd, rm_point p) {          /* :430-432 */
    float c = nd->f[0], s = nd->f[1]; rm_point P;
    P.y = c * p.y + s * p.z; P.z = -s * p.y + c * p.z; P.x = p.x; return P;
}
static rm_point xf_roty(const rm_node* nd, rm_point p) {          /* :454-456 */
    float c = nd->f[0], s = nd->f[1]; rm_point P;
    P.z = c * p.z + s * p.x; P.x = -s * p.z + c * p.x; P.y = p.y; return P;
}
static rm_point xf_rotz(const rm_node* nd, rm_point p) {          /* :478-480 */
    float c = nd->f[0], s = nd->f[1]; rm_point P;
    P.x = c * p.x + s * p.y; P.y = -s * p.x + c * p.y; P.z = p.z; return P;
}
static rm_point xf_rotq(const qmat* Q, rm_point p) {              /* :527-530 */
    rm_point P = p;
    P.x += 2.f * (Q->xx * p.x + Q->xy * p.y + Q->xz * p.z);
    P.y += 2.f * (Q->yx * p.x + Q->yy * p.y + Q->yz * p.z);
    P.z += 2.f * (Q->zx * p.x + Q->zy * p.y + Q->zz * p.z);
    return P;
}
static rm_point xf_ripeto(const rm_node* nd, rm_point p) {        /* :577-581 */
    rm_point a = { nd->f[0], nd->f[1], nd->f[2] };
    float N = floorf(dot3(a, p) / dot3(a, a) + .5f);
    rm_point P = { p.x - N * a.x, p.y - N * a.y, p.z - N * a.z };
    return P;
}

float rmo_dist(const rm_node* nodes, size_t idx, rm_point p, rmo_counters* c) {
    const rm_node* nd = &nodes[idx];
    CNT_DIST(nd->type);
    switch (nd->type) {
    case RM_TYPE_PORTANTA_SFERO: {                                /* :53-61 */
        rm_point P = { p.x - nd->f[0], p.y - nd->f[1], p.z - nd->f[2] };
        float d = length_3(P.x, P.y, P.z) - nd->f[3];
        CNT(flops, 10);
        if (d <= RM_MIN_DIST) return rmo_dist(nodes, idx + nd->o[0], P, c);
        return d;
    }
    case RM_TYPE_SFERO:                                           /* :75 */
        CNT(flops, 7);
        return length_3(p.x, p.y, p.z) - nd->f[0];
    case RM_TYPE_KUBO: {                                          /* :84-91 */
        float qx = fabsf(p.x) - nd->f[0], qy = fabsf(p.y) - nd->f[1], qz = fabsf(p.z) - nd->f[2];
        if (qx < 0.f && qy < 0.f && qz < 0.f) {
            CNT(flops, 5);
            return fmax2(qx, fmax2(qy, qz));
        }
        CNT(flops, 12);
        return length_3(fmax2(qx, 0.f), fmax2(qy, 0.f), fmax2(qz, 0.f));
    }
    case RM_TYPE_CILINDRO: {                                      /* :110-117 */
        float r = length_2(p.x, p.y);
        float qx = r - nd->f[0], qy = fabsf(p.z) - nd->f[1];
        if (qx < 0.f && qy < 0.f) {
            CNT(flops, 6);
            return qx > qy ? qx : qy;
        }
        CNT(flops, 12);
        return length_2(fmax2(qx, 0.f), fmax2(qy, 0.f));
    }
    case RM_TYPE_EBENO: {                                         /* :132 */
        rm_point n = { nd->f[0], nd->f[1], nd->f[2] };
        CNT(flops, 5);
        return dot3(n, p);
    }
    case RM_TYPE_KUNIGAJO_2: case RM_TYPE_KUNIGAJO_3: case RM_TYPE_KUNIGAJO_4: {
        /* :141-145, :160-169, :195-207 : running min in child order */
        int na = arity(nd->type);
        float d = rmo_dist(nodes, idx + nd->o[0], p, c);
        for (int i = 1; i < na; ++i) d = fmin2(d, rmo_dist(nodes, idx + nd->o[i], p, c));
        CNT(flops, na - 1);
        return d;
    }
    case RM_TYPE_KOMUNAJO_2: case RM_TYPE_KOMUNAJO_3: case RM_TYPE_KOMUNAJO_4: {
        /* :240-244, :259-268, :294-306 : running max in child order */
        int na = arity(nd->type);
        float d = rmo_dist(nodes, idx + nd->o[0], p, c);
        for (int i = 1; i < na; ++i) d = fmax2(d, rmo_dist(nodes, idx + nd->o[i], p, c));
        CNT(flops, na - 1);
        return d;
    }
    case RM_TYPE_KOMPLEMENTO:                                     /* :339-341 */
        return -rmo_dist(nodes, idx + nd->o[0], p, c);
    case RM_TYPE_GLATA_KUNIGAJO_2: {                              /* :352-359 */
        float d0 = rmo_dist(nodes, idx + nd->o[0], p, c);
        float d1 = rmo_dist(nodes, idx + nd->o[1], p, c);
        float k = nd->f[0];
        float h = (1.f - (d0 - d1) / k) * .5f;
        CNT(flops, 4);
        if (h > 1.f) return d0;
        if (h < 0.f) return d1;
        CNT(flops, 7);
        return mixf(d0, d1, h) - k * h * (1.f - h);
    }
    case RM_TYPE_GLATA_KOMUNAJO_2: {                              /* :380-387 */
        float d0 = rmo_dist(nodes, idx + nd->o[0], p, c);
        float d1 = rmo_dist(nodes, idx + nd->o[1], p, c);
        float k = nd->f[0];
        float h = (1.f + (d0 - d1) / k) * .5f;
        CNT(flops, 4);
        if (h > 1.f) return d0;
        if (h < 0.f) return d1;
        CNT(flops, 7);
        return mixf(d0, d1, h) + k * h * (1.f - h);
    }
    case RM_TYPE_MOVO:                                            /* :409-414 */
        CNT(flops, 3);
        return rmo_dist(nodes, idx + nd->o[0], xf_movo(nd, p), c);
    case RM_TYPE_ROTACIOX:                                        /* :428-433 */
        CNT(flops, 6);
        return rmo_dist(nodes, idx + nd->o[0], xf_rotx(nd, p), c);
    case RM_TYPE_ROTACIOY:                                        /* :452-457 */
        CNT(flops, 6);
        return rmo_dist(nodes, idx + nd->o[0], xf_roty(nd, p), c);
    case RM_TYPE_ROTACIOZ:                                        /* :476-481 */
        CNT(flops, 6);
        return rmo_dist(nodes, idx + nd->o[0], xf_rotz(nd, p), c);
    case RM_TYPE_ROTACIOQ: {                                      /* :500-531 */
        qmat Q = build_q(nd);
        CNT(flops, 21);   /* the 20-flop matrix build is scene-constant (appendix D) */
        return rmo_dist(nodes, idx + nd->o[0], xf_rotq(&Q, p), c);
    }
    case RM_TYPE_SENFINA_RIPETO:                                  /* :576-582 */
        CNT(flops, 14);
        return rmo_dist(nodes, idx + nd->o[0], xf_ripeto(nd, p), c);
    default:
        /* type_nenio / unknown: the reference would call a null function
         * pointer (Include/defines.h:72).  rmo_validate() rejects such scenes
         * before rendering; reaching this is a harness bug. */
        abort();
    }
}

/* ----------------------------------------------------- node library: normal */

static rm_point pt(float x, float y, float z) { rm_point r = { x, y, z }; return r; }

rm_point rmo_norm(const rm_node* nodes, size_t idx, rm_point p, rmo_counters* c) {
    const rm_node* nd = &nodes[idx];
    CNT_NORM(nd->type);
    switch (nd->type) {
    case RM_TYPE_PORTANTA_SFERO:                                  /* :64-69 */
        CNT(flops, 3);
        return rmo_norm(nodes, idx + nd->o[0], pt(p.x - nd->f[0], p.y - nd->f[1], p.z - nd->f[2]), c);
    case RM_TYPE_SFERO:                                           /* :78 */
        return p;
    case RM_TYPE_KUBO: {                                          /* :94-104 */
        float qx = fabsf(p.x) - nd->f[0], qy = fabsf(p.y) - nd->f[1], qz = fabsf(p.z) - nd->f[2];
        CNT(flops, 3);
        if (qx < 0.f && qy < 0.f && qz < 0.f)
            return qx > qz ? (qx > qy ? pt(p.x > 0.f ? 1.f : -1.f, 0.f, 0.f)
                                      : pt(0.f, p.y > 0.f ? 1.f : -1.f, 0.f))
                           : (qy > qz ? pt(0.f, p.y > 0.f ? 1.f : -1.f, 0.f)
                                      : pt(0.f, 0.f, p.z > 0.f ? 1.f : -1.f));
        return pt(qx > 0.f ? (p.x > 0.f ? 1.f : -1.f) : 0.f,
                  qy > 0.f ? (p.y > 0.f ? 1.f : -1.f) : 0.f,
                  qz > 0.f ? (p.z > 0.f ? 1.f : -1.f) : 0.f);
    }
    case RM_TYPE_CILINDRO: {                                      /* :120-126 */
        float r = length_2(p.x, p.y);
        float qx = r - nd->f[0], qy = fabsf(p.z) - nd->f[1];
        CNT(flops, 6);
        return qx > qy ? pt(p.x, p.y, 0.f) : pt(0.f, 0.f, p.z > 0.f ? 1.f : -1.f);
    }
    case RM_TYPE_EBENO:                                           /* :135 */
        return pt(nd->f[0], nd->f[1], nd->f[2]);
    case RM_TYPE_KUNIGAJO_2: {                                    /* :148-154 */
        float d0 = rmo_dist(nodes, idx + nd->o[0], p, c);
        float d1 = rmo_dist(nodes, idx + nd->o[1], p, c);
        return rmo_norm(nodes, idx + nd->o[d0 < d1 ? 0 : 1], p, c);
    }
    case RM_TYPE_KUNIGAJO_3: case RM_TYPE_KUNIGAJO_4: {           /* :172-189, :210-234 */
        int na = arity(nd->type), i_min = 0;
        float d_min = rmo_dist(nodes, idx + nd->o[0], p, c);
        for (int i = 1; i < na; ++i) {
            float d = rmo_dist(nodes, idx + nd->o[i], p, c);
            if (d_min > d) { d_min = d; i_min = i; }
        }
        return rmo_norm(nodes, idx + nd->o[i_min], p, c);
    }
    case RM_TYPE_KOMUNAJO_2: {                                    /* :247-253 */
        float d0 = rmo_dist(nodes, idx + nd->o[0], p, c);
        float d1 = rmo_dist(nodes, idx + nd->o[1], p, c);
        return rmo_norm(nodes, idx + nd->o[d0 > d1 ? 0 : 1], p, c);
    }
    case RM_TYPE_KOMUNAJO_3: case RM_TYPE_KOMUNAJO_4: {           /* :271-288, :309-333 */
        int na = arity(nd->type), i_max = 0;
        float d_max = rmo_dist(nodes, idx + nd->o[0], p, c);
        for (int i = 1; i < na; ++i) {
            float d = rmo_dist(nodes, idx + nd->o[i], p, c);
            if (d_max < d) { d_max = d; i_max = i; }
        }
        return rmo_norm(nodes, idx + nd->o[i_max], p, c);
    }
    case RM_TYPE_KOMPLEMENTO: {                                   /* :344-346 */
        rm_point N = rmo_norm(nodes, idx + nd->o[0], p, c);
        return pt(-N.x, -N.y, -N.z);
    }
    case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2: { /* :362-374, :390-402 */
        float d0 = rmo_dist(nodes, idx + nd->o[0], p, c);
        float d1 = rmo_dist(nodes, idx + nd->o[1], p, c);
        float k = nd->f[0];
        float h = (nd->type == RM_TYPE_GLATA_KUNIGAJO_2) ? (1.f - (d0 - d1) / k) * .5f
                                                         : (1.f + (d0 - d1) / k) * .5f;
        CNT(flops, 4);
        if (h > 1.f) return rmo_norm(nodes, idx + nd->o[0], p, c);
        if (h < 0.f) return rmo_norm(nodes, idx + nd->o[1], p, c);
        rm_point n0 = rmo_norm(nodes, idx + nd->o[0], p, c);
        rm_point n1 = rmo_norm(nodes, idx + nd->o[1], p, c);
        d0 = r_length_3(n0.x, n0.y, n0.z);
        d1 = r_length_3(n1.x, n1.y, n1.z);
        CNT(flops, 27);
        return pt(mixf(d0 * n0.x, d1 * n1.x, h), mixf(d0 * n0.y, d1 * n1.y, h), mixf(d0 * n0.z, d1 * n1.z, h));
    }
    case RM_TYPE_MOVO:                                            /* :417-422 */
        CNT(flops, 3);
        return rmo_norm(nodes, idx + nd->o[0], xf_movo(nd, p), c);
    case RM_TYPE_ROTACIOX: {                                      /* :436-446 */
        float cs = nd->f[0], sn = nd->f[1];
        rm_point n = rmo_norm(nodes, idx + nd->o[0], xf_rotx(nd, p), c), P;
        P.y = cs * n.y - sn * n.z; P.z = sn * n.y + cs * n.z; P.x = n.x;
        CNT(flops, 12);
        return P;
    }
    case RM_TYPE_ROTACIOY: {                                      /* :460-470 */
        float cs = nd->f[0], sn = nd->f[1];
        rm_point n = rmo_norm(nodes, idx + nd->o[0], xf_roty(nd, p), c), P;
        P.z = cs * n.z - sn * n.x; P.x = sn * n.z + cs * n.x; P.y = n.y;
        CNT(flops, 12);
        return P;
    }
    case RM_TYPE_ROTACIOZ: {                                      /* :484-494 */
        float cs = nd->f[0], sn = nd->f[1];
        rm_point n = rmo_norm(nodes, idx + nd->o[0], xf_rotz(nd, p), c), P;
        P.x = cs * n.x - sn * n.y; P.y = sn * n.x + cs * n.y; P.z = n.z;
        CNT(flops, 12);
        return P;
    }
    case RM_TYPE_ROTACIOQ: {                                      /* :534-570 */
        qmat Q = build_q(nd);
        rm_point N = rmo_norm(nodes, idx + nd->o[0], xf_rotq(&Q, p), c), P = N;
        P.x += 2.f * (Q.xx * N.x + Q.yx * N.y + Q.zx * N.z);
        P.y += 2.f * (Q.xy * N.x + Q.yy * N.y + Q.zy * N.z);
        P.z += 2.f * (Q.xz * N.x + Q.yz * N.y + Q.zz * N.z);
        CNT(flops, 42);
        return P;
    }
    case RM_TYPE_SENFINA_RIPETO:                                  /* :585-591 */
        CNT(flops, 14);
        return rmo_norm(nodes, idx + nd->o[0], xf_ripeto(nd, p), c);
    default:
        abort();
    }
}

/* -------------------------------------------------------------- the kernels */

/* kernel_SetRays, Source/cuda_kernels.cu:740-760 */
rm_ray rmo_set_ray(size_t width, size_t height, const rm_rays_info* info, long long x, long long y) {
    float X = .5f * (float)(2 * x - (long long)width + 1);
    float Y = .5f * (float)(2 * y - (long long)height + 1);
    float Z = (float)info->Depth;
    rm_point pos, dp, s;
    pos.x = X * info->StartWVec.x + Y * info->StartHVec.x;
    pos.y = X * info->StartWVec.y + Y * info->StartHVec.y;
    pos.z = X * info->StartWVec.z + Y * info->StartHVec.z;
    dp.x = Z * info->StartDir.x; dp.y = Z * info->StartDir.y; dp.z = Z * info->StartDir.z;
    float R_1 = r_length_3(pos.x + dp.x, pos.y + dp.y, pos.z + dp.z);
    s.x = pos.x + dp.x; s.y = pos.y + dp.y; s.z = pos.z + dp.z;
    rm_ray r;
    r.d.x = R_1 * s.x; r.d.y = R_1 * s.y; r.d.z = R_1 * s.z;
    r.p.x = pos.x + info->StartPos.x; r.p.y = pos.y + info->StartPos.y; r.p.z = pos.z + info->StartPos.z;
    return r;
}

/* float -> unsigned char as the reference's casts at :694-699 do, with the
 * out-of-range case (undefined in C++, quirk q7) defined as saturation. */
static uint8_t to_u8(float v, rmo_counters* c) {
    if (!(v > -1.f)) { CNT(lum_out_of_range, 1); return 0; }
    if (v >= 256.f)  { CNT(lum_out_of_range, 1); return 255; }
    return (uint8_t)(int)v;
}

/* kernel_Process for one pixel, Source/cuda_kernels.cu:619-717 */
/* t_end: RM_MAX_DIST for the reference's march; a ray window's end (rmo_ray_window) for the windowed one,
 * negative for a ray that is not marched at all. */
/* window != NULL: the march as the product's kernel makes it (raytracing_amd/csrc/device_render.h) - besides the ray
 * window in t_end, a hit whose surface faces away from the light (N_L <= 0) is shaded without its shadow march, whose
 * result :694 multiplies by zero, and a shadow march ends where rmo_shadow_window proves that the rest of it can
 * change nothing.  Not the reference's march: it exists for the same two purposes as the windows (below). */
float rmo_shadow_window(const rm_scene_bound* b, rm_point o, rm_point l, float t0);
static void process_pixel(const rm_node* nodes, rm_ray r, rm_point light, float t_end, const rm_scene_bound* window,
                          uint8_t px[4], uint8_t* hit, uint16_t* evals, rmo_counters* c) {
    const int product = window != NULL;
    float curr_dist, ray_dist = 0.f;
    rm_point curr_norm;
    unsigned nev = 0;
    px[0] = 0; px[1] = 0; px[2] = 0; px[3] = 0xff;                        /* :622 */
    if (hit) *hit = 0;
    if (t_end < 0.f) { if (evals) *evals = 0; return; }

    for (int I = 0; I < RM_PRIMARY_STEPS; ++I) {                          /* :626 */
        curr_dist = rmo_dist(nodes, 0, r.p, c); ++nev;                    /* :627 */
        CNT(root_dist_primary, 1);
        if (curr_dist < RM_MIN_DIST) {                                    /* :629 */
            if (curr_dist < -0.f) {                                       /* :630-633 */
                curr_norm.x = -r.d.x; curr_norm.y = -r.d.y; curr_norm.z = -r.d.z;
            } else {                                                      /* :635-637 */
                curr_norm = rmo_norm(nodes, 0, r.p, c);
                CNT(root_norm, 1);
                float R_1 = r_length_3(curr_norm.x, curr_norm.y, curr_norm.z);
                curr_norm.x = R_1 * curr_norm.x; curr_norm.y = R_1 * curr_norm.y; curr_norm.z = R_1 * curr_norm.z;
                CNT(flops, 9);
            }
            CNT(flops, 5);
            if (dot3(curr_norm, r.d) < 0.f) {                             /* :640 */
                float N_L = dot3(curr_norm, light);                       /* :641 */
                float SHADOW = 1.f;
                rm_point p = r.p;                                         /* :654-658 */
                p.x += (1 + RM_DELTA) * RM_MIN_DIST * light.x;
                p.y += (1 + RM_DELTA) * RM_MIN_DIST * light.y;
                p.z += (1 + RM_DELTA) * RM_MIN_DIST * light.z;
                ray_dist = (1 + RM_DELTA) * RM_MIN_DIST;
                CNT(flops, 11);
                const float s_end = product && N_L > 0.f ? rmo_shadow_window(window, p, light, ray_dist) : RM_MAX_DIST;

                for (int J = 0; J < RM_SHADOW_STEPS && !(product && !(N_L > 0.f)); ++J) {   /* :660 */
                    curr_dist = rmo_dist(nodes, 0, p, c); ++nev;          /* :661 */
                    CNT(root_dist_shadow, 1);
                    if (8 * curr_dist < RM_MIN_DIST) {                    /* :662 */
                        if (curr_dist < -0.f) {                           /* :663-666 */
                            curr_norm.x = -light.x; curr_norm.y = -light.y; curr_norm.z = -light.z;
                        } else {                                          /* :668-670 */
                            curr_norm = rmo_norm(nodes, 0, p, c);
                            CNT(root_norm, 1);
                            float R_1 = r_length_3(curr_norm.x, curr_norm.y, curr_norm.z);
                            curr_norm.x = R_1 * curr_norm.x; curr_norm.y = R_1 * curr_norm.y; curr_norm.z = R_1 * curr_norm.z;
                            CNT(flops, 9);
                        }
                        CNT(flops, 5);
                        if (dot3(curr_norm, r.d) < RM_MIN_DIST) {         /* :673-676 */
                            SHADOW = 0.f;
                            break;
                        }
                    }
                    SHADOW = fmin2(SHADOW, RM_HARDNESS * curr_dist / ray_dist);   /* :679 */
                    CNT(flops, 4);   /* 8*cd, 128*cd, /t, min */
                    if (SHADOW < 0.01f) break;                            /* :680-681 */
                    p.x += curr_dist * light.x;                           /* :683-686 */
                    p.y += curr_dist * light.y;
                    p.z += curr_dist * light.z;
                    ray_dist += curr_dist;
                    CNT(flops, 7);
                    if (ray_dist >= s_end) break;                         /* :689-690: RM_MAX_DIST */
                }

                /* :693-701 */
                float lum = 0xff * (RM_MIN_LUM + (RM_MAX_LUM - RM_MIN_LUM) * (N_L > 0.f ? N_L : 0.f) * SHADOW);
                uint8_t LIGHT = to_u8(lum, c);
                px[0] = to_u8(LIGHT * RM_MATERIAL_R, c);
                px[1] = to_u8(LIGHT * RM_MATERIAL_G, c);
                px[2] = to_u8(LIGHT * RM_MATERIAL_B, c);
                px[3] = 0xff;
                CNT(flops, 7);
                CNT(hits, 1);
                if (hit) *hit = 1;
                if (!(px[0] | px[1] | px[2])) CNT(hit_colour_disagree, 1);
                break;                                                    /* :703 */
            }
        }
        r.p.x += curr_dist * r.d.x;                                       /* :707-710 */
        r.p.y += curr_dist * r.d.y;
        r.p.z += curr_dist * r.d.z;
        ray_dist += curr_dist;
        CNT(flops, 7);
        if (ray_dist >= t_end) break;                                     /* :712-714: RM_MAX_DIST */
    }
    if (evals) *evals = (uint16_t)nev;
}

/* The product's ray windows (raytracing_amd/csrc/device_render.h, ray_window), restated.  They are not part of
 * the reference: the windowed march exists here so that (a) tests can check on the CPU that it changes no pixel
 * (the frame of rmo_render_windowed must equal that of rmo_render for any volume that bounds the scene), and
 * (b) bench.py can count the FLOPs the windowed march executes, which are fewer than the reference's.
 * Returns 0 for a ray that is dropped; *t_end <= RM_MAX_DIST otherwise. */
int rmo_ray_window(const rm_scene_bound* b, rm_ray r, float* t_end) {
    *t_end = RM_MAX_DIST;
    if (!b || b->kind == RM_BOUND_NONE) return 1;
    /* the device applies this to the direction before its normalisation (the test is homogeneous in it); here it
     * gets the normalised one, so |v| is 1 to rounding */
    const float R = b->r + RM_WINDOW_MARGIN;
    const float ocx = r.p.x - b->c[0], ocy = r.p.y - b->c[1], ocz = r.p.z - b->c[2];
    int usable = fmaxf(fmaxf(fabsf(r.p.x), fabsf(r.p.y)), fabsf(r.p.z)) <= RM_WINDOW_MAX_COORD;
    float px = ocx, py = ocy, pz = ocz, qx = r.d.x, qy = r.d.y, qz = r.d.z;
    const float vv = r.d.x * r.d.x + r.d.y * r.d.y + r.d.z * r.d.z;
    float a = vv;
    if (b->kind == RM_BOUND_CYLINDER) {
        const float ux = b->u[0], uy = b->u[1], uz = b->u[2];
        const float ou = ocx * ux + ocy * uy + ocz * uz, vu = r.d.x * ux + r.d.y * uy + r.d.z * uz;
        px -= ou * ux; py -= ou * uy; pz -= ou * uz;
        qx -= vu * ux; qy -= vu * uy; qz -= vu * uz;
        a = qx * qx + qy * qy + qz * qz;
        usable = usable && a >= RM_WINDOW_MIN_SIN2 * vv;
    }
    const float rs = 1.f / sqrtf(a), inv_a = rs * rs;
    const float len = sqrtf(vv), si = len * rs;
    const float s0 = -(px * qx + py * qy + pz * qz) * inv_a;
    const float wx = px + s0 * qx, wy = py + s0 * qy, wz = pz + s0 * qz;
    const float p2 = wx * wx + wy * wy + wz * wz;
    const float mag = fabsf(ocx) + fabsf(ocy) + fabsf(ocz) + R;
    const float Re = R + 5e-6f * mag * si;
    const float gap = Re * Re - p2;
    const float s_leave = s0 + sqrtf(fmaxf(gap, 0.f) * inv_a) * 1.00001f;
    const float leave = s_leave * len + (RM_WINDOW_SLACK + 5e-6f * mag * (si * si) + 2e-6f * fabsf(s0 * len));
    if (usable && leave < RM_MAX_DIST) *t_end = leave;
    return !(usable && (gap < 0.f || leave < 0.f));
}

/* The product's shadow windows (device_render.h, shadow_window), restated: the ray_dist at which a shadow march that
 * starts at o with ray_dist t0 and moves by cd * l per step may end, because from there on it is outside the grown volume
 * for good, no sample can be near a surface and :679's quotient cannot fall below 1; RM_MAX_DIST where that is not proved. */
float rmo_shadow_window(const rm_scene_bound* b, rm_point o, rm_point l, float t0) {
    if (!b || b->kind == RM_BOUND_NONE) return RM_MAX_DIST;
    const float R = b->r + RM_WINDOW_MARGIN;
    const float ocx = o.x - b->c[0], ocy = o.y - b->c[1], ocz = o.z - b->c[2];
    float px = ocx, py = ocy, pz = ocz, qx = l.x, qy = l.y, qz = l.z;
    const float vv = l.x * l.x + l.y * l.y + l.z * l.z;
    int usable = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) <= RM_WINDOW_MAX_COORD && vv >= 0x1p-20f && vv <= 0x1p20f;
    float a = vv;
    if (b->kind == RM_BOUND_CYLINDER) {
        const float ux = b->u[0], uy = b->u[1], uz = b->u[2];
        const float ou = ocx * ux + ocy * uy + ocz * uz, vu = l.x * ux + l.y * uy + l.z * uz;
        px -= ou * ux; py -= ou * uy; pz -= ou * uz;
        qx -= vu * ux; qy -= vu * uy; qz -= vu * uz;
        a = qx * qx + qy * qy + qz * qz;
        usable = usable && a >= RM_WINDOW_MIN_SIN2 * vv;
    }
    const float rs = 1.f / sqrtf(a), inv_a = rs * rs;
    const float len = sqrtf(vv), si = len * rs, rlen = 1.f / len;
    const float s0 = -(px * qx + py * qy + pz * qz) * inv_a;
    const float wx = px + s0 * qx, wy = py + s0 * qy, wz = pz + s0 * qz;
    const float p2 = wx * wx + wy * wy + wz * wz;
    const float mag = fabsf(ocx) + fabsf(ocy) + fabsf(ocz) + R;
    const float Re = R + 5e-6f * mag * si;
    const float gap = Re * Re - p2;
    const float s_leave = s0 + sqrtf(fmaxf(gap, 0.f) * inv_a) * 1.00001f;
    const float t_end = t0 + s_leave + (RM_WINDOW_SLACK + 5e-6f * mag * (si * si) + 2e-6f * fabsf(s0 * len)) * rlen * 1.00001f;
    const int proved = usable && gap > 0.f && s_leave > 0.f
                       && t_end * (1.f / RM_HARDNESS) <= 4.5f
                       && a * gap >= Re * Re * (1.02f * 1.02f / (RM_HARDNESS * RM_HARDNESS));
    return proved ? t_end : RM_MAX_DIST;
}

void rmo_render_rows_windowed(const rm_node* nodes, size_t n, size_t width, size_t height,
                              const rm_rays_info* info, const rm_scene_bound* window, size_t y0, size_t y1,
                              uint8_t* rgba, uint8_t* hit, uint16_t* evals, rmo_counters* c) {
    (void)n;
    rm_point light = info->LightSource;                                   /* :623 */
    for (size_t y = y0; y < y1 && y < height; ++y)
        for (size_t x = 0; x < width; ++x) {
            rm_ray r = rmo_set_ray(width, height, info, (long long)x, (long long)y);
            size_t i = y * width + x;
            float t_end = RM_MAX_DIST;
            if (window && !rmo_ray_window(window, r, &t_end)) t_end = -1.f;
            process_pixel(nodes, r, light, t_end, window, rgba + 4 * i, hit ? hit + i : NULL,
                          evals ? evals + i : NULL, c);
        }
}

void rmo_render_rows(const rm_node* nodes, size_t n, size_t width, size_t height,
                     const rm_rays_info* info, size_t y0, size_t y1,
                     uint8_t* rgba, uint8_t* hit, uint16_t* evals, rmo_counters* c) {
    rmo_render_rows_windowed(nodes, n, width, height, info, NULL, y0, y1, rgba, hit, evals, c);
}

typedef struct {
    const rm_node* nodes; size_t n, width, height; const rm_rays_info* info; const rm_scene_bound* window;
    int tid, nthreads; uint8_t* rgba; uint8_t* hit; uint16_t* evals; rmo_counters c; int count;
} job_t;

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    /* rows dealt round-robin: the cost is concentrated mid-screen */
    for (size_t y = (size_t)j->tid; y < j->height; y += (size_t)j->nthreads)
        rmo_render_rows_windowed(j->nodes, j->n, j->width, j->height, j->info, j->window, y, y + 1,
                                 j->rgba, j->hit, j->evals, j->count ? &j->c : NULL);
    return NULL;
}

double rmo_render(const rm_node* nodes, size_t n, size_t width, size_t height,
                  const rm_rays_info* info, int threads,
                  uint8_t* rgba, uint8_t* hit, uint16_t* evals, rmo_counters* c) {
    return rmo_render_windowed(nodes, n, width, height, info, NULL, threads, rgba, hit, evals, c);
}

double rmo_render_windowed(const rm_node* nodes, size_t n, size_t width, size_t height,
                           const rm_rays_info* info, const rm_scene_bound* window, int threads,
                           uint8_t* rgba, uint8_t* hit, uint16_t* evals, rmo_counters* c) {
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    job_t* jobs = (job_t*)calloc((size_t)threads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; ++t) {
        job_t j = { nodes, n, width, height, info, window, t, threads, rgba, hit, evals, {{0}}, c != NULL };
        jobs[t] = j;
        if (threads > 1) pthread_create(&th[t], NULL, worker, &jobs[t]);
        else worker(&jobs[t]);
    }
    for (int t = 0; t < threads; ++t) {
        if (threads > 1) pthread_join(th[t], NULL);
        if (c) {
            const uint64_t* src = (const uint64_t*)&jobs[t].c;
            uint64_t* dst = (uint64_t*)c;
            for (size_t k = 0; k < sizeof(rmo_counters) / sizeof(uint64_t); ++k) dst[k] += src[k];
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(jobs); free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* Source/main.cpp:7,127-138 (camera, light), :261-266 (light step), :248-255 (drag) */
void rmo_default_info(rm_rays_info* info, int light_steps, int phi_steps) {
    const float M_SQRT1_5f = .447213595499957939281f;
    float scale = powf(2.f, -6.1f), theta = 0.f, cos_theta = 1.f, sin_theta = 0.f,
          phi = 0.f, cos_phi = 1.f, sin_phi = 0.f;
    float cos_1 = cosf(M_PI / 180.f), sin_1 = sinf(M_PI / 180.f), deg = M_PI / 360.f;
    (void)theta;
    for (int k = 0; k < phi_steps; ++k) {
        phi += (float)1 * deg;
        cos_phi = cosf(phi);
        sin_phi = sinf(phi);
    }
    memset(info, 0, sizeof(*info));
    info->Depth = 1000;
    info->LightSource = pt(-2.f * M_SQRT1_5f, 0.f, M_SQRT1_5f);
    info->StartPos = pt(0.f, 0.f, 0.f);
    info->StartDir = pt(scale * cos_theta * cos_phi, scale * cos_theta * sin_phi, scale * sin_theta);
    info->StartWVec = pt(scale * sin_phi, -scale * cos_phi, 0.f);
    info->StartHVec = pt(scale * sin_theta * cos_phi, scale * sin_theta * sin_phi, -scale * cos_theta);
    for (int k = 0; k < light_steps; ++k) {
        rm_point L = info->LightSource;
        info->LightSource = pt(cos_1 * L.x - sin_1 * L.y, sin_1 * L.x + cos_1 * L.y, L.z);
    }
}

/* Source/main.cpp:127-266 as a scripted event path: the state the reference's event loop keeps
 * (scale, theta, phi, the pause flag) and what each SDL event does to it.  Events are int32
 * triples (kind, a, b); RMO_EV_FRAME is one pass of the loop body after the events were polled:
 * the light turns unless paused (:261-266) and the block that SetInfo would receive is emitted. */
size_t rmo_camera_script(const int32_t* ev, size_t n_events, rm_rays_info* out, size_t cap) {
    const float M_SQRT1_5f = .447213595499957939281f, half_pi = 1.57079632679489661923f;   /* M_PI_2f */
    float scale = powf(2.f, -6.1f), theta = 0.f, cos_theta = 1.f, sin_theta = 0.f,
          phi = 0.f, cos_phi = 1.f, sin_phi = 0.f;
    float cos_1 = cosf(M_PI / 180.f), sin_1 = sinf(M_PI / 180.f), deg = M_PI / 360.f;
    int pause = 0;
    size_t frames = 0;
    rm_rays_info I;
    memset(&I, 0, sizeof I);
    I.Depth = 1000;
    I.LightSource = pt(-2.f * M_SQRT1_5f, 0.f, M_SQRT1_5f);
    I.StartPos = pt(0.f, 0.f, 0.f);
#define RMO_VECTORS() do { \
        I.StartDir = pt(scale * cos_theta * cos_phi, scale * cos_theta * sin_phi, scale * sin_theta); \
        I.StartWVec = pt(scale * sin_phi, -scale * cos_phi, 0.f); \
        I.StartHVec = pt(scale * sin_theta * cos_phi, scale * sin_theta * sin_phi, -scale * cos_theta); } while (0)
    RMO_VECTORS();
    for (size_t k = 0; k < n_events; ++k) {
        const int32_t kind = ev[3 * k], a = ev[3 * k + 1], b = ev[3 * k + 2];
        rm_point L = I.LightSource;
        switch (kind) {
        case RMO_EV_FRAME:                                                            /* :261-270 */
            if (!pause) I.LightSource = pt(cos_1 * L.x - sin_1 * L.y, sin_1 * L.x + cos_1 * L.y, L.z);
            if (frames < cap) out[frames] = I;
            ++frames;
            break;
        case RMO_EV_LEFT:                                                             /* :158-164 */
            I.LightSource = pt(cos_1 * L.x - sin_1 * L.y, sin_1 * L.x + cos_1 * L.y, L.z);
            break;
        case RMO_EV_RIGHT:                                                            /* :165-171 */
            I.LightSource = pt(cos_1 * L.x + sin_1 * L.y, -sin_1 * L.x + cos_1 * L.y, L.z);
            break;
        case RMO_EV_SPACE:                                                            /* :172-186 */
            scale = powf(2.f, -2.1f);
            theta = 0.f; cos_theta = 1.f; sin_theta = 0.f;
            phi = 0.f; cos_phi = 1.f; sin_phi = 0.f;
            I.StartPos = pt(0.f, 0.f, 0.f);
            RMO_VECTORS();
            break;
        case RMO_EV_PAUSE:  pause = 1; break;                                         /* :187-188 */
        case RMO_EV_RESUME: pause = 0; break;                                         /* :195-196 */
        case RMO_EV_WHEEL:                                                            /* :219-226 */
            scale *= powf(2.f, (float)a * .1f);
            RMO_VECTORS();
            break;
        case RMO_EV_PAN: {                                                            /* :231-238 */
            float S = 2.f / scale;
            I.StartPos.x -= (float)a * I.StartWVec.x * S;
            I.StartPos.y -= (float)a * I.StartWVec.y * S;
            I.StartPos.z -= (float)a * I.StartWVec.z * S;
            I.StartPos.x -= (float)b * I.StartHVec.x * S;
            I.StartPos.y -= (float)b * I.StartHVec.y * S;
            I.StartPos.z -= (float)b * I.StartHVec.z * S;
            RMO_VECTORS();                                                            /* :253-255 */
            break;
        }
        case RMO_EV_DRAG:                                                             /* :239-255 */
            theta += (float)b * deg;
            if (theta > half_pi) theta = half_pi;
            else if (theta < -half_pi) theta = -half_pi;
            cos_theta = cosf(theta);
            sin_theta = sinf(theta);
            phi += (float)a * deg;
            cos_phi = cosf(phi);
            sin_phi = sinf(phi);
            RMO_VECTORS();
            break;
        default: break;
        }
    }
#undef RMO_VECTORS
    return frames;
}
