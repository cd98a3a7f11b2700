// scene_program.h - the two device-side forms of a scene, shared by the host
// flattener (scene_program.cpp) and the kernels (raymarch_kernels.hip).
//
// A scene arrives as the reference's node table: an array of `primitives::bazo`
// whose child links are offsets relative to the node (Include/types.h:56-61,
// Source/cuda_kernels.cu:59,141-142).  The reference evaluates it by recursion
// through per-node function pointers (Include/defines.h:75-76).  Here the host
// turns it into
//
//   1. a TREE TABLE (TreeNode[n], 48 B per node) staged in LDS by every
//      workgroup.  Child links are absolute, scene constants are hoisted
//      (rotacioQ's matrix, senfina_ripeto's dot(a,a)).  Lanes walk it one by
//      one (divergently) where the reference's control flow is per-pixel:
//      surface normals, and the child distances normals are selected by.
//
//   2. a wave-uniform DISTANCE PROGRAM (Slot[m], 32 B per slot): the tree in
//      evaluation order as straight-line code for a small register machine.
//      Every lane of a wavefront executes the same slot at the same time, so
//      slots are fetched with scalar loads and their operands sit in SGPRs.
//      It evaluates DIST(root, p), the innermost cost of the march.
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

namespace rm {

// ---- tree table ------------------------------------------------------------

enum TreeCode : uint32_t {
    TC_PORTANTA = 0,  // w1 child, w2-4 t, w5 r
    TC_SFERO,         // w1 r
    TC_KUBO,          // w1-3 b
    TC_CILINDRO,      // w1 r, w2 h
    TC_EBENO,         // w1-3 n
    TC_KUNIGAJO,      // w1..w[arity] children      (arity in bits 8..15 of w0)
    TC_KOMUNAJO,      // w1..w[arity] children
    TC_KOMPLEMENTO,   // w1 child
    TC_GLATA_KUNIGAJO,// w1 w2 children, w3 k, w4 RN(1/k) or 0 (device_math.h div_const)
    TC_GLATA_KOMUNAJO,// w1 w2 children, w3 k, w4 RN(1/k) or 0
    TC_MOVO,          // w1 child, w2-4 t
    TC_ROTX,          // w1 child, w2 cos, w3 sin
    TC_ROTY,
    TC_ROTZ,
    TC_ROTQ,          // w1 child, w2-10 Q (xx xy xz yx yy yz zx zy zz)
    TC_RIPETO         // w1 child, w2-4 a, w5 dot(a,a), w6 RN(1/dot(a,a)) or 0
};

struct alignas(16) TreeNode { uint32_t w[12]; };
static_assert(sizeof(TreeNode) == 48, "tree node is 48 bytes");

// ---- distance program --------------------------------------------------------
//
// Machine state per lane: F (frame point), P (work point), A (accumulator),
// B (register copy of value-stack slot 0) and a per-lane LDS stack whose slot
// numbers are fixed when the program is built (the program has no loops, so the
// stack depth at every instruction is static).
//
// Slot word 0:  bits 0-7 opcode | bit 8 SRC_F | bit 9 DST_F | bit 10 NEG |
//               bits 11-12 combine (1 min, 2 max; the running value starts at +inf,
//               so a first value is a min too) | bits 16-31 stack slot
enum Op : uint32_t {
    OP_END = 0,
    // transforms: dst = T(src).  w1.. operands
    OP_MOVO,      // w1-3 t
    OP_ROTX,      // w1 cos, w2 sin
    OP_ROTY,
    OP_ROTZ,
    OP_ROTQ_XY,   // rows x,y of P = p + 2 (Q p) into temporaries: w1-6 = xx xy xz yx yy yz
    OP_ROTQ_Z,    // row z, then P = (tx, ty, z): w1-3 = zx zy zz   (always follows OP_ROTQ_XY, same src)
    OP_RIPETO,    // w1-3 a, w4 dot(a,a), w5 RN(1/dot(a,a)) or 0 (device_math.h div_const)
    OP_SETF,      // F = P
    // primitives: v = prim(src - t) ; v = NEG ? -v : v ; A = combine(A, v)
    OP_SFERO,     // w1-3 t, w4 r
    OP_KUBO,      // w1-3 t, w4-6 b
    OP_CILINDRO,  // w1-3 t, w4 r, w5 h
    OP_EBENO,     // w1-3 t, w4-6 n
    // value stack
    OP_PUSHA_R,   // B = A ; A = +inf
    OP_POPC_R,    // A = combine(B, A)
    OP_PUSHA,     // stack[slot] = A ; A = +inf
    OP_POPC,      // A = combine(stack[slot], A)
    OP_NEG,       // A = -A
    OP_GLATA_R,   // A = smooth(B, A, k)            w1 k, w2 RN(1/k) or 0, NEG bit: 0 kunigajo, 1 komunajo
    OP_GLATA,     // A = smooth(stack[slot], A, k)
    // frames
    OP_PUSHF,     // stack[slot..slot+2] = F ; F = P
    OP_POPF,      // F = stack[slot..slot+2]
    // portanta_sfero: P = src - t ; d = |P| - r ; stack[slot] = d ;
    //                 if no lane has d <= MIN_DIST, skip w5 slots (to OP_PORTANTA_END)
    OP_PORTANTA,      // w1-3 t, w4 r, w5 skip
    OP_PORTANTA_END,  // d = stack[slot] ; A = (d <= MIN_DIST) ? A : d
    // Culling inside a union (min) chain.  The next w5 slots compute A = min(A, DIST(sub-tree, F))
    // and nothing else; the sub-tree's surface lies inside the sphere (centre w1-3, F's
    // coordinates).  If for every lane that still marches |F - c| - r exceeds A by a margin that
    // covers every rounding error (w4 = r + margin, see scene_build.cpp), min() would return A
    // bit for bit, and the slots are skipped.
    // With F_NEG the chain is a max chain and the sub-tree is komplemento(X), X inside the sphere:
    // it is at most r - |F - c|, and the same test runs on -A.
    OP_CULL,          // w1-3 c, w4 r + margin, w5 skip
    OP_COUNT
};

enum : uint32_t {
    F_SRC_F = 1u << 8,
    F_DST_F = 1u << 9,
    F_NEG   = 1u << 10,
    F_COMB_SHIFT = 11,
    F_COMB_MASK  = 3u << 11,
    COMB_SET = 0, COMB_MIN = 1, COMB_MAX = 2,
    F_SLOT_SHIFT = 16
};

struct alignas(32) Slot { uint32_t w[8]; };
static_assert(sizeof(Slot) == 32, "program slot is 32 bytes");

}  // namespace rm
