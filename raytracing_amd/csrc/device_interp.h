// device_interp.h - the two generic evaluators of a flattened scene: the
// wave-uniform distance program and the per-lane tree walker (any scene, no
// run-time compilation).  hipcc only.
#pragma once
#include "device_math.h"
#include "scene_program.h"

namespace rm {

// =============================================================================
// Wave-uniform distance program (scene_program.h).  Every lane of the wave must
// call this together: `pc` and every slot word are wave-uniform.
//
// Fetch: one 32-byte slot = one s_load_dwordx8 into SGPRs, issued a whole slot
// ahead of its use (the scalar data cache answers in ~250 cycles on gfx950; the
// loads of a slot's operands must never sit on the dependent path).  The load is
// written as inline asm because the compiler would split it and wait for the
// opcode word first.  Operands are used straight from SGPRs by the VALU.
//
// Combine: the running value A starts at +inf (program start and after every
// push), so "set" is "min" and both kinds are one v_med3_f32 with the third
// operand -inf (min) or +inf (max).
//
// `stack` points at this lane's column of the workgroup's LDS stack; level k of
// it lives at stack[k * RM_BLOCK_THREADS].
// =============================================================================
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ u32x8 fetch_slot(const Slot* prog, uint32_t byte_offset) {
    u32x8 v;
    asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(v) : "s"(prog), "s"(byte_offset));
    return v;
}
// the fetched registers may not be read (or copied) before this
__device__ __forceinline__ void await_slot(u32x8& v) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v));
}

struct Machine {
    P3 F, P;          // frame point, work point
    float A, B;       // running value, register copy of value slot 0
    float tx, ty;     // rows x,y of a rotacioQ in flight
};

__device__ __forceinline__ float chain(float a, float v, float sel) { return __builtin_amdgcn_fmed3f(a, v, sel); }

// Executes one slot.  Returns the number of slots to skip after it (portanta_sfero only).
template <bool SRC_F>
__device__ __forceinline__ uint32_t exec_slot(Machine& m, const u32x8 w, float* stack, bool active) {
    const uint32_t w0 = w[0], op = w0 & 0xff, slot = w0 >> F_SLOT_SHIFT;
    const P3 S = SRC_F ? m.F : m.P;
    const float f1 = as_f(w[1]), f2 = as_f(w[2]), f3 = as_f(w[3]), f4 = as_f(w[4]),
                f5 = as_f(w[5]), f6 = as_f(w[6]);
    // -inf continues a min chain, +inf a max chain
    const float sel = as_f((w0 & (COMB_MAX << F_COMB_SHIFT)) ? 0x7f800000u : 0xff800000u);
    const uint32_t sign = (w0 & F_NEG) ? 0x80000000u : 0u;
    uint32_t skip = 0;
    bool wrote_p = false;
    switch (op) {
    case OP_MOVO:    m.P = movo(S, f1, f2, f3); wrote_p = true; break;
    case OP_ROTX:    m.P = rotx(S, f1, f2); wrote_p = true; break;
    case OP_ROTY:    m.P = roty(S, f1, f2); wrote_p = true; break;
    case OP_ROTZ:    m.P = rotz(S, f1, f2); wrote_p = true; break;
    case OP_ROTQ_XY:          // rows x and y of P = p + 2*(Q p); the point itself is still needed for row z
        m.tx = S.x + 2.f * (f1 * S.x + f2 * S.y + f3 * S.z);
        m.ty = S.y + 2.f * (f4 * S.x + f5 * S.y + f6 * S.z);
        break;
    case OP_ROTQ_Z:
        m.P = P3{m.tx, m.ty, S.z + 2.f * (f1 * S.x + f2 * S.y + f3 * S.z)};
        wrote_p = true;
        break;
    case OP_RIPETO:  m.P = ripeto(S, f1, f2, f3, f4, f5); wrote_p = true; break;
    case OP_SETF:    m.F = m.P; break;
    case OP_SFERO:   m.A = chain(m.A, as_f(__builtin_bit_cast(uint32_t, sfero_dist(movo(S, f1, f2, f3), f4)) ^ sign), sel); break;
    case OP_KUBO:    m.A = chain(m.A, as_f(__builtin_bit_cast(uint32_t, kubo_dist(movo(S, f1, f2, f3), f4, f5, f6)) ^ sign), sel); break;
    case OP_CILINDRO:m.A = chain(m.A, as_f(__builtin_bit_cast(uint32_t, cilindro_dist(movo(S, f1, f2, f3), f4, f5)) ^ sign), sel); break;
    case OP_EBENO:   m.A = chain(m.A, as_f(__builtin_bit_cast(uint32_t, ebeno_dist(movo(S, f1, f2, f3), f4, f5, f6)) ^ sign), sel); break;
    case OP_PUSHA_R: m.B = m.A; m.A = __builtin_inff(); break;
    case OP_POPC_R:  m.A = chain(m.B, m.A, sel); break;
    case OP_PUSHA:   stack[slot * RM_BLOCK_THREADS] = m.A; m.A = __builtin_inff(); break;
    case OP_POPC:    m.A = chain(stack[slot * RM_BLOCK_THREADS], m.A, sel); break;
    case OP_NEG:     m.A = -m.A; break;
    case OP_GLATA_R: m.A = glata(m.B, m.A, f1, f2, (w0 & F_NEG) != 0); break;
    case OP_GLATA:   m.A = glata(stack[slot * RM_BLOCK_THREADS], m.A, f1, f2, (w0 & F_NEG) != 0); break;
    case OP_PUSHF:
        stack[slot * RM_BLOCK_THREADS] = m.F.x;
        stack[(slot + 1) * RM_BLOCK_THREADS] = m.F.y;
        stack[(slot + 2) * RM_BLOCK_THREADS] = m.F.z;
        m.F = m.P;
        break;
    case OP_POPF:
        m.F.x = stack[slot * RM_BLOCK_THREADS];
        m.F.y = stack[(slot + 1) * RM_BLOCK_THREADS];
        m.F.z = stack[(slot + 2) * RM_BLOCK_THREADS];
        break;
    case OP_PORTANTA: {            // Source/cuda_kernels.cu:53-61
        m.P = movo(S, f1, f2, f3);
        float d = length_3(m.P.x, m.P.y, m.P.z) - f4;
        stack[slot * RM_BLOCK_THREADS] = d;
        // no lane inside the bounding sphere: the child's slots are dead for this wave
        if (__builtin_amdgcn_ballot_w64(d <= RM_MIN_DIST) == 0) skip = w[5];
        break;
    }
    case OP_CULL:                  // scene_program.h; lanes that do not march have no say
        if (cull_all(S, (w0 & F_NEG) ? -m.A : m.A, (unsigned long long)__builtin_amdgcn_ballot_w64(active), f1, f2, f3, f4)) skip = w[5];
        break;
    case OP_PORTANTA_END: {
        float d = stack[slot * RM_BLOCK_THREADS];
        m.A = (d <= RM_MIN_DIST) ? m.A : d;
        break;
    }
    default: break;
    }
    if (wrote_p && (w0 & F_DST_F)) m.F = m.P;
    return skip;
}

__device__ __forceinline__ float run_program(const Slot* __restrict__ prog, P3 E, float* stack, bool active) {
    Machine m;
    m.F = E; m.P = E;
    m.A = __builtin_inff(); m.B = 0.f; m.tx = 0.f; m.ty = 0.f;
    uint32_t pc = 0;
    u32x8 cur = fetch_slot(prog, 0);
    await_slot(cur);
    while ((cur[0] & 0xff) != OP_END) {
        u32x8 nxt = fetch_slot(prog, (pc + 1) * (uint32_t)sizeof(Slot));      // in flight while `cur` executes
        const uint32_t skip = (cur[0] & F_SRC_F) ? exec_slot<true>(m, cur, stack, active) : exec_slot<false>(m, cur, stack, active);
        pc += 1;
        if (skip) {                     // rare: the prefetched slot is not the next one
            await_slot(nxt);
            pc += skip;
            nxt = fetch_slot(prog, pc * (uint32_t)sizeof(Slot));
        }
        await_slot(nxt);
        cur = nxt;
    }
    return m.A;
}

// =============================================================================
// Per-lane tree walker over the LDS tree table: the reference's recursion
// (Include/defines.h:75-76) unrolled onto an explicit private stack.  Lanes may
// be at different nodes; used for normals and the child distances that select
// them, where control flow is per pixel by nature.
// =============================================================================
struct WalkStack {
    uint32_t hdr[kWalkDepth];       // node | child index << 16
    float acc[kWalkDepth];
    float x[kWalkDepth], y[kWalkDepth], z[kWalkDepth];
};

__device__ __forceinline__ P3 tree_transform(uint32_t code, const uint32_t* w, P3 p) {
    switch (code) {
    case TC_MOVO:   return movo(p, as_f(w[2]), as_f(w[3]), as_f(w[4]));
    case TC_ROTX:   return rotx(p, as_f(w[2]), as_f(w[3]));
    case TC_ROTY:   return roty(p, as_f(w[2]), as_f(w[3]));
    case TC_ROTZ:   return rotz(p, as_f(w[2]), as_f(w[3]));
    case TC_ROTQ:   return rotq(p, as_f(w[2]), as_f(w[3]), as_f(w[4]), as_f(w[5]), as_f(w[6]), as_f(w[7]),
                                as_f(w[8]), as_f(w[9]), as_f(w[10]));
    default:        return ripeto(p, as_f(w[2]), as_f(w[3]), as_f(w[4]), as_f(w[5]), as_f(w[6]));
    }
}

__device__ __forceinline__ float tree_leaf_dist(uint32_t code, const uint32_t* w, P3 p) {
    switch (code) {
    case TC_SFERO:    return sfero_dist(p, as_f(w[1]));
    case TC_KUBO:     return kubo_dist(p, as_f(w[1]), as_f(w[2]), as_f(w[3]));
    case TC_CILINDRO: return cilindro_dist(p, as_f(w[1]), as_f(w[2]));
    default:          return ebeno_dist(p, as_f(w[1]), as_f(w[2]), as_f(w[3]));
    }
}

// DIST(node `root`, p).  Frames are pushed above `base`.
__device__ __forceinline__ float walk_dist(const TreeNode* tree, uint32_t root, P3 p, WalkStack& st, int base) {
    int sp = base;
    uint32_t node = root;
    float v = 0.f;
    for (;;) {
        // ---- descend until a value is known
        for (;;) {
            const uint32_t* w = tree[node].w;
            const uint32_t code = w[0] & 0xff;
            if (code >= TC_SFERO && code <= TC_EBENO) { v = tree_leaf_dist(code, w, p); break; }
            if (code >= TC_MOVO) { p = tree_transform(code, w, p); node = w[1]; continue; }
            if (code == TC_PORTANTA) {                       // Source/cuda_kernels.cu:53-61
                P3 q = movo(p, as_f(w[2]), as_f(w[3]), as_f(w[4]));
                float d = length_3(q.x, q.y, q.z) - as_f(w[5]);
                if (d <= RM_MIN_DIST) { p = q; node = w[1]; continue; }
                v = d;
                break;
            }
            // komplemento, kunigajo_N, komunajo_N, glata_*: evaluate child 0 first
            st.hdr[sp] = node;
            st.x[sp] = p.x; st.y[sp] = p.y; st.z[sp] = p.z;
            ++sp;
            node = w[1];
        }
        // ---- hand the value to the waiting frames
        for (;;) {
            if (sp == base) return v;
            const uint32_t hdr = st.hdr[sp - 1];
            const uint32_t fnode = hdr & 0xffff, k = hdr >> 16;
            const uint32_t* w = tree[fnode].w;
            const uint32_t code = w[0] & 0xff, arity = (w[0] >> 8) & 0xff;
            if (code == TC_KOMPLEMENTO) { v = -v; --sp; continue; }            // :339-341
            if (code == TC_KUNIGAJO || code == TC_KOMUNAJO) {                  // :141-145 ... :294-306
                float acc = v;
                if (k > 0) acc = (code == TC_KUNIGAJO) ? min2(st.acc[sp - 1], v) : max2(st.acc[sp - 1], v);
                if (k + 1 == arity) { v = acc; --sp; continue; }
                st.acc[sp - 1] = acc;
                st.hdr[sp - 1] = fnode | ((k + 1) << 16);
                p = P3{st.x[sp - 1], st.y[sp - 1], st.z[sp - 1]};
                node = w[2 + k];
                break;
            }
            // glata_kunigajo_2 / glata_komunajo_2                              // :352-359, :380-387
            if (k == 0) {
                st.acc[sp - 1] = v;
                st.hdr[sp - 1] = fnode | (1u << 16);
                p = P3{st.x[sp - 1], st.y[sp - 1], st.z[sp - 1]};
                node = w[2];
                break;
            }
            v = glata(st.acc[sp - 1], v, as_f(w[3]), as_f(w[4]), code == TC_GLATA_KOMUNAJO);
            --sp;
        }
    }
}

// NORM(node `root`, p), unnormalised as in the reference.
__device__ __forceinline__ P3 walk_norm(const TreeNode* tree, uint32_t root, P3 p, WalkStack& st) {
    int sp = 0;
    uint32_t node = root;
    P3 N{0.f, 0.f, 0.f};
    for (;;) {
        // ---- descend to the leaf whose normal is wanted
        for (;;) {
            const uint32_t* w = tree[node].w;
            const uint32_t code = w[0] & 0xff, arity = (w[0] >> 8) & 0xff;
            if (code == TC_SFERO) { N = p; break; }                            // :78
            if (code == TC_KUBO) {                                             // :94-104
                float qx = fabsf(p.x) - as_f(w[1]), qy = fabsf(p.y) - as_f(w[2]), qz = fabsf(p.z) - as_f(w[3]);
                float sx = p.x > 0.f ? 1.f : -1.f, sy = p.y > 0.f ? 1.f : -1.f, sz = p.z > 0.f ? 1.f : -1.f;
                if (qx < 0.f && qy < 0.f && qz < 0.f) {
                    if (qx > qz) N = (qx > qy) ? P3{sx, 0.f, 0.f} : P3{0.f, sy, 0.f};
                    else         N = (qy > qz) ? P3{0.f, sy, 0.f} : P3{0.f, 0.f, sz};
                } else {
                    N = P3{qx > 0.f ? sx : 0.f, qy > 0.f ? sy : 0.f, qz > 0.f ? sz : 0.f};
                }
                break;
            }
            if (code == TC_CILINDRO) {                                         // :120-126
                float qx = length_2(p.x, p.y) - as_f(w[1]), qy = fabsf(p.z) - as_f(w[2]);
                N = (qx > qy) ? P3{p.x, p.y, 0.f} : P3{0.f, 0.f, p.z > 0.f ? 1.f : -1.f};
                break;
            }
            if (code == TC_EBENO) { N = P3{as_f(w[1]), as_f(w[2]), as_f(w[3])}; break; }   // :135
            if (code == TC_MOVO || code == TC_RIPETO) { p = tree_transform(code, w, p); node = w[1]; continue; }
            if (code == TC_PORTANTA) { p = movo(p, as_f(w[2]), as_f(w[3]), as_f(w[4])); node = w[1]; continue; }  // :64-69
            if (code >= TC_ROTX && code <= TC_ROTQ) {        // rotate in now, rotate the normal back later
                p = tree_transform(code, w, p);
                st.hdr[sp++] = node;
                node = w[1];
                continue;
            }
            if (code == TC_KOMPLEMENTO) { st.hdr[sp++] = node; node = w[1]; continue; }
            {
                // combinators: the children's distances decide where the normal comes from.
                // kunigajo_N / komunajo_N forward to the child with the smallest / largest
                // distance; ties go to child 1 for arity 2 (:153, :252) and to the earlier
                // child for arity 3 and 4 (:179,186,217,224,231 / :278,285,316,323,330).
                const bool nary = code == TC_KUNIGAJO || code == TC_KOMUNAJO;
                const bool is_min = code == TC_KUNIGAJO;
                const uint32_t kids = nary ? arity : 2u;
                float d0 = 0.f, d1 = 0.f, best = 0.f;
                uint32_t pick = 0;
                for (uint32_t k = 0; k < kids; ++k) {
                    float d = walk_dist(tree, w[1 + k], p, st, sp);
                    if (k == 0) { d0 = d; best = d; continue; }
                    if (k == 1) d1 = d;
                    if (kids == 2) pick = (is_min ? (d0 < d) : (d0 > d)) ? 0u : 1u;
                    else if (is_min ? (best > d) : (best < d)) { best = d; pick = k; }
                }
                if (nary) { node = w[1 + pick]; continue; }
                // glata_*: :362-374, :390-402
                float h = glata_h(d0, d1, as_f(w[3]), as_f(w[4]), code == TC_GLATA_KOMUNAJO);
                if (h > 1.f) { node = w[1]; continue; }
                if (h < 0.f) { node = w[2]; continue; }
                st.hdr[sp] = node;                 // phase 0: normal of child 0 pending
                st.acc[sp] = h;
                st.x[sp] = p.x; st.y[sp] = p.y; st.z[sp] = p.z;
                ++sp;
                node = w[1];
            }
        }
        // ---- carry the normal back up through the waiting frames
        for (;;) {
            if (sp == 0) return N;
            const uint32_t hdr = st.hdr[sp - 1];
            const uint32_t fnode = hdr & 0xffff, phase = hdr >> 16;
            const uint32_t* w = tree[fnode].w;
            const uint32_t code = w[0] & 0xff;
            if (code == TC_KOMPLEMENTO) { N = P3{-N.x, -N.y, -N.z}; --sp; continue; }      // :346
            if (code == TC_ROTX) {                                                          // :443-445
                float c = as_f(w[2]), s = as_f(w[3]);
                N = P3{N.x, c * N.y - s * N.z, s * N.y + c * N.z}; --sp; continue;
            }
            if (code == TC_ROTY) {                                                          // :467-469
                float c = as_f(w[2]), s = as_f(w[3]);
                N = P3{s * N.z + c * N.x, N.y, c * N.z - s * N.x}; --sp; continue;
            }
            if (code == TC_ROTZ) {                                                          // :491-493
                float c = as_f(w[2]), s = as_f(w[3]);
                N = P3{c * N.x - s * N.y, s * N.x + c * N.y, N.z}; --sp; continue;
            }
            if (code == TC_ROTQ) {                                                          // :566-569 (transpose)
                float xx = as_f(w[2]), xy = as_f(w[3]), xz = as_f(w[4]), yx = as_f(w[5]), yy = as_f(w[6]),
                      yz = as_f(w[7]), zx = as_f(w[8]), zy = as_f(w[9]), zz = as_f(w[10]);
                N = P3{N.x + 2.f * (xx * N.x + yx * N.y + zx * N.z),
                       N.y + 2.f * (xy * N.x + yy * N.y + zy * N.z),
                       N.z + 2.f * (xz * N.x + yz * N.y + zz * N.z)};
                --sp; continue;
            }
            // glata blend
            if (phase == 0) {
                p = P3{st.x[sp - 1], st.y[sp - 1], st.z[sp - 1]};
                st.x[sp - 1] = N.x; st.y[sp - 1] = N.y; st.z[sp - 1] = N.z;   // keep n0
                st.hdr[sp - 1] = fnode | (1u << 16);
                node = w[2];
                break;
            }
            {
                float h = st.acc[sp - 1];
                P3 n0{st.x[sp - 1], st.y[sp - 1], st.z[sp - 1]}, n1 = N;
                float d0 = r_length_3(n0.x, n0.y, n0.z), d1 = r_length_3(n1.x, n1.y, n1.z);
                N = P3{mixf(d0 * n0.x, d1 * n1.x, h), mixf(d0 * n0.y, d1 * n1.y, h), mixf(d0 * n0.z, d1 * n1.z, h)};
                --sp;
            }
        }
    }
}

// ---- workgroup prologue: stage the tree table in LDS -------------------------

__device__ __forceinline__ void stage_tree(const TreeNode* __restrict__ src, uint32_t n_nodes, uint32_t* lds) {
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = threadIdx.x; i < n_nodes * 3; i += RM_BLOCK_THREADS) d[i] = s[i];
    __syncthreads();
}

// Scene evaluator handed to render_body<> (device_render.h): program for DIST(root),
// walker for normals and for DIST when prm.evaluator == 1.
struct InterpScene {
    static constexpr bool kNarrow = false;       // (device_render.h: the march loops mask their ballots instead)
    const Slot* program;
    const TreeNode* tree;
    float* stack;
    uint32_t evaluator;
    typedef WalkStack Scratch;      // private memory, kept apart so the uniform members stay in SGPRs

    __device__ __forceinline__ void init(const RenderParams& prm) {
        extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
        program = prm.program;
        tree = reinterpret_cast<const TreeNode*>(lds);
        stack = reinterpret_cast<float*>(lds + prm.n_nodes * 12) + threadIdx.x;
        evaluator = prm.evaluator;
        stage_tree(prm.tree, prm.n_nodes, lds);
    }
    // every lane of the wave must call this together; `active`: this lane needs the value
    __device__ __forceinline__ float dist(P3 p, Scratch& st, bool active) {
        return evaluator == 0 ? run_program(program, p, stack, active) : walk_dist(tree, 0, p, st, 0);
    }
    __device__ __forceinline__ P3 norm(P3 p, Scratch& st) { return walk_norm(tree, 0, p, st); }
    // What the normal may re-use of the distance just evaluated at the same point: nothing here
    // (the generated scenes of jit.cpp keep their transformed points and child distances).
    struct Keep {};
    // amask: the lanes whose distance matters (a wave-level mask, as the march loop keeps it)
    __device__ __forceinline__ float dist_keep(P3 p, Scratch& st, unsigned long long amask, Keep&) { return dist(p, st, rm_lane_of_mask(amask)); }
    __device__ __forceinline__ P3 norm_keep(P3 p, Scratch& st, const Keep&) { return norm(p, st); }
};

}  // namespace rm
