// scene_build.cpp - bazo/.sce codecs, scene validation and flattening.
//
// Compiled without FMA contraction (-ffp-contract=off): the two scene constants
// hoisted here (rotacioQ's matrix, Source/cuda_kernels.cu:504-525, and
// senfina_ripeto's dot(a,a), :578) must round exactly as the per-call
// computation in the reference does.
#include "scene_build.h"

#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstring>

namespace rm {

// ---------------------------------------------------------------- type tables

int node_arity(uint32_t type) {
    switch (type) {
    case RM_TYPE_SFERO: case RM_TYPE_KUBO: case RM_TYPE_CILINDRO: case RM_TYPE_EBENO: return 0;
    case RM_TYPE_PORTANTA_SFERO: case RM_TYPE_KOMPLEMENTO: case RM_TYPE_MOVO:
    case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY: case RM_TYPE_ROTACIOZ:
    case RM_TYPE_ROTACIOQ: case RM_TYPE_SENFINA_RIPETO: return 1;
    case RM_TYPE_KUNIGAJO_2: case RM_TYPE_KOMUNAJO_2:
    case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2: return 2;
    case RM_TYPE_KUNIGAJO_3: case RM_TYPE_KOMUNAJO_3: return 3;
    case RM_TYPE_KUNIGAJO_4: case RM_TYPE_KOMUNAJO_4: return 4;
    default: return -1;
    }
}

int node_floats(uint32_t type) {
    switch (type) {
    case RM_TYPE_PORTANTA_SFERO: case RM_TYPE_ROTACIOQ: return 4;
    case RM_TYPE_KUBO: case RM_TYPE_EBENO: case RM_TYPE_MOVO: case RM_TYPE_SENFINA_RIPETO: return 3;
    case RM_TYPE_CILINDRO: case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY: case RM_TYPE_ROTACIOZ: return 2;
    case RM_TYPE_SFERO: case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2: return 1;
    default: return 0;
    }
}

// --------------------------------------------------------------------- codecs

// Payload structs (Include/objects_list.h:8-30): the child counters come first,
// the floats follow; with `cw`-byte counters the floats start at arity*cw.
int decode_bazo(const void* records, size_t n, int layout, rm_node* out) {
    if ((!records || !out) && n) return RM_ERR_INVALID_ARG;
    size_t type_at, data_at; int cw;
    switch (layout) {
    case RM_BAZO_LP64:  type_at = 0;  data_at = 4; cw = 8; break;
    case RM_BAZO_LLP64: type_at = 0;  data_at = 4; cw = 4; break;
    case RM_BAZO_SCE:   type_at = 40; data_at = 0; cw = 4; break;
    default: return RM_ERR_INVALID_ARG;
    }
    const unsigned char* rec = static_cast<const unsigned char*>(records);
    for (size_t i = 0; i < n; ++i, rec += RM_BAZO_BYTES) {
        rm_node nd;
        std::memset(&nd, 0, sizeof nd);
        std::memcpy(&nd.type, rec + type_at, 4);
        int na = node_arity(nd.type), nf = node_floats(nd.type);
        if (na < 0) { out[i] = nd; continue; }      // nenio / unknown: kept, rejected if reachable
        if (na * cw + nf * 4 > RM_PRIMITIVE_PAYLOAD) return RM_ERR_UNSUPPORTED;
        const unsigned char* d = rec + data_at;
        for (int k = 0; k < na; ++k) {
            if (cw == 4) { int32_t v; std::memcpy(&v, d + 4 * k, 4); nd.o[k] = v; }
            else {
                int64_t v; std::memcpy(&v, d + 8 * k, 8);
                if (v > INT32_MAX || v < INT32_MIN) return RM_ERR_BAD_SCENE;
                nd.o[k] = static_cast<int32_t>(v);
            }
        }
        std::memcpy(nd.f, d + na * cw, static_cast<size_t>(nf) * 4);
        out[i] = nd;
    }
    return RM_OK;
}

void encode_sce(const rm_node* nodes, size_t n, std::vector<unsigned char>& bytes) {
    bytes.assign(n * RM_BAZO_BYTES, 0);
    for (size_t i = 0; i < n; ++i) {
        unsigned char* rec = bytes.data() + i * RM_BAZO_BYTES;
        int na = node_arity(nodes[i].type), nf = node_floats(nodes[i].type);
        if (na < 0) na = 0;
        std::memcpy(rec, nodes[i].o, static_cast<size_t>(na) * 4);
        std::memcpy(rec + na * 4, nodes[i].f, static_cast<size_t>(nf) * 4);
        std::memcpy(rec + 40, &nodes[i].type, 4);
    }
}

int load_sce(const char* path, std::vector<rm_node>& out) {
    if (!path) return RM_ERR_INVALID_ARG;
    std::FILE* f = std::fopen(path, "rb");
    if (!f) return RM_ERR_IO;
    std::vector<unsigned char> bytes;
    unsigned char buf[4096];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + got);
    std::fclose(f);
    if (bytes.empty() || bytes.size() % RM_BAZO_BYTES) return RM_ERR_IO;
    out.resize(bytes.size() / RM_BAZO_BYTES);
    return decode_bazo(bytes.data(), out.size(), RM_BAZO_SCE, out.data());
}

int save_sce(const char* path, const rm_node* nodes, size_t n) {
    if (!path || (!nodes && n)) return RM_ERR_INVALID_ARG;
    std::vector<unsigned char> bytes;
    encode_sce(nodes, n, bytes);
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return RM_ERR_IO;
    size_t put = std::fwrite(bytes.data(), 1, bytes.size(), f);
    int rc = std::fclose(f);
    return (put == bytes.size() && rc == 0) ? RM_OK : RM_ERR_IO;
}

// ------------------------------------------------------------------ flattener

namespace {

inline uint32_t fbits(float v) { uint32_t u; std::memcpy(&u, &v, 4); return u; }

enum Src { SRC_P = 0, SRC_F = 1 };

struct Flattener {
    const rm_node* nodes;
    size_t n;
    const DivisorSet* exact = nullptr;
    bool cull = true;                // emit OP_CULL in union chains
    std::vector<Slot> prog;
    std::vector<int32_t> slot_node;  // parallel to prog
    std::vector<unsigned char> seen;
    uint32_t sp = 0, max_sp = 0;     // LDS stack depth in dwords
    bool f_live = false;             // an enclosing combinator still needs F
    bool b_live = false;             // register B holds a pushed value
    long last_xf = -1;               // slot index of the transform that last wrote P
    uint32_t depth = 0, max_depth = 0;
    int err = RM_OK;

    Slot& emit(uint32_t op, uint32_t flags = 0) {
        Slot s; std::memset(&s, 0, sizeof s);
        s.w[0] = op | flags;
        prog.push_back(s);
        slot_node.push_back(-1);
        last_xf = -1;
        if (prog.size() > kMaxProgramSlots) err = RM_ERR_TOO_LARGE;
        return prog.back();
    }
    uint32_t alloc(uint32_t dwords) {
        uint32_t at = sp; sp += dwords; if (sp > max_sp) max_sp = sp; return at;
    }

    // komplemento* / at most one movo / primitive  ->  one fused slot
    bool try_prim(size_t idx, Src src, uint32_t comb) {
        bool neg = false, have_t = false;
        float t[3] = {0.f, 0.f, 0.f};
        size_t at = idx;
        uint32_t hops = 0;
        for (;;) {
            const rm_node& nd = nodes[at];
            if (nd.type == RM_TYPE_KOMPLEMENTO) { neg = !neg; }
            else if (nd.type == RM_TYPE_MOVO && !have_t) { have_t = true; t[0] = nd.f[0]; t[1] = nd.f[1]; t[2] = nd.f[2]; }
            else break;
            at += static_cast<size_t>(nd.o[0]);
            ++hops;
        }
        const rm_node& nd = nodes[at];
        uint32_t op;
        switch (nd.type) {
        case RM_TYPE_SFERO: op = OP_SFERO; break;
        case RM_TYPE_KUBO: op = OP_KUBO; break;
        case RM_TYPE_CILINDRO: op = OP_CILINDRO; break;
        case RM_TYPE_EBENO: op = OP_EBENO; break;
        default: return false;
        }
        // a komplemento BELOW the movo negates the same value, so one flag covers
        // every order in which the two kinds of wrapper were nested
        Slot& s = emit(op, (src == SRC_F ? F_SRC_F : 0) | (neg ? F_NEG : 0) | (comb << F_COMB_SHIFT));
        slot_node.back() = static_cast<int32_t>(idx);
        s.w[1] = fbits(t[0]); s.w[2] = fbits(t[1]); s.w[3] = fbits(t[2]);
        for (int k = 0; k < node_floats(nd.type); ++k) s.w[4 + k] = fbits(nd.f[k]);
        if (depth + hops + 1 > max_depth) max_depth = depth + hops + 1;
        return true;
    }

    void push_a(bool& in_b, uint32_t& slot) {
        if (!b_live) { b_live = true; in_b = true; slot = 0; emit(OP_PUSHA_R); }
        else { in_b = false; slot = alloc(1); emit(OP_PUSHA, slot << F_SLOT_SHIFT); }
    }
    void pop_release(bool in_b) { if (in_b) b_live = false; else sp -= 1; }

    // make F hold the point currently in `src`; returns the LDS slot F was saved to, or -1
    long enter_frame(Src src) {
        if (src == SRC_F) return -1;
        if (f_live) {
            uint32_t slot = alloc(3);
            emit(OP_PUSHF, slot << F_SLOT_SHIFT);
            return static_cast<long>(slot);
        }
        if (last_xf >= 0) { long at = last_xf; prog[static_cast<size_t>(at)].w[0] |= F_DST_F; last_xf = -1; }
        else emit(OP_SETF);
        return -1;
    }
    void leave_frame(long slot) {
        if (slot >= 0) { emit(OP_POPF, static_cast<uint32_t>(slot) << F_SLOT_SHIFT); sp -= 3; }
    }

    bool descend(size_t idx) {
        if (err) return false;
        if (idx >= n) { err = RM_ERR_BAD_SCENE; return false; }
        if (node_arity(nodes[idx].type) < 0) { err = RM_ERR_BAD_SCENE; return false; }
        if (++depth > max_depth) max_depth = depth;
        if (depth > kMaxTreeDepth) { err = RM_ERR_TOO_LARGE; return false; }
        seen[idx] = 1;
        return true;
    }
    size_t child(size_t idx, int k) {
        int32_t o = nodes[idx].o[k];
        if (o <= 0 || idx + static_cast<size_t>(o) >= n) { err = RM_ERR_BAD_SCENE; return idx; }
        return idx + static_cast<size_t>(o);
    }

    // mark every node of a fused wrapper chain as reachable and check its links
    bool check_chain(size_t idx) {
        size_t at = idx;
        for (uint32_t hops = 0;; ++hops) {
            if (at >= n || node_arity(nodes[at].type) < 0 || depth + hops + 1 > kMaxTreeDepth) {
                err = at >= n || node_arity(nodes[at].type) < 0 ? RM_ERR_BAD_SCENE : RM_ERR_TOO_LARGE;
                return false;
            }
            seen[at] = 1;
            if (node_arity(nodes[at].type) != 1) return true;
            int32_t o = nodes[at].o[0];
            if (o <= 0) { err = RM_ERR_BAD_SCENE; return false; }
            uint32_t ty = nodes[at].type;
            if (ty != RM_TYPE_KOMPLEMENTO && ty != RM_TYPE_MOVO) return true;
            at += static_cast<size_t>(o);
        }
    }

    // A = DIST(node idx, point in src)
    void eval(size_t idx, Src src) {
        if (err) return;
        if (!check_chain(idx)) return;
        if (try_prim(idx, src, COMB_MIN)) return;    // the running value is +inf here: min == set
        if (!descend(idx)) return;
        const rm_node& nd = nodes[idx];
        const uint32_t sflag = (src == SRC_F ? F_SRC_F : 0);
        switch (nd.type) {
        case RM_TYPE_KOMPLEMENTO:
            eval(child(idx, 0), src);
            emit(OP_NEG);
            break;
        case RM_TYPE_MOVO: {
            Slot& s = emit(OP_MOVO, sflag);
            s.w[1] = fbits(nd.f[0]); s.w[2] = fbits(nd.f[1]); s.w[3] = fbits(nd.f[2]);
            last_xf = static_cast<long>(prog.size()) - 1;
            eval(child(idx, 0), SRC_P);
            break;
        }
        case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY: case RM_TYPE_ROTACIOZ: {
            uint32_t op = nd.type == RM_TYPE_ROTACIOX ? OP_ROTX : nd.type == RM_TYPE_ROTACIOY ? OP_ROTY : OP_ROTZ;
            Slot& s = emit(op, sflag);
            s.w[1] = fbits(nd.f[0]); s.w[2] = fbits(nd.f[1]);
            last_xf = static_cast<long>(prog.size()) - 1;
            eval(child(idx, 0), SRC_P);
            break;
        }
        case RM_TYPE_ROTACIOQ: {
            float Q[9];
            build_q(nd, Q);
            emit(OP_ROTQ_XY, sflag);
            size_t at = prog.size() - 1;
            for (int k = 0; k < 6; ++k) prog[at].w[1 + k] = fbits(Q[k]);
            emit(OP_ROTQ_Z, sflag);
            for (int k = 0; k < 3; ++k) prog[at + 1].w[1 + k] = fbits(Q[6 + k]);
            last_xf = static_cast<long>(at + 1);
            eval(child(idx, 0), SRC_P);
            break;
        }
        case RM_TYPE_SENFINA_RIPETO: {
            Slot& s = emit(OP_RIPETO, sflag);
            s.w[1] = fbits(nd.f[0]); s.w[2] = fbits(nd.f[1]); s.w[3] = fbits(nd.f[2]);
            s.w[4] = fbits(ripeto_aa(nd));
            s.w[5] = fbits(divisor_reciprocal(exact, ripeto_aa(nd)));
            last_xf = static_cast<long>(prog.size()) - 1;
            eval(child(idx, 0), SRC_P);
            break;
        }
        case RM_TYPE_PORTANTA_SFERO: {
            uint32_t slot = alloc(1);
            Slot& s = emit(OP_PORTANTA, sflag | (slot << F_SLOT_SHIFT));
            s.w[1] = fbits(nd.f[0]); s.w[2] = fbits(nd.f[1]); s.w[3] = fbits(nd.f[2]); s.w[4] = fbits(nd.f[3]);
            size_t at = prog.size() - 1;
            eval(child(idx, 0), SRC_P);
            prog[at].w[5] = static_cast<uint32_t>(prog.size() - (at + 1));
            emit(OP_PORTANTA_END, slot << F_SLOT_SHIFT);
            sp -= 1;
            break;
        }
        case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2: {
            long saved = enter_frame(src);
            bool was_live = f_live; f_live = true;
            eval(child(idx, 0), SRC_F);
            bool in_b; uint32_t slot;
            push_a(in_b, slot);
            eval(child(idx, 1), SRC_F);
            Slot& s = emit(in_b ? OP_GLATA_R : OP_GLATA,
                           (nd.type == RM_TYPE_GLATA_KOMUNAJO_2 ? F_NEG : 0) | (slot << F_SLOT_SHIFT));
            s.w[1] = fbits(nd.f[0]);
            s.w[2] = fbits(divisor_reciprocal(exact, nd.f[0]));
            pop_release(in_b);
            f_live = was_live;
            leave_frame(saved);
            break;
        }
        default: {   // kunigajo_N / komunajo_N
            int na = node_arity(nd.type);
            bool is_min = nd.type == RM_TYPE_KUNIGAJO_2 || nd.type == RM_TYPE_KUNIGAJO_3 || nd.type == RM_TYPE_KUNIGAJO_4;
            long saved = enter_frame(src);
            bool was_live = f_live; f_live = true;
            if (is_min && cull) {
                std::vector<Item> items;
                gather(idx, items);
                emit_union(items);
            } else {
                eval(child(idx, 0), SRC_F);
                for (int k = 1; k < na; ++k) accum(child(idx, k), is_min ? COMB_MIN : COMB_MAX);
            }
            f_live = was_live;
            leave_frame(saved);
            break;
        }
        }
        --depth;
    }

    // ---- bounding spheres (for OP_CULL) ------------------------------------------------
    // A sphere (centre c, radius r, in the coordinates of the node's input point) with
    // DIST(node, p) >= |p - c| - r for every p, in exact arithmetic; `reach` bounds the
    // magnitude of every intermediate coordinate offset below the node (the scale of the
    // rounding errors of its evaluation); `leaves` counts primitive slots.
    struct Bound { bool ok = false; double c[3] = {0, 0, 0}, r = 0, reach = 0; uint32_t leaves = 0; };

    static Bound enclose(const Bound& a, const Bound& b) {     // smallest sphere around two spheres
        Bound o; o.ok = true;
        double d[3] = { b.c[0] - a.c[0], b.c[1] - a.c[1], b.c[2] - a.c[2] };
        double dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (dist + b.r <= a.r) { o = a; }
        else if (dist + a.r <= b.r) { o = b; }
        else {
            o.r = (dist + a.r + b.r) / 2;
            double k = (o.r - a.r) / dist;
            for (int i = 0; i < 3; ++i) o.c[i] = a.c[i] + d[i] * k;
        }
        o.reach = a.reach > b.reach ? a.reach : b.reach;
        o.leaves = a.leaves + b.leaves;
        return o;
    }

    Bound bound_of(size_t idx, uint32_t guard = 0) {
        Bound none;
        if (idx >= n || guard > kMaxTreeDepth) return none;
        const rm_node& nd = nodes[idx];
        auto kid = [&](int k) -> size_t { int32_t o = nd.o[k]; return o > 0 ? idx + static_cast<size_t>(o) : n; };
        auto pos = [](float v) { return v > 0 ? static_cast<double>(v) : 0.0; };
        Bound b;
        switch (nd.type) {
        case RM_TYPE_SFERO: b.ok = true; b.r = pos(nd.f[0]); b.leaves = 1; break;
        case RM_TYPE_KUBO:
            b.ok = true; b.leaves = 1;
            b.r = std::sqrt(pos(nd.f[0]) * pos(nd.f[0]) + pos(nd.f[1]) * pos(nd.f[1]) + pos(nd.f[2]) * pos(nd.f[2]));
            break;
        case RM_TYPE_CILINDRO:
            b.ok = true; b.leaves = 1;
            b.r = std::sqrt(pos(nd.f[0]) * pos(nd.f[0]) + pos(nd.f[1]) * pos(nd.f[1]));
            break;
        case RM_TYPE_MOVO: {                       // child sees p - t
            b = bound_of(kid(0), guard + 1);
            if (!b.ok) return none;
            double tl = 0;
            for (int i = 0; i < 3; ++i) { b.c[i] += nd.f[i]; tl += static_cast<double>(nd.f[i]) * nd.f[i]; }
            b.reach += std::sqrt(tl);
            break;
        }
        case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY: case RM_TYPE_ROTACIOZ: case RM_TYPE_ROTACIOQ: {
            // child sees M p: the centre goes back through M^-1 = M^T, if M is a rotation
            double M[9];
            if (nd.type == RM_TYPE_ROTACIOQ) {
                float Q[9]; build_q(nd, Q);
                for (int i = 0; i < 9; ++i) M[i] = 2.0 * Q[i] + ((i % 4 == 0) ? 1.0 : 0.0);
            } else {
                const double c = nd.f[0], s_ = nd.f[1];
                const double X[9] = {1, 0, 0, 0, c, s_, 0, -s_, c}, Y[9] = {c, 0, -s_, 0, 1, 0, s_, 0, c}, Z[9] = {c, s_, 0, -s_, c, 0, 0, 0, 1};
                const double* src = nd.type == RM_TYPE_ROTACIOX ? X : nd.type == RM_TYPE_ROTACIOY ? Y : Z;
                for (int i = 0; i < 9; ++i) M[i] = src[i];
            }
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double dot = 0;
                    for (int k = 0; k < 3; ++k) dot += M[3 * i + k] * M[3 * j + k];
                    if (!(std::fabs(dot - (i == j ? 1.0 : 0.0)) < 1e-5)) return none;     // not a rotation (or NaN)
                }
            b = bound_of(kid(0), guard + 1);
            if (!b.ok) return none;
            double c0[3] = { b.c[0], b.c[1], b.c[2] };
            for (int j = 0; j < 3; ++j) b.c[j] = M[j] * c0[0] + M[3 + j] * c0[1] + M[6 + j] * c0[2];
            b.r = b.r * (1 + 1e-4) + 1e-6;
            break;
        }
        case RM_TYPE_KUNIGAJO_2: case RM_TYPE_KUNIGAJO_3: case RM_TYPE_KUNIGAJO_4: {
            const int na = node_arity(nd.type);
            b = bound_of(kid(0), guard + 1);
            for (int k = 1; k < na && b.ok; ++k) {
                Bound o = bound_of(kid(k), guard + 1);
                if (!o.ok) return none;
                b = enclose(b, o);
            }
            if (!b.ok) return none;
            break;
        }
        case RM_TYPE_KOMUNAJO_2: case RM_TYPE_KOMUNAJO_3: case RM_TYPE_KOMUNAJO_4: {
            // max(d_i) >= every d_i: any child's sphere will do, take the smallest
            const int na = node_arity(nd.type);
            double reach = 0; uint32_t leaves = 0;
            for (int k = 0; k < na; ++k) {
                Bound o = bound_of(kid(k), guard + 1);
                leaves += o.ok ? o.leaves : 1;
                if (o.reach > reach) reach = o.reach;
                if (o.ok && (!b.ok || o.r < b.r)) b = o;
            }
            if (!b.ok) return none;
            b.reach = reach; b.leaves = leaves;
            break;
        }
        default: return none;                      // unbounded or not handled: never culled
        }
        for (int i = 0; i < 3; ++i) if (!std::isfinite(b.c[i])) return none;
        if (!std::isfinite(b.r) || !std::isfinite(b.reach)) return none;
        return b;
    }

    // Emits OP_CULL for the sub-tree `idx` about to be merged into a min chain; returns the
    // slot to patch with the skip count, or -1.
    long begin_cull(size_t idx, uint32_t comb) {
        if (comb != COMB_MIN || !cull) return -1;
        Bound b = bound_of(idx);
        if (!b.ok || b.leaves < 2) return -1;
        return emit_cull(b);
    }
    long emit_cull(const Bound& b) {
        const double cl = std::sqrt(b.c[0] * b.c[0] + b.c[1] * b.c[1] + b.c[2] * b.c[2]);
        // the device compares |F - c|^2 with (A + w4)^2 (1 + 2^-9): a relative margin of 2^-10 on
        // the distance plus this absolute one, against rounding errors of a few 2^-23 of the same scale
        const double k = b.r * (1 + 1e-4) + (b.r + cl + b.reach + 1.0) / 1024.0;
        Slot& s = emit(OP_CULL, F_SRC_F);
        s.w[1] = fbits(static_cast<float>(b.c[0])); s.w[2] = fbits(static_cast<float>(b.c[1])); s.w[3] = fbits(static_cast<float>(b.c[2]));
        s.w[4] = fbits(static_cast<float>(k) * 1.0001f + 1e-6f);
        return static_cast<long>(prog.size()) - 1;
    }
    void end_cull(long at) {
        if (at >= 0) prog[static_cast<size_t>(at)].w[5] = static_cast<uint32_t>(prog.size() - (static_cast<size_t>(at) + 1));
    }

    // ---- normals of a scene that is one union of primitives ------------------------------
    // NORM(kunigajo_2) forwards to child 0 if d0 < d1, else to child 1 (Source/cuda_kernels.cu:153);
    // kunigajo_3/4 forward to the first child with the smallest distance (:179,186,217,224,231).
    // d_k is the exact minimum over the child's members, so the nested choice picks, among the
    // members with the smallest distance, the first one in this order: at a binary union the
    // members of child 1 before those of child 0, at a 3- or 4-way union child 0 first.
    void priority_order(size_t idx, std::vector<size_t>& out, uint32_t guard = 0) {
        if (idx >= n || guard > kMaxTreeDepth) return;
        const rm_node& nd = nodes[idx];
        if (!is_union(nd.type)) { out.push_back(idx); return; }
        const int na = node_arity(nd.type);
        auto kid = [&](int k) -> size_t { int32_t o = nd.o[k]; return o > 0 ? idx + static_cast<size_t>(o) : n; };
        if (na == 2) { priority_order(kid(1), out, guard + 1); priority_order(kid(0), out, guard + 1); }
        else for (int k = 0; k < na; ++k) priority_order(kid(k), out, guard + 1);
    }
    // union_rank of SceneBuild: non-empty iff the program is [transforms]* then only primitive
    // slots (min, source F) and culls, and those primitives are exactly the members of the union
    // found below the root's transforms.
    std::vector<uint32_t> single_union_ranks(const std::vector<Slot>& program, const std::vector<int32_t>& slot_nodes) {
        std::vector<uint32_t> none;
        size_t pc = 0;
        auto opof = [&](size_t at) { return program[at].w[0] & 0xff; };
        while (pc < program.size() && (opof(pc) == OP_MOVO || opof(pc) == OP_ROTX || opof(pc) == OP_ROTY || opof(pc) == OP_ROTZ ||
                                       opof(pc) == OP_ROTQ_XY || opof(pc) == OP_ROTQ_Z || opof(pc) == OP_RIPETO || opof(pc) == OP_SETF)) ++pc;
        size_t prims = 0;
        for (; pc < program.size() && opof(pc) != OP_END; ++pc) {
            const uint32_t w0 = program[pc].w[0], op = w0 & 0xff;
            if (op == OP_CULL) continue;
            if (op < OP_SFERO || op > OP_EBENO) return none;
            if (!(w0 & F_SRC_F) || ((w0 & F_COMB_MASK) >> F_COMB_SHIFT) != COMB_MIN || slot_nodes[pc] < 0) return none;
            ++prims;
        }
        if (prims < 2) return none;
        size_t at = 0;                               // the union below the root's transforms
        for (uint32_t hops = 0; at < n && hops <= kMaxTreeDepth; ++hops) {
            const uint32_t ty = nodes[at].type;
            if (is_union(ty)) break;
            if (ty != RM_TYPE_MOVO && ty != RM_TYPE_ROTACIOX && ty != RM_TYPE_ROTACIOY && ty != RM_TYPE_ROTACIOZ &&
                ty != RM_TYPE_ROTACIOQ && ty != RM_TYPE_SENFINA_RIPETO) return none;
            if (nodes[at].o[0] <= 0) return none;
            at += static_cast<size_t>(nodes[at].o[0]);
        }
        if (at >= n || !is_union(nodes[at].type)) return none;
        std::vector<size_t> order;
        priority_order(at, order);
        if (order.size() != prims) return none;
        std::vector<uint32_t> rank(n, 0);
        for (size_t k = 0; k < order.size(); ++k) rank[order[k]] = static_cast<uint32_t>(order.size() - k);
        for (size_t k = 0; k < program.size(); ++k)
            if (slot_nodes[k] >= 0 && rank[static_cast<size_t>(slot_nodes[k])] == 0) return none;   // a member the order does not know
        return rank;
    }

    // ---- unions as bounding-volume hierarchies -------------------------------------------
    // min is exact, commutative and associative, so the members of a union chain may be
    // evaluated in any order and grouping (only the sign of a zero can differ).
    // The chain is regrouped by position: a top-down median split over the members'
    // bounding spheres, an OP_CULL in front of every group.
    struct Item { size_t idx; uint32_t depth; Bound b; };
    static bool is_union(uint32_t ty) { return ty == RM_TYPE_KUNIGAJO_2 || ty == RM_TYPE_KUNIGAJO_3 || ty == RM_TYPE_KUNIGAJO_4; }

    // members of the chain rooted at union node `idx` (already entered with descend())
    void gather(size_t idx, std::vector<Item>& items) {
        const int na = node_arity(nodes[idx].type);
        for (int k = 0; k < na && !err; ++k) {
            const size_t c = child(idx, k);
            if (err || !check_chain(c)) return;
            if (is_union(nodes[c].type)) {
                if (!descend(c)) return;
                gather(c, items);
                --depth;
            } else {
                Item it; it.idx = c; it.depth = depth; it.b = bound_of(c);
                items.push_back(it);
            }
        }
    }
    void emit_item(const Item& it) {
        const uint32_t keep = depth;
        depth = it.depth;
        accum(it.idx, COMB_MIN);
        depth = keep;
    }
    void emit_group(std::vector<Item>& v, size_t lo, size_t hi, bool root) {
        if (err) return;
        if (hi - lo <= 2) { for (size_t i = lo; i < hi; ++i) emit_item(v[i]); return; }
        Bound e = v[lo].b;
        double mn[3], mx[3];
        for (int a = 0; a < 3; ++a) mn[a] = mx[a] = v[lo].b.c[a];
        for (size_t i = lo + 1; i < hi; ++i) {
            e = enclose(e, v[i].b);
            for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], v[i].b.c[a]); mx[a] = std::max(mx[a], v[i].b.c[a]); }
        }
        const long at = root ? -1 : emit_cull(e);
        int axis = 0;
        for (int a = 1; a < 3; ++a) if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
        const size_t mid = lo + (hi - lo) / 2;
        std::nth_element(v.begin() + static_cast<long>(lo), v.begin() + static_cast<long>(mid), v.begin() + static_cast<long>(hi),
                         [axis](const Item& x, const Item& y) { return x.b.c[axis] < y.b.c[axis] || (x.b.c[axis] == y.b.c[axis] && x.idx < y.idx); });
        emit_group(v, lo, mid, false);
        emit_group(v, mid, hi, false);
        end_cull(at);
    }
    // A = min(A, every member); A is +inf on entry of a chain, so the first member sets it
    void emit_union(std::vector<Item>& items) {
        std::vector<Item> bounded;
        for (const Item& it : items) { if (it.b.ok) bounded.push_back(it); else emit_item(it); }
        emit_group(bounded, 0, bounded.size(), true);
    }

    // A = comb(A, DIST(node idx, F))
    void accum(size_t idx, uint32_t comb) {
        if (err) return;
        if (!check_chain(idx)) return;
        if (try_prim(idx, SRC_F, comb)) return;
        const long cull_at = comb == COMB_MAX ? begin_cull_max(idx) : begin_cull(idx, comb);
        accum_body(idx, comb);
        end_cull(cull_at);
    }
    // The same test for a max chain (intersection): a member komplemento(X) with X inside a sphere
    // (c, r) is at most r - |F - c| (X's distance is at least |F - c| - r), so where that is below
    // the running maximum A for every marching lane, max() would return A.  The slot carries
    // F_NEG: the device applies the min-chain test to -A.
    long begin_cull_max(size_t idx) {
        if (!cull || idx >= n || nodes[idx].type != RM_TYPE_KOMPLEMENTO || nodes[idx].o[0] <= 0) return -1;
        Bound b = bound_of(idx + static_cast<size_t>(nodes[idx].o[0]));
        if (!b.ok || b.leaves < 3) return -1;      // the test costs about one sphere: measured no gain on chair.sce's pair
        const long at = emit_cull(b);
        prog[static_cast<size_t>(at)].w[0] |= F_NEG;
        return at;
    }
    void accum_body(size_t idx, uint32_t comb) {
        const uint32_t ty = idx < n ? nodes[idx].type : 0;
        const bool same_kind =
            (comb == COMB_MIN && (ty == RM_TYPE_KUNIGAJO_2 || ty == RM_TYPE_KUNIGAJO_3 || ty == RM_TYPE_KUNIGAJO_4)) ||
            (comb == COMB_MAX && (ty == RM_TYPE_KOMUNAJO_2 || ty == RM_TYPE_KOMUNAJO_3 || ty == RM_TYPE_KOMUNAJO_4));
        if (same_kind) {
            // min(A, min(x, y)) == min(min(A, x), y): min and max are exact, so a
            // nested combinator of the same kind continues the running chain
            if (!descend(idx)) return;
            int na = node_arity(ty);
            for (int k = 0; k < na; ++k) accum(child(idx, k), comb);
            --depth;
            return;
        }
        bool in_b; uint32_t slot;
        push_a(in_b, slot);
        eval(idx, SRC_F);
        emit(in_b ? OP_POPC_R : OP_POPC, (comb << F_COMB_SHIFT) | (slot << F_SLOT_SHIFT));
        pop_release(in_b);
    }

    // Source/cuda_kernels.cu:504-525, same operations in the same order
    static void build_q(const rm_node& nd, float Q[9]) {
        const float qw = nd.f[0], qx = nd.f[1], qy = nd.f[2], qz = nd.f[3];
        float xx = qx * qx, yy = qy * qy, zz = qz * qz, temp;
        temp = xx + yy + zz;
        xx -= temp; yy -= temp; zz -= temp;
        float xy = qx * qy; temp = qz * qw; float yx = xy + temp; xy -= temp;
        float yz = qy * qz; temp = qx * qw; float zy = yz + temp; yz -= temp;
        float zx = qz * qx; temp = qy * qw; float xz = zx + temp; zx -= temp;
        Q[0] = xx; Q[1] = xy; Q[2] = xz; Q[3] = yx; Q[4] = yy; Q[5] = yz; Q[6] = zx; Q[7] = zy; Q[8] = zz;
    }

    void build_tree(std::vector<TreeNode>& tree) {
        tree.assign(n, TreeNode{});
        for (size_t i = 0; i < n; ++i) {
            if (!seen[i]) continue;
            const rm_node& nd = nodes[i];
            TreeNode& t = tree[i];
            const int na = node_arity(nd.type);
            auto kid = [&](int k) { return static_cast<uint32_t>(i + static_cast<size_t>(nd.o[k])); };
            auto put = [&](int at, int count) { for (int k = 0; k < count; ++k) t.w[at + k] = fbits(nd.f[k]); };
            uint32_t code = 0;
            switch (nd.type) {
            case RM_TYPE_PORTANTA_SFERO: code = TC_PORTANTA; t.w[1] = kid(0); put(2, 4); break;
            case RM_TYPE_SFERO: code = TC_SFERO; put(1, 1); break;
            case RM_TYPE_KUBO: code = TC_KUBO; put(1, 3); break;
            case RM_TYPE_CILINDRO: code = TC_CILINDRO; put(1, 2); break;
            case RM_TYPE_EBENO: code = TC_EBENO; put(1, 3); break;
            case RM_TYPE_KUNIGAJO_2: case RM_TYPE_KUNIGAJO_3: case RM_TYPE_KUNIGAJO_4:
                code = TC_KUNIGAJO; for (int k = 0; k < na; ++k) t.w[1 + k] = kid(k); break;
            case RM_TYPE_KOMUNAJO_2: case RM_TYPE_KOMUNAJO_3: case RM_TYPE_KOMUNAJO_4:
                code = TC_KOMUNAJO; for (int k = 0; k < na; ++k) t.w[1 + k] = kid(k); break;
            case RM_TYPE_KOMPLEMENTO: code = TC_KOMPLEMENTO; t.w[1] = kid(0); break;
            case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2:
                code = nd.type == RM_TYPE_GLATA_KUNIGAJO_2 ? TC_GLATA_KUNIGAJO : TC_GLATA_KOMUNAJO;
                t.w[1] = kid(0); t.w[2] = kid(1); put(3, 1);
                t.w[4] = fbits(divisor_reciprocal(exact, nd.f[0]));
                break;
            case RM_TYPE_MOVO: code = TC_MOVO; t.w[1] = kid(0); put(2, 3); break;
            case RM_TYPE_ROTACIOX: code = TC_ROTX; t.w[1] = kid(0); put(2, 2); break;
            case RM_TYPE_ROTACIOY: code = TC_ROTY; t.w[1] = kid(0); put(2, 2); break;
            case RM_TYPE_ROTACIOZ: code = TC_ROTZ; t.w[1] = kid(0); put(2, 2); break;
            case RM_TYPE_ROTACIOQ: {
                code = TC_ROTQ; t.w[1] = kid(0);
                float Q[9]; build_q(nd, Q);
                for (int k = 0; k < 9; ++k) t.w[2 + k] = fbits(Q[k]);
                break;
            }
            case RM_TYPE_SENFINA_RIPETO:
                code = TC_RIPETO; t.w[1] = kid(0); put(2, 3);
                t.w[5] = fbits(ripeto_aa(nd));
                t.w[6] = fbits(divisor_reciprocal(exact, ripeto_aa(nd)));
                break;
            default: break;
            }
            t.w[0] = code | (static_cast<uint32_t>(na) << 8);
        }
    }
};

}  // namespace

// Source/cuda_kernels.cu:578: dot(a, a), summed left to right
float ripeto_aa(const rm_node& nd) { return nd.f[0] * nd.f[0] + nd.f[1] * nd.f[1] + nd.f[2] * nd.f[2]; }

std::vector<float> scene_divisors(const rm_node* nodes, size_t n) {
    std::set<uint32_t> have;
    std::vector<float> out;
    for (size_t i = 0; i < n; ++i) {
        float c;
        if (nodes[i].type == RM_TYPE_SENFINA_RIPETO) c = ripeto_aa(nodes[i]);
        else if (nodes[i].type == RM_TYPE_GLATA_KUNIGAJO_2 || nodes[i].type == RM_TYPE_GLATA_KOMUNAJO_2) c = nodes[i].f[0];
        else continue;
        uint32_t u; std::memcpy(&u, &c, 4);
        if (have.insert(u).second) out.push_back(c);
    }
    return out;
}

float divisor_reciprocal(const DivisorSet* exact, float c) {
    uint32_t u; std::memcpy(&u, &c, 4);
    if (!exact || !exact->count(u)) return 0.f;
    return 1.0f / c;                     // correctly rounded: IEEE division on the host
}

int build_scene(const rm_node* nodes, size_t n, SceneBuild& out, const DivisorSet* exact, bool cull) {
    if (!nodes || n == 0) return RM_ERR_INVALID_ARG;
    if (n > RM_MAX_NODES) return RM_ERR_TOO_LARGE;
    Flattener fl;
    fl.nodes = nodes; fl.n = n; fl.exact = exact; fl.cull = cull;
    fl.seen.assign(n, 0);
    fl.eval(0, SRC_F);
    if (fl.err) return fl.err;
    fl.emit(OP_END);
    fl.emit(OP_END);      // the interpreter fetches one slot past the one it executes
    if (fl.err) return fl.err;
    out.program.swap(fl.prog);
    out.slot_node.swap(fl.slot_node);
    out.union_rank = fl.single_union_ranks(out.program, out.slot_node);
    out.value_slots = fl.max_sp;
    out.tree_depth = fl.max_depth;
    out.reachable = 0;
    for (unsigned char s : fl.seen) out.reachable += s;
    fl.build_tree(out.tree);
    out.seen = fl.seen;
    return RM_OK;
}

std::string disassemble(const std::vector<Slot>& program) {
    static const char* names[OP_COUNT] = {
        "END", "MOVO", "ROTX", "ROTY", "ROTZ", "ROTQ_XY", "ROTQ_Z", "RIPETO", "SETF",
        "SFERO", "KUBO", "CILINDRO", "EBENO",
        "PUSHA_R", "POPC_R", "PUSHA", "POPC", "NEG", "GLATA_R", "GLATA",
        "PUSHF", "POPF", "PORTANTA", "PORTANTA_END", "CULL" };
    static const char* combs[4] = { "-", "min", "max", "?" };
    std::string s;
    char line[256];
    for (size_t pc = 0; pc < program.size(); ++pc) {
        const uint32_t w0 = program[pc].w[0];
        const uint32_t op = w0 & 0xff;
        float f[7];
        std::memcpy(f, &program[pc].w[1], sizeof f);
        std::snprintf(line, sizeof line, "%4zu %-12s src=%c dst=%c%s comb=%s slot=%u  %g %g %g %g %g %g %g\n",
                      pc, op < OP_COUNT ? names[op] : "?", (w0 & F_SRC_F) ? 'F' : 'P', (w0 & F_DST_F) ? 'F' : 'P',
                      (w0 & F_NEG) ? " neg" : "", combs[(w0 & F_COMB_MASK) >> F_COMB_SHIFT], w0 >> F_SLOT_SHIFT,
                      f[0], f[1], f[2], f[3], f[4], f[5], f[6]);
        s += line;
        if (op == OP_END) break;
    }
    return s;
}

}  // namespace rm
