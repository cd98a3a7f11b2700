// jit.cpp - scene -> HIP source -> hiprtc -> module (see jit.h).
#include "jit.h"

#include <hip/hiprtc.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "scene_build.h"

namespace rm {

namespace {

// device headers, flattened at build time by make_jit_bundle.py
const char kPrelude[] =
#include "jit_bundle.inc"
    ;

// a float as its exact bit pattern: no decimal round trip, infinities and NaNs included
std::string lit(float v) {
    uint32_t u; std::memcpy(&u, &v, 4);
    char buf[32];
    std::snprintf(buf, sizeof buf, "as_f(0x%08xu)", u);
    return buf;
}
std::string lit3(const float* f) { return lit(f[0]) + ", " + lit(f[1]) + ", " + lit(f[2]); }

std::string fn(char kind, size_t idx) { return std::string(1, kind) + std::to_string(idx); }

// ripeto(point, a, dot(a,a), its reciprocal), the zero components of `a` named (device_math.h, ripeto_t)
std::string ripeto_call(const std::string& pv, float ax, float ay, float az, float aa, float raa) {
    auto z = [](float v) { return v == 0.f ? "true" : "false"; };
    return std::string("ripeto_t<") + z(ax) + ", " + z(ay) + ", " + z(az) + ">(" + pv + ", " + lit(ax) + ", " + lit(ay) + ", " +
           lit(az) + ", " + lit(aa) + ", " + lit(raa) + ")";
}

// Source/cuda_kernels.cu:504-525, same operations in the same order as scene_build.cpp
void build_q(const rm_node& nd, float Q[9]) {
    const float qw = nd.f[0], qx = nd.f[1], qy = nd.f[2], qz = nd.f[3];
    float xx = qx * qx, yy = qy * qy, zz = qz * qz, temp;
    temp = xx + yy + zz;
    xx -= temp; yy -= temp; zz -= temp;
    float xy = qx * qy; temp = qz * qw; float yx = xy + temp; xy -= temp;
    float yz = qy * qz; temp = qx * qw; float zy = yz + temp; yz -= temp;
    float zx = qz * qx; temp = qy * qw; float xz = zx + temp; zx -= temp;
    Q[0] = xx; Q[1] = xy; Q[2] = xz; Q[3] = yx; Q[4] = yy; Q[5] = yz; Q[6] = zx; Q[7] = zy; Q[8] = zz;
}

// the two kernels built on a JitScene, whichever generator wrote it
const char kJitKernels[] = R"(
#ifndef RM_KERNEL_ATTR
#define RM_KERNEL_ATTR
#endif
extern "C" __global__ void __launch_bounds__(RM_BLOCK_THREADS) RM_KERNEL_ATTR rm_render_jit(const rm::RenderParams prm) {
    rm::render_body<rm::JitScene>(prm);
}

extern "C" __global__ void __launch_bounds__(RM_BLOCK_THREADS)
rm_eval_points_jit(const rm::RenderParams prm, const float* __restrict__ xyz, uint32_t count,
                   float* __restrict__ dist_out, float* __restrict__ norm_out) {
    rm::JitScene scene;
    rm::JitScene::Scratch scratch;
    scene.init(prm);
    const uint32_t i = blockIdx.x * RM_BLOCK_THREADS + threadIdx.x;
    const bool live = i < count;
    rm::P3 p{0.f, 0.f, 0.f};
    if (live) p = rm::P3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    rm::JitScene::Keep keep;
    const unsigned long long amask = __builtin_amdgcn_ballot_w64(live);
    const float d = scene.dist_keep(p, scratch, amask, keep);         // wave-uniform code: every lane runs it
    // The march takes a normal only where the distance says a surface is near; this hook takes one at every point,
    // so its distance pass may not skip what the march's may (Fused::bound_skip)
    if (norm_out) (void)scene.dist_keep_all(p, scratch, amask, keep);
    if (live) {
        dist_out[i] = d;
        if (norm_out) {
            rm::P3 N = scene.norm_keep(p, scratch, keep);
            norm_out[3 * i] = N.x; norm_out[3 * i + 1] = N.y; norm_out[3 * i + 2] = N.z;
        }
    }
}
)";

uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}


// -----------------------------------------------------------------------------
// Distance and normal of a small scene as ONE pair of functions over named
// temporaries: the march evaluates the distance every trip, and a lane that then
// needs its normal (Source/cuda_kernels.cu:635, :668) needs it at the same point.
// The normal re-uses the transformed points and the child distances the distance
// pass just computed (struct SceneKeep, registers after inlining) and adds only
// its own selects, the leaf's normal and the rotations back.  The separate
// n<i>(p) functions recompute all of that; they remain for sub-trees behind a
// portanta_sfero, whose distance is evaluated only near its sphere.
// Inner min/max are v_med3 against an infinity as in the distance program: they
// differ from the reference's compare-and-select only in the sign of a zero,
// which no comparison, glata_h or later arithmetic turns into another value.
// -----------------------------------------------------------------------------
struct Fused {
    const rm_node* nodes;
    const DivisorSet* exact;
    std::string decl, dist, norm;
    int counter = 0;
    // A temporary of the distance pass and the region of the pass that computes it: 0 = every call, r > 0 =
    // only when the r-th skippable block runs (bound_skip below).  The distance pass re-uses a temporary only
    // where it is sure to have been computed; the normal's pass may use any (it runs only after a trip in
    // which no block was skipped, see bound_skip).
    struct Temp { std::string var; int region; };
    std::map<std::pair<size_t, std::string>, Temp> dmemo, pmemo;
    int region = 0, regions = 0;      // the block being emitted (0: none); blocks so far
    int glata_depth = 0;              // smooth combinators above the node being emitted
    bool in_norm = false;             // emitting the normal's pass
    bool allow_skip = true, allow_sphere_skip = true, allow_cell_cache = true;
    std::string cache_decl;           // struct SceneCache: the lattices' cells, kept from one evaluation to the next
    Fused(const rm_node* nodes_, const DivisorSet* exact_) : nodes(nodes_), exact(exact_) {
        if (const char* env = std::getenv("RM_BOUND_SKIP")) allow_skip = std::atoi(env) != 0;      // experiments only
        if (const char* env = std::getenv("RM_SPHERE_SKIP")) allow_sphere_skip = std::atoi(env) != 0;
        if (const char* env = std::getenv("RM_CELL_CACHE")) cell_cache_forced = std::atoi(env);
    }
    int cell_cache_forced = -1;       // RM_CELL_CACHE (experiments): 0 never, 1 always
    // cell_cache keeps one to four values per lattice and lane alive across trips.  It is used where the lattices are
    // a large part of an evaluation (infinity.sce: half of it) and left out where they are a small one and the registers
    // are worth more (chair.sce: 7 %, and the kernel no longer fits eight wavefronts per SIMD with it: measured 66.5 with,
    // 69.2 Grays/s without).  The shares come from a rough count of vector instructions per node type.
    void weigh_lattices(size_t n, const unsigned char* seen) {
        double total = 0.0, saved = 0.0;
        for (size_t i = 0; i < n; ++i) {
            if (!seen[i]) continue;
            const rm_node& nd = nodes[i];
            switch (nd.type) {
            case RM_TYPE_SFERO: total += 13; break;
            case RM_TYPE_KUBO: total += 19; break;
            case RM_TYPE_CILINDRO: total += 40; break;
            case RM_TYPE_EBENO: total += 6; break;
            case RM_TYPE_PORTANTA_SFERO: total += 14; break;
            case RM_TYPE_MOVO: total += 3; break;
            case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY: case RM_TYPE_ROTACIOZ: total += 6; break;
            case RM_TYPE_ROTACIOQ: total += 18; break;
            case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2: total += 20; break;
            case RM_TYPE_SENFINA_RIPETO: {
                int nz = 0;
                for (int k = 0; k < 3; ++k) nz += nd.f[k] != 0.f;
                total += nz <= 1 ? 9 : 9 + 2 * nz;
                if (cell_cacheable(nd)) saved += 6.5;
                break;
            }
            default: total += node_arity(nd.type) > 1 ? node_arity(nd.type) - 1 : 0; break;
            }
        }
        allow_cell_cache = cell_cache_forced >= 0 ? cell_cache_forced != 0 : saved >= 0.15 * total;
    }
    bool usable(const Temp& t) const { return in_norm || t.region == 0 || t.region == region; }

    // lean: the normal's pass re-uses the distance pass's temporaries down to the first node that is not a transform
    // (`anchor`) only, and takes the normal of that sub-tree by the plain n<i>() functions, which compute what they need
    // themselves.  The temporaries below the anchor are then locals of the distance pass, dead at its end, not registers
    // that live across the march loop.
    // It pays where many temporaries sit below the anchor and normals are rare (chair.sce: six of them, 14.5 k normal
    // passes a frame against 909 k trips; 69.3 -> 71.1 Grays/s), and costs a second evaluation of the sub-tree per normal:
    // taken from kLeanFloats temporaries on (generate_scene_source tries the plain form first and counts them).
    bool lean = false, below_anchor = false;
    size_t anchor = 0;
    int below_floats = 0;             // registers the temporaries below the anchor take
    static constexpr int kLeanFloats = 6;
    std::string locals;
    std::string new_f() {
        std::string v = "d" + std::to_string(counter++);
        if (below_anchor && !in_norm) { ++below_floats; if (lean) { locals += "    float t" + v + ";\n"; return "t" + v; } }      // (d<i> and n<i> are the nodes' functions)
        decl += "    float " + v + ";\n"; return "k." + v;
    }
    std::string new_p() {
        std::string v = "p" + std::to_string(counter++);
        if (below_anchor && !in_norm) { below_floats += 3; if (lean) { locals += "    P3 t" + v + ";\n"; return "t" + v; } }
        decl += "    P3 " + v + ";\n"; return "k." + v;
    }
    size_t kid(size_t i, int k) const { return i + static_cast<size_t>(nodes[i].o[k]); }
    static bool is_min(uint32_t t) { return t == RM_TYPE_KUNIGAJO_2 || t == RM_TYPE_KUNIGAJO_3 || t == RM_TYPE_KUNIGAJO_4; }

    std::string qlit(const rm_node& nd) {
        float Q[9]; build_q(nd, Q);
        std::string q;
        for (int k = 0; k < 9; ++k) q += (k ? ", " : "") + lit(Q[k]);
        return q;
    }

    std::string new_c() { std::string v = "c" + std::to_string(counter++); cache_decl += "    float " + v + " = 0.f;\n"; return "s." + v; }

    // cell_cache (round 3).  senfina_ripeto takes N = floor(dot(a, p) / dot(a, a) + .5f) lattice vectors off the point
    // (:577-581): a quotient, a floor and their guard, a fifth of chair.sce's step and half of infinity.sce's, and from
    // one evaluation to the next N almost never changes (99.9 % of infinity.sce's, lane by lane).  N is a monotonic
    // function of the rounded dot product d alone (a rounded quotient by a constant, a rounded sum, a floor), so every d
    // with |d - N aa| <= aa (.5 - E) gives the N it is tested against, where E covers the roundings between d and the
    // floor's argument ((|N| + 1) 2^-23) and those of the test itself ((|N| + 1) 2^-24): E = 129 * 2^-21 for |N| <= 127,
    // 2.7 times what is needed.  For a lattice along one axis d is monotonic in that coordinate and the test is on the
    // reduced coordinate itself: |p.z - N a.z| < |a.z| (.5 - E) (the same E covers A p.z, A A and N A rounded).
    // Each lane keeps N a (and N aa) of its last full evaluation (struct SceneCache, a few registers that live across
    // trips and tiles: an entry is valid for any point that passes its test, whatever ray it came from).  A trip takes
    // the reduced point p - N a - the reference's own subtraction, N a being the product it forms - and one wave-wide
    // test; where a marching lane fails it the whole wave takes the reference's arithmetic and refreshes its entries.
    // An N beyond 127 is stored as +infinity, which fails every test.
    static constexpr double kCellMargin = 129.0 / 2097152.0;
    bool cell_cacheable(const rm_node& nd) const {
        if (!allow_cell_cache || nd.type != RM_TYPE_SENFINA_RIPETO) return false;
        int nz = 0;
        for (int k = 0; k < 3; ++k) {
            const float a = std::fabs(nd.f[k]);
            if (!(a == a) || a > 1e6f || (a != 0.f && a < 1e-6f)) return false;
            nz += a != 0.f;
        }
        const float aa = ripeto_aa(nd);
        return nz >= 1 && aa >= 1e-12f && aa <= 1e12f;
    }
    static float round_down(double v) {      // the largest float not above v (v > 0)
        float f = static_cast<float>(v);
        if (static_cast<double>(f) > v) f = std::nextafterf(f, 0.f);
        return f;
    }
    // the chain of lattices that starts at node i (each the only child of the one before), on the point pv
    std::string cached_lattices(size_t i, const std::string& pv) {
        struct Link { size_t node; std::string in, out, dot_args, e, mid; std::string t[3], o[3]; int nz = 0, axis = -1; float thr = 0.f; };
        std::vector<Link> links;
        std::string in = pv;
        for (size_t at = i; cell_cacheable(nodes[at]); at = kid(at, 0)) {
            const rm_node& nd = nodes[at];
            Link l; l.node = at; l.in = in;
            l.out = "P3{";
            for (int k = 0; k < 3; ++k) {
                const char* c = k == 0 ? ".x" : k == 1 ? ".y" : ".z";
                if (nd.f[k] != 0.f) { l.t[k] = new_f(); l.o[k] = new_c(); ++l.nz; l.axis = k; l.out += (k ? ", " : "") + l.t[k]; }
                else l.out += (k ? ", " : "") + in + c;
            }
            l.out += "}";
            auto z = [](float v) { return v == 0.f ? "true" : "false"; };
            l.dot_args = std::string("<") + z(nd.f[0]) + ", " + z(nd.f[1]) + ", " + z(nd.f[2]) + ">(";
            const double aa = static_cast<double>(ripeto_aa(nd));
            if (l.nz == 1) l.thr = round_down(std::fabs(static_cast<double>(nd.f[l.axis])) * (0.5 - kCellMargin));
            else { l.thr = round_down(aa * (0.5 - kCellMargin)); l.mid = new_c(); l.e = "e" + std::to_string(counter++); }
            links.push_back(l);
            in = l.out;
            if (node_arity(nodes[at].type) != 1) break;
        }
        auto dot_of = [&](const Link& l) { return "ripeto_dot" + l.dot_args + l.in + ", " + lit3(nodes[l.node].f) + ")"; };
        // ---- the reduced points by the cells kept, and the test
        std::vector<std::string> ballots;
        std::vector<std::pair<float, std::vector<std::string>>> single;      // one-axis tests by threshold
        for (const Link& l : links) {
            for (int k = 0; k < 3; ++k)
                if (!l.t[k].empty()) dist += "    " + l.t[k] + " = " + l.in + (k == 0 ? ".x" : k == 1 ? ".y" : ".z") + " - " + l.o[k] + ";\n";
            if (l.nz == 1) {
                bool filed = false;
                for (auto& g : single) if (g.first == l.thr && g.second.size() < 3) { g.second.push_back(l.t[l.axis]); filed = true; break; }
                if (!filed) single.push_back({l.thr, {l.t[l.axis]}});
            } else {
                dist += "    const float " + l.e + " = " + dot_of(l) + ";\n";
                ballots.push_back("__builtin_amdgcn_ballot_w64(!(fabsf(" + l.e + " - " + l.mid + ") < " + lit(l.thr) + "))");
            }
        }
        for (const auto& g : single) {
            std::string w = g.second[0];
            if (g.second.size() > 1) {      // the largest magnitude of the group, one instruction (written out: see the combinators)
                w = "w" + std::to_string(counter++);
                dist += "    float " + w + ";\n";
                if (g.second.size() == 2)
                    dist += "    asm(\"v_max_f32_e64 %0, |%1|, |%2|\" : \"=v\"(" + w + ") : \"v\"(" + g.second[0] + "), \"v\"(" + g.second[1] + "));\n";
                else
                    dist += "    asm(\"v_max3_f32 %0, |%1|, |%2|, |%3|\" : \"=v\"(" + w + ") : \"v\"(" + g.second[0] + "), \"v\"(" + g.second[1] +
                            "), \"v\"(" + g.second[2] + "));\n";
            }
            ballots.push_back("__builtin_amdgcn_ballot_w64(!(fabsf(" + w + ") < " + lit(g.first) + "))");
        }
        std::string test;
        for (size_t k = 0; k < ballots.size(); ++k) test += (k ? " | " : "") + ballots[k];
        // ---- some marching lane has left a cell (or holds none): the reference's arithmetic for the wave, cells refreshed
        dist += "    if (!SKIP || ((" + test + ") & amask) != 0) {\n";
        for (size_t j = 0; j < links.size(); ++j) {
            const Link& l = links[j];
            const rm_node& nd = nodes[l.node];
            const float aa = ripeto_aa(nd);
            const std::string id = std::to_string(counter++), n = "n" + id, ok = "ok" + id;
            const std::string d = (j == 0 && l.nz > 1) ? l.e : dot_of(l);
            dist += "        const float " + n + " = floorf(div_const_for_floor(" + d + ", " + lit(aa) + ", " + lit(divisor_reciprocal(exact, aa)) + ") + .5f);\n";
            dist += "        const bool " + ok + " = fabsf(" + n + ") <= 127.f;\n";
            for (int k = 0; k < 3; ++k) {
                if (l.t[k].empty()) continue;
                const std::string o = "o" + std::to_string(counter++);
                dist += "        const float " + o + " = " + n + " * " + lit(nd.f[k]) + ";\n";
                dist += "        " + l.t[k] + " = " + l.in + (k == 0 ? ".x" : k == 1 ? ".y" : ".z") + " - " + o + ";\n";
                dist += "        " + l.o[k] + " = " + ok + " ? " + o + " : __builtin_inff();\n";
            }
            if (l.nz > 1) dist += "        " + l.mid + " = " + ok + " ? " + n + " * " + lit(aa) + " : __builtin_inff();\n";
        }
        dist += "    }\n";
        std::string first;
        for (const Link& l : links) {
            pmemo[std::make_pair(l.node, l.in)] = Temp{l.out, region};
            if (first.empty()) first = l.out;
        }
        return first;
    }

    // the point node i hands to its child, for i a transform (or a portanta_sfero)
    std::string point(size_t i, const std::string& pv) {
        const auto key = std::make_pair(i, pv);
        auto it = pmemo.find(key);
        if (it != pmemo.end() && usable(it->second)) return it->second.var;
        const rm_node& nd = nodes[i];
        if (!in_norm && cell_cacheable(nd)) return cached_lattices(i, pv);
        // A transform that hands a component through unchanged (a movo with a +0 offset: x - (+0) is x, bit for
        // bit; the axis of a rotaciox/y/z) keeps no copy of it: the march carries its temporaries from trip to
        // trip in registers, and a copy there is a v_mov per trip.
        bool same[3] = {false, false, false};
        if (nd.type == RM_TYPE_MOVO) {
            for (int k = 0; k < 3; ++k) { uint32_t bits; std::memcpy(&bits, &nd.f[k], 4); same[k] = bits == 0u; }
        } else if (nd.type == RM_TYPE_ROTACIOX || nd.type == RM_TYPE_ROTACIOY || nd.type == RM_TYPE_ROTACIOZ) {
            same[nd.type == RM_TYPE_ROTACIOX ? 0 : nd.type == RM_TYPE_ROTACIOY ? 1 : 2] = true;
        }
        if (same[0] || same[1] || same[2]) {
            const std::string call = nd.type == RM_TYPE_MOVO ? "movo(" + pv + ", " + lit3(nd.f) + ")"
                                     : std::string(nd.type == RM_TYPE_ROTACIOX ? "rotx(" : nd.type == RM_TYPE_ROTACIOY ? "roty(" : "rotz(") +
                                           pv + ", " + lit(nd.f[0]) + ", " + lit(nd.f[1]) + ")";
            std::string q = "P3{";
            for (int k = 0; k < 3; ++k) {
                const char* c = k == 0 ? ".x" : k == 1 ? ".y" : ".z";
                std::string part = pv + c;
                if (!same[k]) { part = new_f(); dist += "    " + part + " = " + call + c + ";\n"; }
                q += (k ? ", " : "") + part;
            }
            q += "}";
            pmemo[key] = Temp{q, in_norm ? 0 : region};
            return q;
        }
        const std::string q = new_p();
        std::string e;
        switch (nd.type) {
        case RM_TYPE_MOVO: case RM_TYPE_PORTANTA_SFERO: e = "movo(" + pv + ", " + lit3(nd.f) + ")"; break;
        case RM_TYPE_ROTACIOX: e = "rotx(" + pv + ", " + lit(nd.f[0]) + ", " + lit(nd.f[1]) + ")"; break;
        case RM_TYPE_ROTACIOY: e = "roty(" + pv + ", " + lit(nd.f[0]) + ", " + lit(nd.f[1]) + ")"; break;
        case RM_TYPE_ROTACIOZ: e = "rotz(" + pv + ", " + lit(nd.f[0]) + ", " + lit(nd.f[1]) + ")"; break;
        case RM_TYPE_ROTACIOQ: e = "rotq(" + pv + ", " + qlit(nd) + ")"; break;
        default: {                              // senfina_ripeto
            const float aa = ripeto_aa(nd);
            e = ripeto_call(pv, nd.f[0], nd.f[1], nd.f[2], aa, divisor_reciprocal(exact, aa));
        }
        }
        dist += "    " + q + " = " + e + ";\n";
        pmemo[key] = Temp{q, in_norm ? 0 : region};
        return q;
    }

    // operands (node, negated) of the min (want_min) or max chain that node i, negated or not, continues
    void chain(size_t i, bool neg, bool want_min, std::vector<std::pair<size_t, bool>>& ops) const {
        const rm_node& nd = nodes[i];
        const bool comb = nd.type == RM_TYPE_KUNIGAJO_2 || nd.type == RM_TYPE_KUNIGAJO_3 || nd.type == RM_TYPE_KUNIGAJO_4 ||
                          nd.type == RM_TYPE_KOMUNAJO_2 || nd.type == RM_TYPE_KOMUNAJO_3 || nd.type == RM_TYPE_KOMUNAJO_4;
        if (nd.type == RM_TYPE_KOMPLEMENTO) { chain(kid(i, 0), !neg, want_min, ops); return; }
        if (comb && (is_min(nd.type) != neg) == want_min) {
            for (int c = 0; c < node_arity(nd.type); ++c) chain(kid(i, c), neg, want_min, ops);
            return;
        }
        ops.emplace_back(i, neg);
    }

    // An upper bound of -DIST(node i, p) that holds at every point, if the node is a solid under transforms
    // only: -(|q| - r) <= r for a sphere; a box's distance is at least -min(b) (its value at the centre) when
    // no half-extent is negative, a cylinder's at least -min(r, h) likewise.  The rounded arithmetic keeps the
    // bound (|q| >= 0 and rounding is monotonic: RN(|q| - r) >= -r; the box's and the cylinder's terms likewise).
    bool carve_bound(size_t i, float* bound) const {
        for (;;) {
            const rm_node& nd = nodes[i];
            switch (nd.type) {
            case RM_TYPE_MOVO: case RM_TYPE_SENFINA_RIPETO: case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY:
            case RM_TYPE_ROTACIOZ: case RM_TYPE_ROTACIOQ:
                i = kid(i, 0);
                continue;
            case RM_TYPE_SFERO:
                *bound = nd.f[0];
                return nd.f[0] == nd.f[0] && nd.f[0] < 1e30f;
            case RM_TYPE_KUBO: {
                const float m = std::min(nd.f[0], std::min(nd.f[1], nd.f[2]));
                *bound = m;
                return m >= 0.f && m < 1e30f;            // false for NaN
            }
            case RM_TYPE_CILINDRO: {
                const float m = std::min(nd.f[0], nd.f[1]);
                *bound = m;
                return m >= 0.f && m < 1e30f;
            }
            default: return false;
            }
        }
    }

    // a sphere under transforms only (sphere_skip in d() below): the transforms from the top, the leaf, and
    // the least sum of squares that is sure to put the point outside the sphere
    struct SphereLeaf { std::vector<size_t> transforms; size_t leaf = 0; float clear = 0.f; std::string point, sumsq, var; };
    bool sphere_leaf(size_t i, SphereLeaf* out) const {
        for (;;) {
            const rm_node& nd = nodes[i];
            switch (nd.type) {
            case RM_TYPE_MOVO: case RM_TYPE_SENFINA_RIPETO: case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY:
            case RM_TYPE_ROTACIOZ: case RM_TYPE_ROTACIOQ:
                out->transforms.push_back(i);
                i = kid(i, 0);
                continue;
            case RM_TYPE_SFERO: {
                const float r = nd.f[0];
                if (!(r >= 1e-12f && r <= 1e12f)) return false;        // r*r normal and finite; false for NaN
                const double need = static_cast<double>(r) * static_cast<double>(r) * (1.0 + 0x1p-20);
                float c = static_cast<float>(need);
                if (static_cast<double>(c) < need) c = std::nextafterf(c, INFINITY);
                out->leaf = i;
                out->clear = c;
                return true;
            }
            default: return false;
            }
        }
    }

    std::string d(size_t i, const std::string& pv) {
        const auto key = std::make_pair(i, pv);
        auto it = dmemo.find(key);
        if (it != dmemo.end() && usable(it->second)) return it->second.var;
        const rm_node& nd = nodes[i];
        if (i == anchor && !in_norm) below_anchor = true;
        std::string v;
        auto leaf = [&](const std::string& call) { v = new_f(); dist += "    " + v + " = " + call + ";\n"; };
        switch (nd.type) {
        case RM_TYPE_SFERO:    leaf("sfero_dist(" + pv + ", " + lit(nd.f[0]) + ")"); break;
        case RM_TYPE_KUBO:     leaf("kubo_dist(" + pv + ", " + lit3(nd.f) + ")"); break;
        case RM_TYPE_CILINDRO: leaf("cilindro_dist(" + pv + ", " + lit(nd.f[0]) + ", " + lit(nd.f[1]) + ")"); break;
        case RM_TYPE_EBENO:    leaf("ebeno_dist(" + pv + ", " + lit3(nd.f) + ")"); break;
        case RM_TYPE_PORTANTA_SFERO: {
            const std::string q = point(i, pv);
            leaf("length_3(" + q + ".x, " + q + ".y, " + q + ".z) - " + lit(nd.f[3]));
            dist += "    if (" + v + " <= RM_MIN_DIST) " + v + " = " + fn('d', kid(i, 0)) + "(" + q + ");\n";
            break;
        }
        case RM_TYPE_KOMPLEMENTO: { const std::string a = d(kid(i, 0), pv); leaf("-" + a); break; }
        case RM_TYPE_MOVO: case RM_TYPE_SENFINA_RIPETO: case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY:
        case RM_TYPE_ROTACIOZ: case RM_TYPE_ROTACIOQ:
            v = d(kid(i, 0), point(i, pv));
            break;
        case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2: {
            ++glata_depth;
            const std::string a = d(kid(i, 0), pv), b = d(kid(i, 1), pv);
            --glata_depth;
            leaf("glata(" + a + ", " + b + ", " + lit(nd.f[0]) + ", " + lit(divisor_reciprocal(exact, nd.f[0])) + ", " +
                 (nd.type == RM_TYPE_GLATA_KOMUNAJO_2 ? "true" : "false") + ")");
            break;
        }
        default: {                              // kunigajo_N / komunajo_N
            // The whole chain under this node as one list: nested combinators of the same effective
            // kind are merged, a komplemento flips min and max below it and negates the operands
            // (-min(a, b) = max(-a, -b)).  Emitted as v_min3 / v_max3 with the negations in the
            // instruction, written out: through the compiler each operand it cannot prove canonical
            // would cost an extra v_max x, x.
            std::vector<std::pair<size_t, bool>> members;
            const bool mn = is_min(nd.type);
            chain(i, false, mn, members);
            const char* op3 = mn ? "v_min3_f32" : "v_max3_f32";
            const char* op2 = mn ? "v_min_f32_e64" : "v_max_f32_e64";
            v = new_f();
            // folds `ops` (temporary, negated) into v; `acc`: v already holds a running value
            auto fold = [&](const std::vector<std::pair<std::string, bool>>& ops, bool acc_in, const std::string& pad) {
                auto src = [&](size_t k, int arg) { return std::string(ops[k].second ? "-%" : "%") + std::to_string(arg); };
                size_t at = 0;
                std::string acc = acc_in ? v : std::string();          // variable holding the running value, never negated
                while (at < ops.size()) {
                    const size_t left = ops.size() - at;
                    std::string text, ins;
                    if (acc.empty() && left >= 3) {
                        text = std::string(op3) + " %0, " + src(at, 1) + ", " + src(at + 1, 2) + ", " + src(at + 2, 3);
                        ins = "\"v\"(" + ops[at].first + "), \"v\"(" + ops[at + 1].first + "), \"v\"(" + ops[at + 2].first + ")";
                        at += 3;
                    } else if (acc.empty() && left == 2) {
                        text = std::string(op2) + " %0, " + src(at, 1) + ", " + src(at + 1, 2);
                        ins = "\"v\"(" + ops[at].first + "), \"v\"(" + ops[at + 1].first + ")";
                        at += 2;
                    } else if (acc.empty()) {       // a single operand: a copy, negated or not
                        dist += pad + v + " = " + (ops[at].second ? "-" : "") + ops[at].first + ";\n";
                        acc = v;
                        at += 1;
                        continue;
                    } else if (left >= 2) {
                        text = std::string(op3) + " %0, %1, " + src(at, 2) + ", " + src(at + 1, 3);
                        ins = "\"v\"(" + acc + "), \"v\"(" + ops[at].first + "), \"v\"(" + ops[at + 1].first + ")";
                        at += 2;
                    } else {
                        text = std::string(op2) + " %0, %1, " + src(at, 2);
                        ins = "\"v\"(" + acc + "), \"v\"(" + ops[at].first + ")";
                        at += 1;
                    }
                    dist += pad + "asm(\"" + text + "\" : \"=v\"(" + v + ") : " + ins + ");\n";
                    acc = v;
                }
            };
            // bound_skip: the carved-out solids of a komunajo (A minus B: max(A, -B)) are behind a test.  -B never
            // exceeds its static bound (carve_bound), so where the max over the other members already exceeds
            // every such bound for all marching lanes, max() returns that value bit for bit and the solids are
            // not evaluated (chair.sce: a ray that is more than 60 units from the box skips both spheres, two
            // of the three square roots of a step).  A skipped block leaves its temporaries stale.  The normal's
            // pass reads them only for a lane whose root distance this chain decided and which is at a surface:
            // this chain's value is then below RM_MIN_DIST in magnitude, so with a bound of at least RM_MIN_DIST
            // the block ran.  That argument needs the root value to BE this chain's value (through min, max,
            // komplemento and transforms); under a glata it is a blend, so no block is made there.
            std::vector<std::pair<size_t, bool>> open, carved;
            float bound = 0.f;
            if (!mn && allow_skip && region == 0 && glata_depth == 0) {
                for (const auto& m : members) {
                    float b;
                    if (m.second && carve_bound(m.first, &b)) { carved.push_back(m); bound = carved.size() == 1 ? b : std::max(bound, b); }
                    else open.push_back(m);
                }
                if (open.empty() || carved.empty() || !(bound >= RM_MIN_DIST)) { open = members; carved.clear(); }
            } else {
                open = members;
            }
            std::vector<std::pair<std::string, bool>> ops;
            for (const auto& m : open) ops.emplace_back(d(m.first, pv), m.second);
            fold(ops, false, "    ");
            if (!carved.empty()) {
                dist += "    if (!SKIP || (__builtin_amdgcn_ballot_w64(!(" + v + " > " + lit(bound) + ")) & amask) != 0) {\n";
                region = ++regions;
                ops.clear();
                // sphere_skip: where every carved-out solid is a sphere, its square root is behind a second test.
                // With q = x*x + y*y + z*z as sfero_dist forms it, q >= clear (clear: r*r*(1 + 2^-20) rounded up)
                // puts sqrt(q) two units in the last place above r or more, so the rounded root exceeds r and
                // -(sqrt(q) - r) < 0.  Where that holds for every carved sphere and the other members' max is
                // >= 0 (all marching lanes), max() returns that max bit for bit (a -0 included: the spheres' terms
                // are below zero, not zeros).  Unlike the outer test this one passes AT the surface of the other
                // members, where the normal's pass reads the spheres' distances to choose its branch: they are
                // set to +infinity, which loses every comparison the true values (negated: below zero, below
                // the winning member) lose, and ties with nothing.
                std::vector<SphereLeaf> balls;
                bool all_balls = allow_sphere_skip;
                for (const auto& m : carved) {
                    SphereLeaf b;
                    if (!all_balls || !sphere_leaf(m.first, &b)) { all_balls = false; break; }
                    balls.push_back(b);
                }
                if (all_balls) {
                    std::string test = "__builtin_amdgcn_ballot_w64(!(" + v + " >= 0.f))";      // a ballot each: straight from the compares
                    for (auto& b : balls) {
                        std::string q = pv;
                        for (size_t t : b.transforms) q = point(t, q);
                        b.point = q;
                        b.sumsq = "q" + std::to_string(counter++);
                        dist += "        const float " + b.sumsq + " = " + q + ".x * " + q + ".x + " + q + ".y * " + q + ".y + " + q + ".z * " + q + ".z;\n";
                        test += " | __builtin_amdgcn_ballot_w64(!(" + b.sumsq + " >= " + lit(b.clear) + "))";
                    }
                    for (auto& b : balls) b.var = new_f();
                    dist += "        if (!SKIP || ((" + test + ") & amask) != 0) {\n";
                    for (size_t k = 0; k < balls.size(); ++k) {
                        const SphereLeaf& b = balls[k];
                        dist += "            " + b.var + " = sqrt_exact(" + b.sumsq + ") - " + lit(nodes[b.leaf].f[0]) + ";\n";
                        dmemo[std::make_pair(b.leaf, b.point)] = Temp{b.var, region};
                        ops.emplace_back(b.var, carved[k].second);
                    }
                    fold(ops, true, "            ");
                    dist += "        } else {\n";
                    for (const auto& b : balls) dist += "            " + b.var + " = __builtin_inff();\n";
                    dist += "        }\n";
                } else {
                    for (const auto& m : carved) ops.emplace_back(d(m.first, pv), m.second);
                    fold(ops, true, "        ");
                }
                region = 0;
                dist += "    }\n";
            }
        }
        }
        dmemo[key] = Temp{v, in_norm ? 0 : region};
        return v;
    }

    // DIST(node i, pv) for the normal's selections: the variable of the distance pass where it made
    // one, else (a combinator or komplemento the pass merged into its parent's chain) an expression
    // over its children, evaluated only by the lanes that take a normal
    std::string value(size_t i, const std::string& pv) {
        auto it = dmemo.find(std::make_pair(i, pv));
        if (it != dmemo.end() && usable(it->second)) return it->second.var;
        const rm_node& nd = nodes[i];
        if (nd.type == RM_TYPE_KOMPLEMENTO) return "(-" + value(kid(i, 0), pv) + ")";
        const int na = node_arity(nd.type);
        const bool comb = nd.type == RM_TYPE_KUNIGAJO_2 || nd.type == RM_TYPE_KUNIGAJO_3 || nd.type == RM_TYPE_KUNIGAJO_4 ||
                          nd.type == RM_TYPE_KOMUNAJO_2 || nd.type == RM_TYPE_KOMUNAJO_3 || nd.type == RM_TYPE_KOMUNAJO_4;
        if (!comb) return d(i, pv);
        std::string e = value(kid(i, 0), pv);
        for (int c = 1; c < na; ++c) e = std::string(is_min(nd.type) ? "min2(" : "max2(") + e + ", " + value(kid(i, c), pv) + ")";
        return e;
    }

    // statements that leave NORM(node i, pv) in `out`
    void n(size_t i, const std::string& pv, const std::string& out, int depth) {
        const std::string pad(static_cast<size_t>(depth) * 4, ' ');
        const rm_node& nd = nodes[i];
        if (lean && i == anchor) { norm += pad + out + " = " + fn('n', i) + "(" + pv + ");\n"; return; }
        switch (nd.type) {
        case RM_TYPE_SFERO:    norm += pad + out + " = " + pv + ";\n"; break;
        case RM_TYPE_KUBO:     norm += pad + out + " = kubo_norm(" + pv + ", " + lit3(nd.f) + ");\n"; break;
        case RM_TYPE_CILINDRO: norm += pad + out + " = cilindro_norm(" + pv + ", " + lit(nd.f[0]) + ", " + lit(nd.f[1]) + ");\n"; break;
        case RM_TYPE_EBENO:    norm += pad + out + " = P3{" + lit3(nd.f) + "};\n"; break;
        case RM_TYPE_PORTANTA_SFERO:
            norm += pad + out + " = " + fn('n', kid(i, 0)) + "(" + point(i, pv) + ");\n";
            break;
        case RM_TYPE_KOMPLEMENTO:
            n(kid(i, 0), pv, out, depth);
            norm += pad + out + " = neg3(" + out + ");\n";
            break;
        case RM_TYPE_MOVO: case RM_TYPE_SENFINA_RIPETO:
            n(kid(i, 0), point(i, pv), out, depth);
            break;
        case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY: case RM_TYPE_ROTACIOZ: {
            const char* ax = nd.type == RM_TYPE_ROTACIOX ? "x" : nd.type == RM_TYPE_ROTACIOY ? "y" : "z";
            n(kid(i, 0), point(i, pv), out, depth);
            norm += pad + out + " = rot" + ax + "_back(" + out + ", " + lit(nd.f[0]) + ", " + lit(nd.f[1]) + ");\n";
            break;
        }
        case RM_TYPE_ROTACIOQ:
            n(kid(i, 0), point(i, pv), out, depth);
            norm += pad + out + " = rotq_back(" + out + ", " + qlit(nd) + ");\n";
            break;
        case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2: {
            const std::string a = value(kid(i, 0), pv), b = value(kid(i, 1), pv), id = std::to_string(counter++);
            const std::string h = "h" + id, na = "na" + id, nb = "nb" + id;
            norm += pad + "const float " + h + " = glata_h(" + a + ", " + b + ", " + lit(nd.f[0]) + ", " +
                    lit(divisor_reciprocal(exact, nd.f[0])) + ", " + (nd.type == RM_TYPE_GLATA_KOMUNAJO_2 ? "true" : "false") + ");\n";
            norm += pad + "if (" + h + " > 1.f) {\n";
            n(kid(i, 0), pv, out, depth + 1);
            norm += pad + "} else if (" + h + " < 0.f) {\n";
            n(kid(i, 1), pv, out, depth + 1);
            norm += pad + "} else {\n" + pad + "    P3 " + na + ", " + nb + ";\n";
            n(kid(i, 0), pv, na, depth + 1);
            n(kid(i, 1), pv, nb, depth + 1);
            norm += pad + "    " + out + " = glata_blend(" + na + ", " + nb + ", " + h + ");\n" + pad + "}\n";
            break;
        }
        default: {                              // kunigajo_N / komunajo_N
            const int na = node_arity(nd.type);
            const bool mn = is_min(nd.type);
            std::vector<std::string> a;
            for (int c = 0; c < na; ++c) a.push_back(value(kid(i, c), pv));
            if (na == 2) {                      // ties go to child 1 (:153, :252)
                norm += pad + "if (" + a[0] + (mn ? " < " : " > ") + a[1] + ") {\n";
                n(kid(i, 0), pv, out, depth + 1);
                norm += pad + "} else {\n";
                n(kid(i, 1), pv, out, depth + 1);
                norm += pad + "}\n";
            } else {                            // ties go to the earlier child (:179,186 ... :316,323,330)
                const std::string id = std::to_string(counter++), best = "best" + id, pick = "pick" + id;
                norm += pad + "float " + best + " = " + a[0] + "; int " + pick + " = 0;\n";
                for (int c = 1; c < na; ++c)
                    norm += pad + "if (" + best + (mn ? " > " : " < ") + a[static_cast<size_t>(c)] + ") { " + best + " = " +
                            a[static_cast<size_t>(c)] + "; " + pick + " = " + std::to_string(c) + "; }\n";
                for (int c = 0; c < na; ++c) {
                    norm += pad + (c == 0 ? "if (" + pick + " == 0) {\n"
                                   : c + 1 < na ? "} else if (" + pick + " == " + std::to_string(c) + ") {\n" : "} else {\n");
                    n(kid(i, c), pv, out, depth + 1);
                }
                norm += pad + "}\n";
            }
        }
        }
    }

    std::string source(size_t count, const unsigned char* seen) {
        weigh_lattices(count, seen);
        anchor = 0;
        for (;;) {
            const uint32_t t = nodes[anchor].type;
            if (t == RM_TYPE_MOVO || t == RM_TYPE_SENFINA_RIPETO || t == RM_TYPE_ROTACIOX || t == RM_TYPE_ROTACIOY ||
                t == RM_TYPE_ROTACIOZ || t == RM_TYPE_ROTACIOQ) anchor = kid(anchor, 0);
            else break;
        }
        const std::string root = d(0, "E");
        in_norm = true;
        n(0, "E", "N", 1);
        // SKIP: the blocks of bound_skip may be skipped (amask: the lanes whose distance matters).  The march asks
        // with SKIP = true; a caller that takes normals at points of its own choosing (known-answer hook) with false.
        return "struct SceneKeep {\n" + decl + "};\n"
               "struct SceneCache {\n" + cache_decl + "};\n"
               "template <bool SKIP>\n"
               "__device__ __forceinline__ float scene_dist_keep(P3 E, SceneKeep& k, SceneCache& s, unsigned long long amask) {\n" + locals + dist +
               "    (void)amask; (void)s;\n    return " + root + ";\n}\n"
               "__device__ __forceinline__ P3 scene_norm_keep(P3 E, const SceneKeep& k) {\n    P3 N{0.f, 0.f, 0.f};\n" + norm +
               "    (void)E; (void)k;\n    return N;\n}\n";
    }
};

}  // namespace

namespace { std::string program_function(const std::vector<Slot>& program, uint32_t value_slots); }

std::string generate_scene_source(const rm_node* nodes, size_t n, const unsigned char* seen, const DivisorSet* exact,
                                  const std::vector<Slot>* program, uint32_t value_slots) {
    std::string s = "namespace rm {\n";
    const std::string FD = "__device__ __forceinline__ float ", FN = "__device__ __forceinline__ P3 ";
    // children carry higher indices than their parents (links point forward), so
    // emitting from the last node down defines every callee before its callers
    for (size_t i = n; i-- > 0;) {
        if (!seen[i]) continue;
        const rm_node& nd = nodes[i];
        auto kid = [&](int k) { return i + static_cast<size_t>(nd.o[k]); };
        const std::string d = FD + fn('d', i) + "(P3 p) { ", nn = FN + fn('n', i) + "(P3 p) { ";
        s += "// node " + std::to_string(i) + "\n";
        switch (nd.type) {
        case RM_TYPE_SFERO:
            s += d + "return sfero_dist(p, " + lit(nd.f[0]) + "); }\n";
            s += nn + "return p; }\n";
            break;
        case RM_TYPE_KUBO:
            s += d + "return kubo_dist(p, " + lit3(nd.f) + "); }\n";
            s += nn + "return kubo_norm(p, " + lit3(nd.f) + "); }\n";
            break;
        case RM_TYPE_CILINDRO:
            s += d + "return cilindro_dist(p, " + lit(nd.f[0]) + ", " + lit(nd.f[1]) + "); }\n";
            s += nn + "return cilindro_norm(p, " + lit(nd.f[0]) + ", " + lit(nd.f[1]) + "); }\n";
            break;
        case RM_TYPE_EBENO:
            s += d + "return ebeno_dist(p, " + lit3(nd.f) + "); }\n";
            s += nn + "return P3{" + lit3(nd.f) + "}; }\n";
            break;
        case RM_TYPE_PORTANTA_SFERO:
            s += d + "P3 q = movo(p, " + lit3(nd.f) + "); float e = length_3(q.x, q.y, q.z) - " + lit(nd.f[3]) +
                 "; if (e <= RM_MIN_DIST) return " + fn('d', kid(0)) + "(q); return e; }\n";
            s += nn + "return " + fn('n', kid(0)) + "(movo(p, " + lit3(nd.f) + ")); }\n";
            break;
        case RM_TYPE_KOMPLEMENTO:
            s += d + "return -" + fn('d', kid(0)) + "(p); }\n";
            s += nn + "return neg3(" + fn('n', kid(0)) + "(p)); }\n";
            break;
        case RM_TYPE_MOVO:
            s += d + "return " + fn('d', kid(0)) + "(movo(p, " + lit3(nd.f) + ")); }\n";
            s += nn + "return " + fn('n', kid(0)) + "(movo(p, " + lit3(nd.f) + ")); }\n";
            break;
        case RM_TYPE_SENFINA_RIPETO: {
            const float aa = ripeto_aa(nd);
            const std::string t = ripeto_call("p", nd.f[0], nd.f[1], nd.f[2], aa, divisor_reciprocal(exact, aa));
            s += d + "return " + fn('d', kid(0)) + "(" + t + "); }\n";
            s += nn + "return " + fn('n', kid(0)) + "(" + t + "); }\n";
            break;
        }
        case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY: case RM_TYPE_ROTACIOZ: {
            const char* ax = nd.type == RM_TYPE_ROTACIOX ? "x" : nd.type == RM_TYPE_ROTACIOY ? "y" : "z";
            const std::string cs = lit(nd.f[0]) + ", " + lit(nd.f[1]);
            s += d + "return " + fn('d', kid(0)) + "(rot" + ax + "(p, " + cs + ")); }\n";
            s += nn + "return rot" + ax + "_back(" + fn('n', kid(0)) + "(rot" + ax + "(p, " + cs + ")), " + cs + "); }\n";
            break;
        }
        case RM_TYPE_ROTACIOQ: {
            float Q[9]; build_q(nd, Q);
            std::string q;
            for (int k = 0; k < 9; ++k) q += (k ? ", " : "") + lit(Q[k]);
            s += d + "return " + fn('d', kid(0)) + "(rotq(p, " + q + ")); }\n";
            s += nn + "return rotq_back(" + fn('n', kid(0)) + "(rotq(p, " + q + ")), " + q + "); }\n";
            break;
        }
        case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2: {
            const std::string kom = nd.type == RM_TYPE_GLATA_KOMUNAJO_2 ? "true" : "false",
                              k = lit(nd.f[0]) + ", " + lit(divisor_reciprocal(exact, nd.f[0]));
            const std::string d0 = fn('d', kid(0)) + "(p)", d1 = fn('d', kid(1)) + "(p)";
            s += d + "return glata(" + d0 + ", " + d1 + ", " + k + ", " + kom + "); }\n";
            s += nn + "float a = " + d0 + ", b = " + d1 + "; float h = glata_h(a, b, " + k + ", " + kom + "); " +
                 "if (h > 1.f) return " + fn('n', kid(0)) + "(p); if (h < 0.f) return " + fn('n', kid(1)) + "(p); " +
                 "return glata_blend(" + fn('n', kid(0)) + "(p), " + fn('n', kid(1)) + "(p), h); }\n";
            break;
        }
        default: {      // kunigajo_N / komunajo_N
            const int na = node_arity(nd.type);
            const bool is_min = nd.type == RM_TYPE_KUNIGAJO_2 || nd.type == RM_TYPE_KUNIGAJO_3 || nd.type == RM_TYPE_KUNIGAJO_4;
            const char* comb = is_min ? "min2" : "max2";
            s += d + "float v = " + fn('d', kid(0)) + "(p); ";
            for (int k = 1; k < na; ++k) s += std::string("v = ") + comb + "(v, " + fn('d', kid(k)) + "(p)); ";
            s += "return v; }\n";
            if (na == 2) {      // ties go to child 1 (:153, :252)
                s += nn + "float a = " + fn('d', kid(0)) + "(p), b = " + fn('d', kid(1)) + "(p); if (a " +
                     (is_min ? "<" : ">") + " b) return " + fn('n', kid(0)) + "(p); return " + fn('n', kid(1)) + "(p); }\n";
            } else {            // ties go to the earlier child (:179,186 ... :316,323,330)
                s += nn + "float best = " + fn('d', kid(0)) + "(p), v; int pick = 0; ";
                for (int k = 1; k < na; ++k)
                    s += "v = " + fn('d', kid(k)) + "(p); if (best " + (is_min ? ">" : "<") + " v) { best = v; pick = " +
                         std::to_string(k) + "; } ";
                for (int k = 0; k + 1 < na; ++k) s += "if (pick == " + std::to_string(k) + ") return " + fn('n', kid(k)) + "(p); ";
                s += "return " + fn('n', kid(na - 1)) + "(p); }\n";
            }
            break;
        }
        }
    }
    // A scene whose distance program holds bounding-sphere tests (scene_program.h, OP_CULL) marches on
    // the compiled program and takes normals from n0; every other one uses the fused pair above.
    bool culls = false;
    if (program)
        for (const Slot& sl : *program) culls = culls || (sl.w[0] & 0xff) == OP_CULL;
    if (culls) {
        s += program_function(*program, value_slots);
        s += R"(
struct JitScene {
    static constexpr bool kNarrow = false;
    struct Scratch {};
    struct Keep {};
    __device__ __forceinline__ void init(const RenderParams&) {}
    __device__ __forceinline__ float dist_keep(P3 p, Scratch&, unsigned long long amask, Keep&) { return scene_program(p, amask); }
    __device__ __forceinline__ float dist_keep_all(P3 p, Scratch& s, unsigned long long amask, Keep& k) { return dist_keep(p, s, amask, k); }
    __device__ __forceinline__ P3 norm_keep(P3 p, Scratch&, const Keep&) { return n0(p); }
};
}  // namespace rm

)";
    } else {
        Fused plain(nodes, exact);
        std::string body = plain.source(n, seen);
        int want_lean = plain.below_floats >= Fused::kLeanFloats;
        if (const char* env = std::getenv("RM_LEAN_KEEP")) want_lean = std::atoi(env);      // experiments only
        if (want_lean) {
            Fused fused(nodes, exact);
            fused.lean = true;
            body = fused.source(n, seen);
        }
        s += body;
        s += R"(
struct JitScene {
    typedef SceneCache Scratch;
    typedef SceneKeep Keep;
    static constexpr bool kNarrow = true;        // the march loops run with the marching lanes as their execution mask (device_render.h)
    __device__ __forceinline__ void init(const RenderParams&) {}
    __device__ __forceinline__ float dist_keep(P3 p, Scratch& s, unsigned long long amask, Keep& k) { return scene_dist_keep<true>(p, k, s, amask); }
    __device__ __forceinline__ float dist_keep_all(P3 p, Scratch& s, unsigned long long, Keep& k) { return scene_dist_keep<false>(p, k, s, ~0ull); }
    __device__ __forceinline__ P3 norm_keep(P3 p, Scratch&, const Keep& k) { return scene_norm_keep(p, k); }
};
}  // namespace rm

)";
    }
    s += kJitKernels;
    return s;
}

// =============================================================================
// Large scenes: the distance PROGRAM (scene_program.h) turned into straight-line
// code, slot by slot, with the same helpers and the same combine (v_med3_f32) as
// the interpreter, so it returns the interpreter's bits.  Skip ranges
// (OP_PORTANTA, OP_CULL) become wave-uniform branches; LDS value slots become
// locals.  Normals stay on the tree walker over the LDS tree table: a normal
// function per node would inline every sub-tree's distance code at every level.
// =============================================================================
namespace {

std::string slot_f(const Slot& sl, int k) { float v; std::memcpy(&v, &sl.w[k], 4); return lit(v); }

// ranks != nullptr: selection mode (generate_program_source, scene_norm): a primitive slot also
// records itself in `best` if it lowers A, or equals it and outranks the holder
// a_inf: A is known to hold +inf here (function entry, after a push): the first member of a min
// chain is then assigned, not combined (med3(+inf, v, -inf) is v; the compiler keeps a v_max for it).
void emit_range(const std::vector<Slot>& prog, size_t lo, size_t hi, std::string& s, int indent, bool& a_inf,
                const std::vector<int32_t>* slot_node = nullptr, const std::vector<uint32_t>* ranks = nullptr) {
    const std::string pad(static_cast<size_t>(indent) * 4, ' ');
    for (size_t pc = lo; pc < hi; ++pc) {
        const Slot& sl = prog[pc];
        const uint32_t w0 = sl.w[0], op = w0 & 0xff, slot = w0 >> F_SLOT_SHIFT;
        const std::string S = (w0 & F_SRC_F) ? "F" : "P";
        const std::string sel = (w0 & (COMB_MAX << F_COMB_SHIFT)) ? "as_f(0x7f800000u)" : "as_f(0xff800000u)";
        const std::string neg = (w0 & F_NEG) ? "-" : "";
        const std::string f1 = slot_f(sl, 1), f2 = slot_f(sl, 2), f3 = slot_f(sl, 3), f4 = slot_f(sl, 4),
                          f5 = slot_f(sl, 5), f6 = slot_f(sl, 6);
        const std::string V = "s" + std::to_string(slot);
        const std::string to_f = (w0 & F_DST_F) ? " F = P;" : "";
        auto prim = [&](const std::string& call) {
            if (!ranks) {
                const bool is_min = !(w0 & (COMB_MAX << F_COMB_SHIFT));
                if (a_inf && is_min) s += pad + "A = " + neg + call + ";\n";
                else s += pad + "A = __builtin_amdgcn_fmed3f(A, " + neg + call + ", " + sel + ");\n";
                a_inf = false;
                return;
            }
            const std::string rk = std::to_string((*ranks)[static_cast<size_t>((*slot_node)[pc])]) + "u";
            // the two constants go through an opaque asm: the compiler would otherwise keep every
            // member's pair in registers of their own for the whole kernel (190 VGPRs on config 4)
            s += pad + "{ const float v = " + neg + call + "; const bool take = v < A || (v == A && " + rk + " > brank); " +
                 "uint32_t id = " + std::to_string(pc) + "u, rk = " + rk + "; asm volatile(\"\" : \"+v\"(id), \"+v\"(rk)); " +
                 "A = take ? v : A; best = take ? id : best; brank = take ? rk : brank; }\n";
        };
        const std::string moved = "movo(" + S + ", " + f1 + ", " + f2 + ", " + f3 + ")";
        switch (op) {
        case OP_END: return;
        case OP_MOVO:    s += pad + "P = " + moved + ";" + to_f + "\n"; break;
        case OP_ROTX:    s += pad + "P = rotx(" + S + ", " + f1 + ", " + f2 + ");" + to_f + "\n"; break;
        case OP_ROTY:    s += pad + "P = roty(" + S + ", " + f1 + ", " + f2 + ");" + to_f + "\n"; break;
        case OP_ROTZ:    s += pad + "P = rotz(" + S + ", " + f1 + ", " + f2 + ");" + to_f + "\n"; break;
        case OP_ROTQ_XY:
            s += pad + "tx = " + S + ".x + 2.f * (" + f1 + " * " + S + ".x + " + f2 + " * " + S + ".y + " + f3 + " * " + S + ".z);\n";
            s += pad + "ty = " + S + ".y + 2.f * (" + f4 + " * " + S + ".x + " + f5 + " * " + S + ".y + " + f6 + " * " + S + ".z);\n";
            break;
        case OP_ROTQ_Z:
            s += pad + "P = P3{tx, ty, " + S + ".z + 2.f * (" + f1 + " * " + S + ".x + " + f2 + " * " + S + ".y + " + f3 + " * " + S + ".z)};" + to_f + "\n";
            break;
        case OP_RIPETO: {
            float a[5];
            std::memcpy(a, &sl.w[1], sizeof a);
            s += pad + "P = " + ripeto_call(S, a[0], a[1], a[2], a[3], a[4]) + ";" + to_f + "\n";
            break;
        }
        case OP_SETF:    s += pad + "F = P;\n"; break;
        case OP_SFERO:   prim("sfero_dist(" + moved + ", " + f4 + ")"); break;
        case OP_KUBO:    prim("kubo_dist(" + moved + ", " + f4 + ", " + f5 + ", " + f6 + ")"); break;
        case OP_CILINDRO:prim("cilindro_dist(" + moved + ", " + f4 + ", " + f5 + ")"); break;
        case OP_EBENO:   prim("ebeno_dist(" + moved + ", " + f4 + ", " + f5 + ", " + f6 + ")"); break;
        case OP_PUSHA_R: s += pad + "B = A; A = __builtin_inff();\n"; a_inf = true; break;
        case OP_POPC_R:  s += pad + "A = __builtin_amdgcn_fmed3f(B, A, " + sel + ");\n"; a_inf = false; break;
        case OP_PUSHA:   s += pad + V + " = A; A = __builtin_inff();\n"; a_inf = true; break;
        case OP_POPC:    s += pad + "A = __builtin_amdgcn_fmed3f(" + V + ", A, " + sel + ");\n"; a_inf = false; break;
        case OP_NEG:     s += pad + "A = -A;\n"; a_inf = false; break;
        case OP_GLATA_R: s += pad + "A = glata(B, A, " + f1 + ", " + f2 + ", " + ((w0 & F_NEG) ? "true" : "false") + ");\n"; a_inf = false; break;
        case OP_GLATA:   s += pad + "A = glata(" + V + ", A, " + f1 + ", " + f2 + ", " + ((w0 & F_NEG) ? "true" : "false") + ");\n"; a_inf = false; break;
        case OP_PUSHF:
            s += pad + V + " = F.x; s" + std::to_string(slot + 1) + " = F.y; s" + std::to_string(slot + 2) + " = F.z; F = P;\n";
            break;
        case OP_POPF:
            s += pad + "F = P3{" + V + ", s" + std::to_string(slot + 1) + ", s" + std::to_string(slot + 2) + "};\n";
            break;
        case OP_PORTANTA: {            // Source/cuda_kernels.cu:53-61
            const size_t skip = sl.w[5];
            s += pad + "P = " + moved + "; " + V + " = length_3(P.x, P.y, P.z) - " + f4 + ";\n";
            s += pad + "if (__builtin_amdgcn_ballot_w64(" + V + " <= RM_MIN_DIST) != 0) {\n";
            emit_range(prog, pc + 1, pc + 1 + skip, s, indent + 1, a_inf, slot_node, ranks);
            s += pad + "}\n";
            pc += skip;
            a_inf = false;
            break;
        }
        case OP_PORTANTA_END: s += pad + "A = (" + V + " <= RM_MIN_DIST) ? A : " + V + ";\n"; a_inf = false; break;
        case OP_CULL: {                // scene_program.h
            const size_t skip = sl.w[5];
            s += pad + "if (!cull_all(" + S + ", " + ((w0 & F_NEG) ? "-A" : "A") + ", amask, " + f1 + ", " + f2 + ", " + f3 + ", " + f4 + ")) {\n";
            emit_range(prog, pc + 1, pc + 1 + skip, s, indent + 1, a_inf, slot_node, ranks);
            s += pad + "}\n";
            pc += skip;
            a_inf = false;          // the range may or may not have run
            break;
        }
        default: break;
        }
    }
}

// [lo, hi) slot ranges of the groups a union's program splits into a few levels down its
// bounding-sphere hierarchy (a range is one OP_CULL with everything it guards, or one primitive),
// and where the transforms in front of the union end.  Empty if the program is not of that form.
struct UnionGroups { size_t prefix_end = 0; std::vector<std::pair<size_t, size_t>> ranges; };
UnionGroups union_groups(const std::vector<Slot>& prog) {
    UnionGroups u;
    auto op_of = [&](size_t pc) { return prog[pc].w[0] & 0xff; };
    auto is_member = [&](size_t pc) { const uint32_t op = op_of(pc); return op == OP_CULL || (op >= OP_SFERO && op <= OP_EBENO); };
    size_t end = 0;
    while (end < prog.size() && op_of(end) != OP_END) ++end;
    while (u.prefix_end < end && !is_member(u.prefix_end)) ++u.prefix_end;
    for (size_t pc = u.prefix_end; pc < end; ++pc) if (!is_member(pc)) return UnionGroups();     // something else than one union
    // split the widest range until there are enough groups (or nothing left to split)
    std::vector<std::pair<size_t, size_t>> r;
    auto items = [&](size_t lo, size_t hi, std::vector<std::pair<size_t, size_t>>& out) {
        for (size_t pc = lo; pc < hi;) {
            const size_t len = op_of(pc) == OP_CULL ? 1 + prog[pc].w[5] : 1;
            out.emplace_back(pc, pc + len);
            pc += len;
        }
    };
    items(u.prefix_end, end, r);
    size_t want = 12;
    if (const char* env = std::getenv("RM_UNION_GROUPS"))        // experiments only
        if (std::atoi(env) > 0) want = static_cast<size_t>(std::atoi(env));
    while (r.size() < want) {
        size_t widest = r.size(), width = 4;                 // a group of fewer than 4 slots is not worth opening
        for (size_t k = 0; k < r.size(); ++k)
            if (op_of(r[k].first) == OP_CULL && r[k].second - r[k].first > width) { widest = k; width = r[k].second - r[k].first; }
        if (widest == r.size()) break;
        std::vector<std::pair<size_t, size_t>> inner;
        items(r[widest].first + 1, r[widest].second, inner);
        r.erase(r.begin() + static_cast<long>(widest));
        r.insert(r.begin() + static_cast<long>(widest), inner.begin(), inner.end());
    }
    if (r.size() >= 4) u.ranges = r;
    return u;
}

// A union of many primitives, its groups visited in an order that changes from call to call: the
// group that held the minimum for the wave's first marching lane last time goes first, the others
// follow in program order.  A ray moves little between two evaluations, so the first group leaves the
// running minimum small and the bounding-sphere tests of most others pass at once (simulated on
// config 4: 56 -> 19 primitives and 69 -> 35 tests per evaluation).  min() over the same members in
// another order: the same value.  The groups are the cases of one switch inside a loop, so each is
// compiled once; `first` lives in the scene's scratch (wave-uniform).
// select: the normal's pass (scene_norm), which also tracks the winning member.
std::string grouped_function(const std::vector<Slot>& program, const UnionGroups& u, bool select,
                             const std::vector<int32_t>* slot_node, const std::vector<uint32_t>* ranks) {
    std::string s = select ? "__device__ __forceinline__ uint32_t scene_pick(P3 E, uint32_t first, P3& F, float& Aout) {\n"
                           : "__device__ __forceinline__ float scene_program(P3 E, unsigned long long amask, uint32_t& first) {\n";
    if (select) s += "    const bool active = true;     // called by the lanes that need a normal: the others are masked off\n    F = E;\n";
    else s += "    P3 F = E;\n";
    s += "    P3 P = E; float A = __builtin_inff(), tx = 0.f, ty = 0.f;\n";
    if (select) s += "    const unsigned long long amask = __builtin_amdgcn_ballot_w64(active);\n";
    if (select) s += "    uint32_t best = 0u, brank = 0u;\n";
    else s += "    uint32_t bestg = first;\n";
    bool a_inf = false;
    emit_range(program, 0, u.prefix_end, s, 1, a_inf, slot_node, ranks);
    const std::string G = std::to_string(u.ranges.size());
    s += "#pragma clang loop unroll(disable)\n";       // one copy of every group, reached through the switch
    s += "    for (uint32_t i = 0; i < " + G + "u; ++i) {\n";
    s += "        const uint32_t g = i == 0 ? first : (i <= first ? i - 1 : i);\n";
    // every member's distance depends on the point alone, so the compiler would compute all of them in
    // front of the loop and keep them in registers: the point is made opaque once per trip
    s += "        P3 Fi = F;\n        asm volatile(\"\" : \"+v\"(Fi.x), \"+v\"(Fi.y), \"+v\"(Fi.z));\n";
    if (!select) s += "        const float before = A;\n";
    s += "        {\n        const P3 F = Fi;\n";
    s += "        switch (g) {\n";
    for (size_t k = 0; k < u.ranges.size(); ++k) {
        s += "        case " + std::to_string(k) + "u: {\n";
        a_inf = false;
        emit_range(program, u.ranges[k].first, u.ranges[k].second, s, 3, a_inf, slot_node, ranks);
        s += "        } break;\n";
    }
    s += "        default: break;\n        }\n        }\n";
    if (!select) s += "        bestg = A < before ? g : bestg;\n";
    s += "    }\n";
    if (select) {
        s += "    (void)tx; (void)ty; (void)P; (void)brank;\n    Aout = A;\n    return best;\n}\n";
    } else {
        s += "    // the next call starts with the group that holds the minimum of the first marching lane\n";
        s += "    // (amask all ones: the caller runs with the marching lanes as its execution mask)\n";
        s += "    if (amask == ~0ull) first = (uint32_t)__builtin_amdgcn_readfirstlane((int)bestg);\n";
        s += "    else if (amask) first = (uint32_t)__builtin_amdgcn_readlane((int)bestg, (int)__builtin_ctzll(amask));\n";
        s += "    (void)tx; (void)ty; (void)P;\n    return A;\n}\n";
    }
    return s;
}

std::string program_function(const std::vector<Slot>& program, uint32_t value_slots) {
    std::string s = "__device__ __forceinline__ float scene_program(P3 E, unsigned long long amask) {\n";
    s += "    P3 F = E, P = E; float A = __builtin_inff(), B = 0.f, tx = 0.f, ty = 0.f;\n";
    if (value_slots) {
        s += "    float";
        for (uint32_t k = 0; k < value_slots; ++k) s += std::string(k ? "," : "") + " s" + std::to_string(k) + " = 0.f";
        s += ";\n";
    }
    bool a_inf = true;
    emit_range(program, 0, program.size(), s, 1, a_inf);
    s += "    (void)B; (void)tx; (void)ty; (void)F; (void)P; (void)amask;\n    return A;\n}\n";
    return s;
}

}  // namespace

std::string generate_program_source(const std::vector<Slot>& program, uint32_t value_slots,
                                    const std::vector<int32_t>& slot_node, const std::vector<uint32_t>& union_rank) {
    const UnionGroups groups = union_rank.empty() ? UnionGroups() : union_groups(program);
    const bool grouped = !groups.ranges.empty();
    std::string s = "namespace rm {\n";
    s += grouped ? grouped_function(program, groups, false, nullptr, nullptr) : program_function(program, value_slots);

    if (union_rank.empty()) {
        // normals by the tree walker over the LDS tree table: a normal function per node would
        // inline every sub-tree's distance code at every level
        s += R"(
struct JitScene {
    static constexpr bool kNarrow = false;
    const TreeNode* tree;
    typedef WalkStack Scratch;
    __device__ __forceinline__ void init(const RenderParams& prm) {
        extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
        tree = reinterpret_cast<const TreeNode*>(lds);
        stage_tree(prm.tree, prm.n_nodes, lds);
    }
    struct Keep {};
    __device__ __forceinline__ float dist_keep(P3 p, Scratch&, unsigned long long amask, Keep&) { return scene_program(p, amask); }
    __device__ __forceinline__ float dist_keep_all(P3 p, Scratch& s, unsigned long long amask, Keep& k) { return dist_keep(p, s, amask, k); }
    __device__ __forceinline__ P3 norm_keep(P3 p, Scratch& st, const Keep&) { return walk_norm(tree, 0, p, st); }
};
}  // namespace rm
)";
        s += kJitKernels;
        return s;
    }

    // The scene is transforms over one union of primitives (SceneBuild::union_rank): its normal is
    // the normal of the member with the smallest distance, equal distances resolved as the
    // reference's nested kunigajo_N do.  One pass over the members, culled like the distance
    // program (a culled group is farther than A by a margin: it holds neither the minimum nor a
    // tie), then the winner's own normal, carried back through the rotations above the union.
    if (grouped) {
        s += grouped_function(program, groups, true, &slot_node, &union_rank);
        s += "__device__ __forceinline__ P3 scene_norm(P3 E, uint32_t first) {\n";
        s += "    P3 F; float A;\n    const uint32_t best = scene_pick(E, first, F, A);\n    (void)A;\n";
        s += "    P3 N{0.f, 0.f, 0.f};\n    switch (best) {\n";
    } else {
        s += "__device__ __forceinline__ P3 scene_norm(P3 E, uint32_t) {\n";
        s += "    const bool active = true;     // called by the lanes that need a normal: the others are masked off\n";
        s += "    P3 F = E, P = E; float A = __builtin_inff(), tx = 0.f, ty = 0.f; uint32_t best = 0u, brank = 0u;\n";
        s += "    const unsigned long long amask = __builtin_amdgcn_ballot_w64(active);\n";
        bool a_inf = true;
        emit_range(program, 0, program.size(), s, 1, a_inf, &slot_node, &union_rank);
        s += "    (void)tx; (void)ty; (void)P; (void)brank; (void)amask;\n    P3 N{0.f, 0.f, 0.f};\n    switch (best) {\n";
    }
    std::vector<std::string> back;            // rotations met on the way down, to undo on the way up
    for (size_t pc = 0; pc < program.size(); ++pc) {
        const Slot& sl = program[pc];
        const uint32_t w0 = sl.w[0], op = w0 & 0xff;
        if (op == OP_END) break;
        const std::string f1 = slot_f(sl, 1), f2 = slot_f(sl, 2), f3 = slot_f(sl, 3), f4 = slot_f(sl, 4),
                          f5 = slot_f(sl, 5), f6 = slot_f(sl, 6);
        if (op == OP_ROTX) back.push_back("N = rotx_back(N, " + f1 + ", " + f2 + ");");
        if (op == OP_ROTY) back.push_back("N = roty_back(N, " + f1 + ", " + f2 + ");");
        if (op == OP_ROTZ) back.push_back("N = rotz_back(N, " + f1 + ", " + f2 + ");");
        if (op == OP_ROTQ_XY) {
            const Slot& z = program[pc + 1];      // always followed by its OP_ROTQ_Z
            back.push_back("N = rotq_back(N, " + f1 + ", " + f2 + ", " + f3 + ", " + f4 + ", " + f5 + ", " + f6 + ", " +
                           slot_f(z, 1) + ", " + slot_f(z, 2) + ", " + slot_f(z, 3) + ");");
        }
        if (op < OP_SFERO || op > OP_EBENO) continue;
        const std::string q = "movo(F, " + f1 + ", " + f2 + ", " + f3 + ")";
        std::string nrm;
        if (op == OP_SFERO) nrm = q;
        else if (op == OP_KUBO) nrm = "kubo_norm(" + q + ", " + f4 + ", " + f5 + ", " + f6 + ")";
        else if (op == OP_CILINDRO) nrm = "cilindro_norm(" + q + ", " + f4 + ", " + f5 + ")";
        else nrm = "P3{" + f4 + ", " + f5 + ", " + f6 + "}";
        if (w0 & F_NEG) nrm = "neg3(" + nrm + ")";
        s += "    case " + std::to_string(pc) + "u: N = " + nrm + "; break;\n";
    }
    s += "    default: break;\n    }\n";
    for (size_t k = back.size(); k-- > 0;) s += "    " + back[k] + "\n";
    s += "    return N;\n}\n";
    s += std::string(R"(
struct JitScene {
    static constexpr bool kNarrow = false;
    struct Scratch { uint32_t first = 0; };      // the group to visit first (wave-uniform)
    __device__ __forceinline__ void init(const RenderParams&) {}
    struct Keep {};
    __device__ __forceinline__ float dist_keep(P3 p, Scratch& st, unsigned long long amask, Keep&) { return )") +
         (grouped ? "scene_program(p, amask, st.first)" : "scene_program(p, amask)") + R"(; }
    __device__ __forceinline__ float dist_keep_all(P3 p, Scratch& s, unsigned long long amask, Keep& k) { return dist_keep(p, s, amask, k); }
    __device__ __forceinline__ P3 norm_keep(P3 p, Scratch& st, const Keep&) { return scene_norm(p, st.first); }
};
}  // namespace rm
)";
    s += kJitKernels;
    return s;
}

namespace {

struct CacheKey {
    int device; uint64_t hash;
    bool operator<(const CacheKey& o) const { return device != o.device ? device < o.device : hash < o.hash; }
};
std::mutex g_mutex;
// Loaded code objects by (device, source hash).  An entry counts the contexts that run it; one that no context
// runs any more stays until kCacheIdle such entries have piled up, then the least recently used ones are unloaded
// (a host that edits its scene every frame would otherwise keep a code object per edit until the process ends).
struct CacheEntry { JitKernels k; int refs = 0; uint64_t used = 0; };
constexpr size_t kCacheIdle = 32;
uint64_t g_clock = 0;
std::map<CacheKey, CacheEntry> g_cache;

void trim_cache() {          // g_mutex held
    for (;;) {
        size_t idle = 0;
        auto oldest = g_cache.end();
        for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
            if (it->second.refs == 0) { ++idle; if (oldest == g_cache.end() || it->second.used < oldest->second.used) oldest = it; }
        if (idle <= kCacheIdle) return;
        if (oldest->second.k.module) (void)hipModuleUnload(oldest->second.k.module);
        g_cache.erase(oldest);
    }
}

hipError_t load_module(const std::vector<char>& code, size_t lds_bytes, JitKernels* out) {
    out->lds_bytes = lds_bytes;
    hipError_t e = hipModuleLoadData(&out->module, code.data());
    if (e != hipSuccess) return e;
    e = hipModuleGetFunction(&out->render, out->module, "rm_render_jit");
    if (e == hipSuccess) e = hipModuleGetFunction(&out->eval_points, out->module, "rm_eval_points_jit");
    if (e == hipSuccess) {
        (void)hipFuncGetAttribute(&out->vgprs, HIP_FUNC_ATTRIBUTE_NUM_REGS, out->render);
        (void)hipFuncGetAttribute(&out->scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, out->render);
        int occ = 0;
        if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, out->render, RM_BLOCK_THREADS, lds_bytes) == hipSuccess && occ > 0)
            out->blocks_per_cu = occ;
    }
    return e;
}

// optional on-disk cache of code objects, the counterpart of the reference's
// CuBin/ directory (Source/cuda_rays.cu:36-58,148-149): RM_JIT_CACHE=<dir>
std::string disk_path(uint64_t hash, const std::string& arch) {
    const char* dir = std::getenv("RM_JIT_CACHE");
    if (!dir || !*dir) return std::string();
    char name[64];
    std::snprintf(name, sizeof name, "/rm_%016llx_", static_cast<unsigned long long>(hash));
    std::string a = arch;
    for (char& c : a) if (c == ':' || c == '+' || c == '/') c = '_';
    return std::string(dir) + name + a + ".co";
}

}  // namespace

std::string jit_full_source(const std::string& scene_source) { return std::string(kPrelude) + scene_source; }

bool jit_build_code(const std::string& arch, const std::string& source, std::vector<char>* code, std::string* log) {
    code->clear();
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, source.c_str(), "rm_scene.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return false;
    const std::string arch_opt = "--offload-arch=" + arch;
    // same arithmetic contract as the ahead-of-time build: no contraction, IEEE sqrt and division
    // (the atomic optimizer is off for the tile queue's sake: device_render.h)
    std::vector<const char*> opts = { arch_opt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                                      // packed fp32 (v_pk_add/mul_f32) issues at 1.7x the cost of a plain op on
                                      // gfx950 and needs moves to pair its operands: measured 6-10 % slower
                                      "-fno-slp-vectorize",
                                      "-mllvm", "-amdgpu-atomic-optimizer-strategy=None" };
    // RM_JIT_FLAGS: extra compiler options, blank-separated (experiments; they become part of the cache key)
    std::vector<std::string> extra;
    if (const char* env = std::getenv("RM_JIT_FLAGS")) {
        std::string cur;
        for (const char* p = env;; ++p) {
            if (*p == ' ' || *p == 0) { if (!cur.empty()) extra.push_back(cur); cur.clear(); if (!*p) break; }
            else cur.push_back(*p);
        }
    }
    for (const std::string& e : extra) opts.push_back(e.c_str());
    hiprtcResult r = hiprtcCompileProgram(prog, static_cast<int>(opts.size()), opts.data());
    size_t log_size = 0;
    if (log && hiprtcGetProgramLogSize(prog, &log_size) == HIPRTC_SUCCESS && log_size > 1) {
        log->resize(log_size);
        (void)hiprtcGetProgramLog(prog, &(*log)[0]);
    }
    size_t code_size = 0;
    if (r == HIPRTC_SUCCESS && hiprtcGetCodeSize(prog, &code_size) == HIPRTC_SUCCESS && code_size) {
        code->resize(code_size);
        if (hiprtcGetCode(prog, code->data()) != HIPRTC_SUCCESS) code->clear();
    }
    (void)hiprtcDestroyProgram(&prog);
    return !code->empty();
}

hipError_t jit_compile(int device, const std::string& scene_source, size_t lds_bytes, JitKernels* out, std::string* log) {
    hipDeviceProp_t props;
    hipError_t e = hipGetDeviceProperties(&props, device);
    if (e != hipSuccess) return e;
    const std::string arch = props.gcnArchName;
    const std::string source = jit_full_source(scene_source);
    const char* flags = std::getenv("RM_JIT_FLAGS");
    const uint64_t hash = fnv1a(flags ? flags : "", fnv1a(arch, fnv1a(source)));

    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_cache.find(CacheKey{device, hash});
    if (it != g_cache.end()) { ++it->second.refs; it->second.used = ++g_clock; *out = it->second.k; return hipSuccess; }

    std::vector<char> code;
    const std::string path = disk_path(hash, arch);
    if (!path.empty()) {
        if (std::FILE* f = std::fopen(path.c_str(), "rb")) {
            char buf[65536]; size_t got;
            while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + got);
            std::fclose(f);
        }
    }
    JitKernels k;
    e = hipSetDevice(device);
    if (e != hipSuccess) return e;
    if (code.empty()) {
        const auto t_build = std::chrono::steady_clock::now();
        (void)jit_build_code(arch, source, &code, log);
        if (code.empty()) return hipErrorInvalidSource;
        const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count();
        // A kernel whose vector registers allow eight wavefronts per SIMD is held to six by its
        // scalar registers (measured on MI355X: 106 SGPRs -> 6 resident, <= 96 -> 7, <= 80 -> 8,
        // whatever the occupancy query says).  Such a kernel is built again with the compiler told
        // to fit eight; the spilled scalars cost less than the two wavefronts bring (infinity.sce
        // 4K 1.105 -> 1.056 ms).  A kernel a few registers above 64 is tried too (chair.sce: 67 -> 64, nothing
        // spilled to memory); the second build is kept only if it fits 64 without scratch.
        // (a kernel further above 64 registers, and the seven-wave build, only where a build takes a few seconds:
        // a union of hundreds of primitives compiles for half a minute a time)
        const bool quick = build_s < 5.0;
        if (load_module(code, lds_bytes, &k) == hipSuccess && k.vgprs > 0 && k.vgprs <= (quick ? 96 : 72) && k.scratch == 0 &&
            !(flags && std::strstr(flags, "RM_KERNEL_ATTR"))) {
            // (eight wavefronts in 64 registers, else seven in 72 - where the eight-wave build has to spill to memory)
            const int fit[2][2] = {{8, 64}, {7, 72}};
            for (int a = 0; a < (quick ? 2 : 1); ++a) {
                std::vector<char> tight;
                JitKernels k2;
                std::string log2;
                const std::string w = std::to_string(fit[a][0]);
                const bool ok = jit_build_code(arch, "#define RM_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(" + w + ", " + w + ")))\n" + source, &tight, &log2) &&
                                load_module(tight, lds_bytes, &k2) == hipSuccess && k2.vgprs <= fit[a][1] && k2.scratch == 0;
                if (k2.module) (void)hipModuleUnload(k2.module);
                if (ok) { code.swap(tight); break; }
            }
        }
        if (k.module) (void)hipModuleUnload(k.module);
        k = JitKernels();
        if (!path.empty()) {
            if (std::FILE* f = std::fopen((path + ".tmp").c_str(), "wb")) {
                const bool ok = std::fwrite(code.data(), 1, code.size(), f) == code.size();
                if (std::fclose(f) == 0 && ok) std::rename((path + ".tmp").c_str(), path.c_str());
            }
        }
    }
    e = load_module(code, lds_bytes, &k);
    if (e != hipSuccess) { if (k.module) (void)hipModuleUnload(k.module); return e; }
    CacheEntry& entry = g_cache[CacheKey{device, hash}];
    entry.k = k; entry.refs = 1; entry.used = ++g_clock;
    *out = k;
    trim_cache();
    return hipSuccess;
}

void jit_release(const JitKernels& k) {
    if (!k.module) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    for (auto& kv : g_cache)
        if (kv.second.k.module == k.module) { if (kv.second.refs > 0) --kv.second.refs; break; }
    trim_cache();
}

hipError_t jit_launch_render(const JitKernels& k, const RenderParams& prm, unsigned blocks, hipStream_t stream) {
    RenderParams copy = prm;
    void* args[] = { &copy };
    return hipModuleLaunchKernel(k.render, blocks, 1, 1, RM_BLOCK_THREADS, 1, 1, static_cast<unsigned>(k.lds_bytes), stream, args, nullptr);
}

hipError_t jit_launch_eval_points(const JitKernels& k, const RenderParams& prm, const float* xyz, uint32_t count,
                                  float* dist_out, float* norm_out, hipStream_t stream) {
    RenderParams copy = prm;
    void* args[] = { &copy, &xyz, &count, &dist_out, &norm_out };
    const unsigned blocks = (count + RM_BLOCK_THREADS - 1) / RM_BLOCK_THREADS;
    return hipModuleLaunchKernel(k.eval_points, blocks, 1, 1, RM_BLOCK_THREADS, 1, 1, static_cast<unsigned>(k.lds_bytes), stream, args, nullptr);
}

}  // namespace rm
