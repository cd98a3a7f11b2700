// scene_bound.cpp - a volume outside of which the scene's distance function is provably large.
//
// The reference marches every primary ray until it has travelled RM_MAX_DIST (Source/cuda_kernels.cu:707-714),
// also the rays that pass the scene by: on chair.sce at 3840x2160 94 % of the rays hit nothing and their ~11
// steps each are half of all the work.  A ray that stays clear of everything solid ends as the black pixel
// it started as (:622, :717) whatever its steps were, so it need not be marched - if "stays clear" is proved.
//
// The proof is a lower bound of DIST(root, x), built bottom-up over the node tree:
//     DIST(node, x) >= dist(x, V) - r        for every point x
// with V a point (a sphere of radius r) or a line (a cylinder of radius r).  It holds for
//   sfero                    |x| - r itself
//   kubo, cilindro           their value is at least the distance to the sphere around their corners / rims
//   movo, rotacio*           the child's bound carried through the rigid map (a rotacio whose cos/sin or
//                            quaternion is not of unit length to 1e-5 gives no bound)
//   kunigajo_N (min)         the sphere enclosing the children's spheres: nearer than each of them
//   komunajo_N (max)         any one child's bound (max >= each operand): the tightest is taken
//   glata_*_2                within |k|/4 of min / max of the operands: the enclosing sphere, grown by |k|/4
//   senfina_ripeto           the child is evaluated at x - N a for an integer N: at least as far from the
//                            child's centre c as x is from the line c + s a, whatever N the rounding picked
//   komplemento, ebeno, portanta_sfero, a lattice of lattices          no bound (the scene fills space)
// The device turns this into: which part [0, t_out] of a ray can come within a margin of the volume
// (device_render.h, ray_window): outside of it DIST exceeds the margin (8 units and more: four hundred times
// RM_MIN_DIST, against a rounding error of DIST below 0.1 for coordinates up to 10^5), so no step there can
// find a surface, and all the march does there is walk on.
#include <cmath>
#include <cstring>
#include <vector>

#include "scene_build.h"

namespace rm {

namespace {

struct Vol {
    int kind = RM_BOUND_NONE;
    double c[3] = {0, 0, 0};      // a point of the volume's centre (sphere: the centre; cylinder: a point of the axis)
    double u[3] = {0, 0, 0};      // cylinder: unit direction of the axis
    double r = 0;
};

double len3(const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// the smallest sphere around two spheres
Vol enclose(const Vol& a, const Vol& b) {
    double d[3] = { b.c[0] - a.c[0], b.c[1] - a.c[1], b.c[2] - a.c[2] };
    const double dist = len3(d);
    if (dist + b.r <= a.r) return a;
    if (dist + a.r <= b.r) return b;
    Vol o;
    o.kind = RM_BOUND_SPHERE;
    o.r = 0.5 * (dist + a.r + b.r);
    const double s = dist > 0 ? (o.r - a.r) / dist : 0;
    for (int k = 0; k < 3; ++k) o.c[k] = a.c[k] + s * d[k];
    return o;
}

// a small sphere around many spheres: the centre walks towards the farthest sphere in ever smaller steps
// (Badoiu and Clarkson's core-set iteration, within a few per cent of the smallest after a hundred steps); the
// radius is then whatever that centre needs, so the result encloses them all whatever the walk did
Vol enclose_all(const std::vector<Vol>& v) {
    if (v.size() == 1) return v[0];
    if (v.size() == 2) return enclose(v[0], v[1]);
    Vol o;
    o.kind = RM_BOUND_SPHERE;
    for (const Vol& a : v) for (int k = 0; k < 3; ++k) o.c[k] += a.c[k] / static_cast<double>(v.size());
    auto reach = [&](const Vol& a) {
        const double d[3] = { a.c[0] - o.c[0], a.c[1] - o.c[1], a.c[2] - o.c[2] };
        return len3(d) + a.r;
    };
    for (int it = 1; it <= 200; ++it) {
        size_t far = 0; double best = -1;
        for (size_t i = 0; i < v.size(); ++i) { const double r = reach(v[i]); if (r > best) { best = r; far = i; } }
        double d[3] = { v[far].c[0] - o.c[0], v[far].c[1] - o.c[1], v[far].c[2] - o.c[2] };
        const double dist = len3(d);
        if (!(dist > 0)) break;
        // towards the farthest point of that sphere, a step of 1/(it+1) of the way
        const double step = (dist + v[far].r) / dist / (it + 1);
        for (int k = 0; k < 3; ++k) o.c[k] += step * d[k];
    }
    for (const Vol& a : v) { const double r = reach(a); if (r > o.r) o.r = r; }
    // never worse than folding them pairwise
    Vol fold = v[0];
    for (size_t i = 1; i < v.size(); ++i) fold = enclose(fold, v[i]);
    return fold.r < o.r ? fold : o;
}

struct Bounder {
    const rm_node* nodes; size_t n;
    int depth = 0;

    // v -> M^T v for the row-major 3x3 M of a node that evaluates its child at M x (so a bound around c in
    // the child's frame is a bound around M^T c in the parent's, M being orthonormal)
    static void back(const double M[9], const double* v, double* out) {
        for (int k = 0; k < 3; ++k) out[k] = M[0 + k] * v[0] + M[3 + k] * v[1] + M[6 + k] * v[2];
    }
    static bool orthonormal(const double M[9]) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += M[3 * i + k] * M[3 * j + k];
                if (!(std::fabs(s - (i == j ? 1.0 : 0.0)) < 1e-5)) return false;
            }
        return true;
    }
    Vol rotated(const Vol& child, const double M[9]) {
        if (child.kind == RM_BOUND_NONE || !orthonormal(M)) return Vol();
        Vol o = child;
        back(M, child.c, o.c);
        if (child.kind == RM_BOUND_CYLINDER) back(M, child.u, o.u);
        return o;
    }

    // the spheres of everything a (nested) union unites, through the movo nodes between its levels; false if
    // one of them has no sphere
    bool union_members(size_t i, std::vector<Vol>& out, const double* shift = nullptr) {
        if (i >= n || depth > 80) return false;
        const rm_node& nd = nodes[i];
        const double zero[3] = {0, 0, 0};
        if (!shift) shift = zero;
        if (nd.type == RM_TYPE_KUNIGAJO_2 || nd.type == RM_TYPE_KUNIGAJO_3 || nd.type == RM_TYPE_KUNIGAJO_4) {
            ++depth;
            bool ok = true;
            for (int k = 0; k < node_arity(nd.type) && ok; ++k) ok = union_members(i + static_cast<size_t>(nd.o[k]), out, shift);
            --depth;
            return ok;
        }
        if (nd.type == RM_TYPE_MOVO) {
            const size_t kid = i + static_cast<size_t>(nd.o[0]);
            if (kid < n && (nodes[kid].type == RM_TYPE_KUNIGAJO_2 || nodes[kid].type == RM_TYPE_KUNIGAJO_3 || nodes[kid].type == RM_TYPE_KUNIGAJO_4)) {
                const double moved[3] = { shift[0] + nd.f[0], shift[1] + nd.f[1], shift[2] + nd.f[2] };
                ++depth;
                const bool ok = union_members(kid, out, moved);
                --depth;
                return ok;
            }
        }
        Vol v = of(i);
        if (v.kind != RM_BOUND_SPHERE) return false;
        for (int k = 0; k < 3; ++k) v.c[k] += shift[k];
        out.push_back(v);
        return true;
    }

    Vol of(size_t i) {
        if (i >= n || depth > 80) return Vol();
        const rm_node& nd = nodes[i];
        auto kid = [&](int k) { return i + static_cast<size_t>(nd.o[k]); };
        ++depth;
        Vol o;
        switch (nd.type) {
        case RM_TYPE_SFERO:
            o.kind = RM_BOUND_SPHERE; o.r = std::fmax(0.0, static_cast<double>(nd.f[0]));
            break;
        case RM_TYPE_KUBO: {          // |max(q, 0)| and max(q) are both >= |x| - |b| (q = |x| - b componentwise)
            const double b[3] = { std::fabs(nd.f[0]), std::fabs(nd.f[1]), std::fabs(nd.f[2]) };
            o.kind = RM_BOUND_SPHERE; o.r = len3(b);
            break;
        }
        case RM_TYPE_CILINDRO: {
            const double b[3] = { std::fabs(nd.f[0]), std::fabs(nd.f[1]), 0 };
            o.kind = RM_BOUND_SPHERE; o.r = len3(b);
            break;
        }
        case RM_TYPE_MOVO: {
            o = of(kid(0));
            if (o.kind != RM_BOUND_NONE) for (int k = 0; k < 3; ++k) o.c[k] += nd.f[k];
            break;
        }
        case RM_TYPE_ROTACIOX: case RM_TYPE_ROTACIOY: case RM_TYPE_ROTACIOZ: {
            const double c = nd.f[0], s = nd.f[1];
            // Source/cuda_kernels.cu:430-432, 454-456, 478-480: the child's point is M x
            const double MX[9] = { 1, 0, 0,  0, c, s,  0, -s, c };
            const double MY[9] = { c, 0, -s,  0, 1, 0,  s, 0, c };
            const double MZ[9] = { c, s, 0,  -s, c, 0,  0, 0, 1 };
            o = rotated(of(kid(0)), nd.type == RM_TYPE_ROTACIOX ? MX : nd.type == RM_TYPE_ROTACIOY ? MY : MZ);
            break;
        }
        case RM_TYPE_ROTACIOQ: {      // :504-530: P = x + 2 (Q x)
            const double w = nd.f[0], x = nd.f[1], y = nd.f[2], z = nd.f[3];
            const double t = x * x + y * y + z * z;
            const double Q[9] = { x * x - t, x * y - z * w, z * x + y * w,
                                  x * y + z * w, y * y - t, y * z - x * w,
                                  z * x - y * w, y * z + x * w, z * z - t };
            double M[9];
            for (int k = 0; k < 9; ++k) M[k] = 2 * Q[k] + ((k % 4 == 0) ? 1.0 : 0.0);
            o = rotated(of(kid(0)), M);
            break;
        }
        case RM_TYPE_SENFINA_RIPETO: {
            const Vol c = of(kid(0));
            const double a[3] = { nd.f[0], nd.f[1], nd.f[2] };
            const double la = len3(a);
            if (c.kind == RM_BOUND_SPHERE && la > 0 && std::isfinite(la)) {
                o = c; o.kind = RM_BOUND_CYLINDER;
                for (int k = 0; k < 3; ++k) o.u[k] = a[k] / la;
            }
            break;
        }
        case RM_TYPE_KUNIGAJO_2: case RM_TYPE_KUNIGAJO_3: case RM_TYPE_KUNIGAJO_4: {
            // the members of a union of unions of ... are enclosed together: folding them pair by pair up the
            // tree grows the sphere at every level (config 4's 341 leaves: 139 against 117 units of radius)
            std::vector<Vol> members;
            if (union_members(i, members)) o = enclose_all(members);
            break;
        }
        case RM_TYPE_GLATA_KUNIGAJO_2: case RM_TYPE_GLATA_KOMUNAJO_2: {
            const Vol a = of(kid(0)), b = of(kid(1));
            if (a.kind == RM_BOUND_SPHERE && b.kind == RM_BOUND_SPHERE) {
                o = enclose(a, b);
                o.r += 0.25 * std::fabs(static_cast<double>(nd.f[0]));         // :356-359, :384-387: k h (1 - h) <= k / 4
            }
            break;
        }
        case RM_TYPE_KOMUNAJO_2: case RM_TYPE_KOMUNAJO_3: case RM_TYPE_KOMUNAJO_4: {
            const int na = node_arity(nd.type);
            for (int k = 0; k < na; ++k) {
                const Vol v = of(kid(k));
                if (v.kind == RM_BOUND_NONE) continue;
                // a sphere before a cylinder, the smaller radius among equals
                if (o.kind == RM_BOUND_NONE || (v.kind == RM_BOUND_SPHERE && o.kind == RM_BOUND_CYLINDER) ||
                    (v.kind == o.kind && v.r < o.r)) o = v;
            }
            break;
        }
        default: break;               // komplemento, ebeno, portanta_sfero, nenio
        }
        --depth;
        if (o.kind != RM_BOUND_NONE && !(std::isfinite(o.r) && std::isfinite(o.c[0]) && std::isfinite(o.c[1]) && std::isfinite(o.c[2]))) o = Vol();
        return o;
    }
};

}  // namespace

void scene_bound(const rm_node* nodes, size_t n, rm_scene_bound* out) {
    std::memset(out, 0, sizeof *out);
    if (!nodes || !n) return;
    Bounder b{nodes, n};
    const Vol v = b.of(0);
    // a volume centred or reaching far out is of no use to the device (rm_types.h: the margin is sized for
    // coordinates up to RM_WINDOW_MAX_COORD)
    if (v.kind == RM_BOUND_NONE || !(v.r <= RM_WINDOW_MAX_COORD)) return;
    double c[3] = { v.c[0], v.c[1], v.c[2] };
    if (v.kind == RM_BOUND_CYLINDER) {           // any point of the axis will do: the one nearest to the origin
        const double s = c[0] * v.u[0] + c[1] * v.u[1] + c[2] * v.u[2];
        for (int k = 0; k < 3; ++k) c[k] -= s * v.u[k];
    }
    for (int k = 0; k < 3; ++k) if (!(std::fabs(c[k]) <= RM_WINDOW_MAX_COORD)) return;
    out->kind = v.kind;
    for (int k = 0; k < 3; ++k) { out->c[k] = static_cast<float>(c[k]); out->u[k] = static_cast<float>(v.u[k]); }
    // rounding of the centre and the axis to float, and rotations that are orthonormal only to 1e-5: a part
    // in 10^4 of slack.  The axis direction tilts by up to 1e-7 in rounding, which over the RM_WINDOW_MAX_COORD +
    // RM_MAX_DIST that a march can be away from the axis point is 3e-3.
    out->r = static_cast<float>(v.r * 1.0001 + 1e-4 * len3(c) + 1e-2);
}

}  // namespace rm
