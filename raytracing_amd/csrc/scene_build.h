// scene_build.h - host-side scene handling: record codecs, validation, and the
// flattener that produces the two device forms declared in scene_program.h.
#pragma once
#include <cstddef>
#include <set>
#include <string>
#include <vector>

#include "../../include/raymarching_amd.h"
#include "scene_program.h"

namespace rm {

// Longest root-to-leaf path the per-lane tree walker's private stack can hold.
constexpr uint32_t kMaxTreeDepth = 64;
// Upper bound on distance-program slots (a node shared by several parents is
// emitted once per path, so a DAG can expand).
constexpr uint32_t kMaxProgramSlots = 1u << 16;

struct SceneBuild {
    std::vector<TreeNode> tree;      // one per input node (unreachable ones zeroed)
    std::vector<Slot>     program;   // distance program, terminated by OP_END
    uint32_t value_slots = 0;        // per-lane LDS stack dwords the program needs
    uint32_t tree_depth  = 0;
    uint32_t reachable   = 0;
    std::vector<unsigned char> seen; // 1 for nodes reachable from node 0
    // For every slot that evaluates a primitive: the node at which its fused chain
    // (komplemento* / movo / primitive) starts, else -1.
    std::vector<int32_t> slot_node;
    // If the whole scene is transforms over ONE union chain of such primitives (the usual
    // shape of a large scene): for each of those start nodes its rank in the order in which
    // the reference's nested kunigajo_N normals resolve equal distances (higher wins), so that
    // a single pass over the members finds the member whose normal the reference returns.
    // Empty otherwise.
    std::vector<uint32_t> union_rank;   // indexed by node
};

int  node_arity(uint32_t type);      // child links, -1 for nenio/unknown
int  node_floats(uint32_t type);     // payload floats

int  decode_bazo(const void* records, size_t n, int layout, rm_node* out);
void encode_sce(const rm_node* nodes, size_t n, std::vector<unsigned char>& bytes);
int  load_sce(const char* path, std::vector<rm_node>& out);
int  save_sce(const char* path, const rm_node* nodes, size_t n);

// Divisor constants (float bit patterns) for which the device has proved the short
// quotient sequence exact (device_math.h div_const; capi.cpp verify_divisors).
typedef std::set<uint32_t> DivisorSet;
// The divisor constants of a node table: dot(a,a) of every senfina_ripeto, k of every glata_*.
std::vector<float> scene_divisors(const rm_node* nodes, size_t n);
float ripeto_aa(const rm_node& nd);                  // dot(a,a) in the reference's operation order
// RN(1/c) if c is in `exact`, else 0 (= use IEEE division)
float divisor_reciprocal(const DivisorSet* exact, float c);

// Validate and flatten.  Returns an rm_status.
// cull: put bounding-sphere tests (OP_CULL) in front of the sub-trees of union chains
int  build_scene(const rm_node* nodes, size_t n, SceneBuild& out, const DivisorSet* exact = nullptr, bool cull = true);

// A volume outside of which DIST(root, x) >= dist(x, volume) - r (scene_bound.cpp); kind RM_BOUND_NONE if the
// tree offers none.
void scene_bound(const rm_node* nodes, size_t n, rm_scene_bound* out);

// Human-readable listing of a distance program (tests).
std::string disassemble(const std::vector<Slot>& program);

}  // namespace rm
