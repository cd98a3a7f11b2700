// jit.h - per-scene specialised kernels, compiled at run time with hiprtc.
//
// The reference compiles its kernels at start-up with NVRTC and caches the cubin
// keyed by a hash of the source (Source/cuda_rays.cu:29-166).  Here run-time
// compilation buys something the reference never had: the scene itself becomes
// straight-line code.  Every node turns into a pair of inline functions
// (distance, normal) with its constants as literals, in the reference's own
// operation order, so the march loop contains no slot fetch, no dispatch and no
// table walk - the scalar unit, which bounds the interpreter, is idle.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>
#include <vector>

#include "../../include/raymarching_amd.h"
#include "kernel_params.h"
#include "scene_build.h"

namespace rm {

// Scenes of up to kJitMaxNodes reachable nodes get a distance and a normal function per node
// (generate_scene_source).  Larger ones, up to kJitMaxSlots program slots and 64 KiB of tree
// table, get their distance program compiled (generate_program_source) and keep the tree walker
// for normals; beyond that the interpreter runs.
constexpr uint32_t kJitMaxNodes = 64;
constexpr uint32_t kJitMaxSlots = 4096;
constexpr size_t   kJitMaxLds   = 64 * 1024;

struct JitKernels {
    hipModule_t   module = nullptr;
    hipFunction_t render = nullptr;
    hipFunction_t eval_points = nullptr;
    int vgprs = 0, scratch = 0;
    int blocks_per_cu = 1;      // resident workgroups per CU of `render`
    size_t lds_bytes = 0;       // dynamic LDS both kernels are launched with (the tree table, for large scenes)
};

// HIP source of the d<i>/n<i> function pairs of every reachable node plus the
// kernels built on them.  `seen[i]` marks reachable nodes (scene_build).
// With `program` (the flattened distance program) the march evaluates that, compiled slot by slot
// with its culls, and the per-node d<i> serve the normal functions only.
std::string generate_scene_source(const rm_node* nodes, size_t n, const unsigned char* seen, const DivisorSet* exact = nullptr,
                                  const std::vector<Slot>* program = nullptr, uint32_t value_slots = 0);

// The same kernels from the flattened distance program of a large scene.
// slot_node / union_rank: SceneBuild's; with a non-empty union_rank normals come from one pass over
// the union's members and the kernels need no LDS, else from the tree walker.
std::string generate_program_source(const std::vector<Slot>& program, uint32_t value_slots,
                                    const std::vector<int32_t>& slot_node, const std::vector<uint32_t>& union_rank);

// prelude (flattened device headers) + scene source: what hiprtc is given
std::string jit_full_source(const std::string& scene_source);
// hiprtc only, no device needed: code object of `source` for `arch` (e.g. "gfx950")
bool jit_build_code(const std::string& arch, const std::string& source, std::vector<char>* code, std::string* log);

// Compile (or fetch from the cache) the kernels of `scene_source` for `device`.
// Returns hipSuccess, or an error with the compiler log in *log.
hipError_t jit_compile(int device, const std::string& scene_source, size_t lds_bytes, JitKernels* out, std::string* log);

// A context is done with kernels it got from jit_compile (none of them may be running any more): the code
// object may be unloaded once enough others have come and gone.
void jit_release(const JitKernels& k);

hipError_t jit_launch_render(const JitKernels& k, const RenderParams& prm, unsigned blocks, hipStream_t stream);
hipError_t jit_launch_eval_points(const JitKernels& k, const RenderParams& prm, const float* xyz, uint32_t count,
                                  float* dist_out, float* norm_out, hipStream_t stream);

}  // namespace rm
