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

// largest scene (reachable nodes) that is turned into code; bigger ones run on the interpreter
constexpr uint32_t kJitMaxNodes = 64;

struct JitKernels {
    hipModule_t   module = nullptr;
    hipFunction_t render = nullptr;
    hipFunction_t eval_points = nullptr;
    int vgprs = 0, scratch = 0;
    int blocks_per_cu = 1;      // resident workgroups per CU of `render`
};

// HIP source of the d<i>/n<i> function pairs of every reachable node plus the
// kernels built on them.  `seen[i]` marks reachable nodes (scene_build).
std::string generate_scene_source(const rm_node* nodes, size_t n, const unsigned char* seen, const DivisorSet* exact = nullptr);

// prelude (flattened device headers) + scene source: what hiprtc is given
std::string jit_full_source(const std::string& scene_source);
// hiprtc only, no device needed: code object of `source` for `arch` (e.g. "gfx950")
bool jit_build_code(const std::string& arch, const std::string& source, std::vector<char>* code, std::string* log);

// Compile (or fetch from the cache) the kernels of `scene_source` for `device`.
// Returns hipSuccess, or an error with the compiler log in *log.
hipError_t jit_compile(int device, const std::string& scene_source, JitKernels* out, std::string* log);

hipError_t jit_launch_render(const JitKernels& k, const RenderParams& prm, unsigned blocks, hipStream_t stream);
hipError_t jit_launch_eval_points(const JitKernels& k, const RenderParams& prm, const float* xyz, uint32_t count,
                                  float* dist_out, float* norm_out, hipStream_t stream);

}  // namespace rm
