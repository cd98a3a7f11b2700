// kernel_params.h - launch parameters shared by capi.cpp and raymarch_kernels.hip.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "scene_program.h"

#define RM_BLOCK_THREADS 256     // 4 wavefronts: a 16x16 pixel block, one 8x8 tile per wave

namespace rm {

// frames of the per-lane tree walker's private stack (>= scene_build.h kMaxTreeDepth)
constexpr int kWalkDepth = 64;

// rays_info (Include/types.h:15-19) as the kernel wants it: Depth already
// converted to float (Source/cuda_kernels.cu:744), passed by value in the
// kernarg segment, so SetInfo needs no device copy.
struct Camera {
    float depth;
    float light[3];
    float pos[3], dir[3], wvec[3], hvec[3];
};

struct RenderParams {
    const Slot*     program;      // distance program (global memory, scalar loads)
    const TreeNode* tree;         // tree table (global memory, staged to LDS)
    uint32_t n_nodes;
    uint32_t width, height;       // whole image
    uint32_t local_rows;          // rows this context renders
    uint32_t band_rows, rank, nranks;
    uint32_t norm_batch;          // lanes that must wait before normals are computed
    uint32_t evaluator;           // 0 distance program, 1 tree walker
    uint8_t* rgba;  size_t pitch;
    uint8_t* hit;   size_t hit_pitch;
    uint32_t* wave_stats;         // optional diagnostics, 4 dwords per wavefront
    Camera cam;
};

#ifndef __HIPCC_RTC__
hipError_t launch_render(const RenderParams& prm, dim3 grid, size_t lds_bytes, hipStream_t stream);
hipError_t launch_eval_points(const RenderParams& prm, int which, uint32_t idx, const float* xyz, uint32_t count,
                              float* dist_out, float* norm_out, size_t lds_bytes, hipStream_t stream);
hipError_t launch_eval_rays(const RenderParams& prm, const int* xy, uint32_t count, float* out, hipStream_t stream);
// raise the kernels' dynamic-LDS ceiling when a scene needs more than the 64 KiB default
hipError_t configure_lds(size_t lds_bytes);
hipError_t kernel_attributes(int* vgprs, int* scratch_bytes);
#endif

}  // namespace rm
