// raymarch_kernels.hip - gfx950 (CDNA4) kernels of the per-pixel raymarch path,
// ahead-of-time build: the generic evaluators (distance program + tree walker).
//
// One fused kernel replaces the reference's kernel_SetRays + kernel_Process
// (Source/cuda_kernels.cu:598-762): it derives each pixel's ray in registers,
// sphere-traces it, takes the surface normal, marches the soft-shadow ray and
// writes one RGBA8 pixel.  kernel_SetPrimitives (:721-728) has no device
// counterpart; node types are resolved on the host (scene_build.cpp).
//
// Arithmetic is fp32 with IEEE add/mul/div/sqrt and NO contraction (build with
// -ffp-contract=off): every expression keeps the reference's operation order,
// so frames are bit-identical to the CPU restatement.
//
// device_math.h    primitives, transforms, conventions
// device_interp.h  distance program interpreter, tree walker
// device_render.h  the kernel body (also compiled at run time against per-scene
//                  generated code, see jit.cpp)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>

#include "device_interp.h"
#include "device_render.h"

namespace rm {

// =============================================================================
// Scheduler block of rm_process_static, the entry point that allocates nothing
// (include/raymarching_amd.h): counter sets, tile lists and status block that a
// context otherwise gets from hipMalloc.  Every GPU the library runs on holds its
// own copy of this array at its own address until the process ends.  In host code
// the array's NAME is only a handle for hipGetSymbolAddress - passing it to a
// kernel would hand the device a host address - so the address is looked up once
// per device, with that device current (rm_create does it), and only that
// pointer is ever used: as kernel argument, for hipMemsetAsync, for copies.
// =============================================================================
static __device__ __align__(256) __attribute__((used)) uint32_t scratch[RM_STATIC_DWORDS];

hipError_t static_scratch_address(int device, uint32_t** ptr) {
    static std::mutex mu;
    static std::map<int, uint32_t*> known;        // device -> address of its copy
    std::lock_guard<std::mutex> lock(mu);
    auto it = known.find(device);
    if (it == known.end()) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        it = known.emplace(device, static_cast<uint32_t*>(p)).first;
    }
    *ptr = it->second;
    return hipSuccess;
}

__global__ void __launch_bounds__(RM_BLOCK_THREADS)
render_kernel(const RenderParams prm) {
    render_body<InterpScene>(prm);
}

// =============================================================================
// Known-answer hooks: node-level DIST/NORM and ray generation on the device.
// =============================================================================
__global__ void __launch_bounds__(RM_BLOCK_THREADS)
eval_points_kernel(const RenderParams prm, int which, uint32_t idx, const float* __restrict__ xyz,
                   uint32_t count, float* __restrict__ dist_out, float* __restrict__ norm_out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const TreeNode* tree = reinterpret_cast<const TreeNode*>(lds);
    float* stack = reinterpret_cast<float*>(lds + prm.n_nodes * 12) + threadIdx.x;
    stage_tree(prm.tree, prm.n_nodes, lds);
    const uint32_t i = blockIdx.x * RM_BLOCK_THREADS + threadIdx.x;
    const bool live = i < count;
    P3 p{0.f, 0.f, 0.f};
    if (live) p = P3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    WalkStack st;
    float d;
    if (which == 0) d = run_program(prm.program, p, stack, live);      // wave-uniform: all lanes run it
    else            d = live ? walk_dist(tree, idx, p, st, 0) : 0.f;
    if (live) {
        dist_out[i] = d;
        if (norm_out) {
            P3 N = walk_norm(tree, idx, p, st);
            norm_out[3 * i] = N.x; norm_out[3 * i + 1] = N.y; norm_out[3 * i + 2] = N.z;
        }
    }
}

__global__ void eval_rays_kernel(const RenderParams prm, const int* __restrict__ xy, uint32_t count,
                                 float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    P3 p, d;
    make_ray(prm.cams[0], prm.width, prm.height, xy[2 * i], xy[2 * i + 1], p, d);
    out[6 * i] = p.x; out[6 * i + 1] = p.y; out[6 * i + 2] = p.z;
    out[6 * i + 3] = d.x; out[6 * i + 4] = d.y; out[6 * i + 5] = d.z;
}

// =============================================================================
// Proofs by enumeration for device_math.h's short sqrt and quotient sequences.
// =============================================================================

// sqrt_exact against IEEE sqrtf over `count` consecutive bit patterns from `first`
__global__ void __launch_bounds__(256)
verify_sqrt_kernel(uint32_t first, unsigned long long count, unsigned long long* bad) {
    unsigned long long mine = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float x = as_f(first + (uint32_t)i);
        const float want = sqrtf(x), got = sqrt_exact(x);
        if (as_u(want) != as_u(got) && !(want != want && got != got)) ++mine;
    }
    if (mine) atomicAdd(bad, mine);
}

// div_const (and floor(div_const_for_floor + .5)) against IEEE division.  blockIdx.y picks the divisor; numerators are `count`
// consecutive bit patterns from `first` (one binade, [1,2), proves a divisor: device_math.h)
__global__ void __launch_bounds__(256)
verify_div_kernel(const float* __restrict__ c, const float* __restrict__ rc, uint32_t first, unsigned long long count,
                  unsigned long long* bad) {
    const float cc = c[blockIdx.y], rr = rc[blockIdx.y];
    unsigned long long mine = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float a = as_f(first + (uint32_t)i);
        const float want = a / cc, got = div_const(a, cc, rr);
        if (as_u(want) != as_u(got) && !(want != want && got != got)) ++mine;
        // and the lattice's one-sidedly guarded variant, in the only form its result is used in (:577)
        const float wantN = floorf(want + .5f), gotN = floorf(div_const_for_floor(a, cc, rr) + .5f);
        if (as_u(wantN) != as_u(gotN) && !(wantN != wantN && gotN != gotN)) ++mine;
    }
    if (mine) atomicAdd(bad + blockIdx.y, mine);
}

// ---- launchers (called from capi.cpp) -----------------------------------------

// The ranks' luminance planes into RGBA8 frames.  A workgroup takes RM_ASM_ROWS rows of one frame (blockIdx.y;
// z = frame) and 4 x 256 quads of 4 pixels along them: which rank's part holds a row, and where, is the same for
// the whole workgroup and comes out of the kernarg segment by scalar loads.  A lane moves 4 luminance bytes in
// and 16 bytes of RGBA8 out (built as :694-699 build them) per step, neighbouring lanes neighbouring quads, so a
// wavefront's load is 256 contiguous bytes and its store 1 KiB; the steps of a lane are independent, and the
// stores bypass the caches (the frame is not read again here).  HBM-bound: 5 B/px.
#ifndef RM_ASM_ROWS
#define RM_ASM_ROWS 2
#endif
#ifndef RM_ASM_QUADS
#define RM_ASM_QUADS 4
#endif

__device__ __forceinline__ uint32_t pixel_of_luminance(uint32_t l8) {
    const float L = (float)l8;
    return to_u8(L * RM_MATERIAL_R) | (to_u8(L * RM_MATERIAL_G) << 8) | (to_u8(L * RM_MATERIAL_B) << 16) | 0xff000000u;
}

__global__ void __launch_bounds__(256)
assemble_luminance_kernel(AssembleParams by_value) {
    (void)by_value;
    typedef const __attribute__((address_space(4))) AssembleParams Kernarg;
    Kernarg* kp = (Kernarg*)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t width = kp->width, height = kp->height;
    const uint32_t band_rows = kp->band_rows, frame = blockIdx.z;
    const bool vector = kp->vector != 0;
    for (uint32_t y0 = blockIdx.y * RM_ASM_ROWS; y0 < height; y0 += gridDim.y * RM_ASM_ROWS) {
#pragma unroll
        for (uint32_t dy = 0; dy < RM_ASM_ROWS; ++dy) {
            const uint32_t y = y0 + dy;
            if (y >= height) break;
            uint32_t r = 0, ly = y;
            if (band_rows) {
                const uint32_t band = y / band_rows, period = kp->period, run = band / period, place = band - run * period;
                r = kp->owner[place];
                ly = (run * kp->own[r] + kp->index[place]) * band_rows + (y - band * band_rows);
            }
            const uint8_t* src_row = kp->parts + (size_t)r * kp->rank_stride + (size_t)frame * kp->part_frame_stride + (size_t)ly * kp->part_pitch;
            uint8_t* dst_row = kp->rgba + (size_t)frame * kp->rgba_frame_stride + (size_t)y * kp->pitch;
#pragma unroll
            for (uint32_t j = 0; j < RM_ASM_QUADS; ++j) {
                const uint32_t x0 = ((blockIdx.x * RM_ASM_QUADS + j) * 256 + threadIdx.x) * 4;
                if (x0 >= width) break;
                const uint8_t* src = src_row + x0;
                uint8_t* dst = dst_row + (size_t)x0 * 4;
                if (vector && x0 + 4 <= width) {
                    const uint32_t l4 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(src));
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 out;
                    out.x = pixel_of_luminance(l4 & 0xffu);
                    out.y = pixel_of_luminance((l4 >> 8) & 0xffu);
                    out.z = pixel_of_luminance((l4 >> 16) & 0xffu);
                    out.w = pixel_of_luminance(l4 >> 24);
                    __builtin_nontemporal_store(out, reinterpret_cast<u32x4*>(dst));
                } else {
                    for (uint32_t k = 0; k < 4 && x0 + k < width; ++k)
                        reinterpret_cast<uint32_t*>(dst)[k] = pixel_of_luminance(src[k]);
                }
            }
        }
    }
}

hipError_t launch_assemble_luminance(const AssembleParams& prm, uint32_t frames, hipStream_t stream) {
    if (!prm.width || !prm.height || !frames) return hipSuccess;
    const unsigned quads = (prm.width + 3) / 4, per_block = 256 * RM_ASM_QUADS;
    const unsigned row_groups = (prm.height + RM_ASM_ROWS - 1) / RM_ASM_ROWS;
    const dim3 grid((quads + per_block - 1) / per_block, row_groups < 65535u ? row_groups : 65535u, frames);
    hipLaunchKernelGGL(assemble_luminance_kernel, grid, dim3(256), 0, stream, prm);
    return hipGetLastError();
}

hipError_t launch_verify_sqrt(uint32_t first, unsigned long long count, unsigned long long* bad, hipStream_t stream) {
    hipLaunchKernelGGL(verify_sqrt_kernel, dim3(2048), dim3(256), 0, stream, first, count, bad);
    return hipGetLastError();
}

hipError_t launch_verify_div(const float* c, const float* rc, uint32_t n, uint32_t first, unsigned long long count,
                             unsigned long long* bad, hipStream_t stream) {
    hipLaunchKernelGGL(verify_div_kernel, dim3(count > (1u << 24) ? 2048 : 64, n), dim3(256), 0, stream, c, rc, first, count, bad);
    return hipGetLastError();
}

hipError_t launch_render(const RenderParams& prm, unsigned blocks, size_t lds_bytes, hipStream_t stream) {
    hipLaunchKernelGGL(render_kernel, dim3(blocks), dim3(RM_BLOCK_THREADS), lds_bytes, stream, prm);
    return hipGetLastError();
}

hipError_t render_occupancy(size_t lds_bytes, int* blocks_per_cu) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void*>(render_kernel),
                                                        RM_BLOCK_THREADS, lds_bytes);
}

hipError_t launch_eval_points(const RenderParams& prm, int which, uint32_t idx, const float* xyz, uint32_t count,
                              float* dist_out, float* norm_out, size_t lds_bytes, hipStream_t stream) {
    dim3 grid((count + RM_BLOCK_THREADS - 1) / RM_BLOCK_THREADS);
    hipLaunchKernelGGL(eval_points_kernel, grid, dim3(RM_BLOCK_THREADS), lds_bytes, stream,
                       prm, which, idx, xyz, count, dist_out, norm_out);
    return hipGetLastError();
}

hipError_t launch_eval_rays(const RenderParams& prm, const int* xy, uint32_t count, float* out, hipStream_t stream) {
    dim3 grid((count + 255) / 256);
    hipLaunchKernelGGL(eval_rays_kernel, grid, dim3(256), 0, stream, prm, xy, count, out);
    return hipGetLastError();
}

hipError_t configure_lds(size_t lds_bytes) {
    if (lds_bytes <= 64 * 1024) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(render_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(eval_points_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t kernel_attributes(int* vgprs, int* scratch_bytes) {
    hipFuncAttributes a;
    hipError_t e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(render_kernel));
    if (e != hipSuccess) return e;
    *vgprs = a.numRegs;
    *scratch_bytes = (int)a.localSizeBytes;
    return hipSuccess;
}

}  // namespace rm
