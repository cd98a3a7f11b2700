// jit.cpp - scene -> HIP source -> hiprtc -> module (see jit.h).
#include "jit.h"

#include <hip/hiprtc.h>

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

uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

}  // namespace

std::string generate_scene_source(const rm_node* nodes, size_t n, const unsigned char* seen, const DivisorSet* exact) {
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
            const std::string t = "ripeto(p, " + lit3(nd.f) + ", " + lit(aa) + ", " + lit(divisor_reciprocal(exact, aa)) + ")";
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
    s += R"(
struct JitScene {
    struct Scratch {};
    __device__ __forceinline__ void init(const RenderParams&) {}
    __device__ __forceinline__ float dist(P3 p, Scratch&, bool) { return d0(p); }
    __device__ __forceinline__ P3 norm(P3 p, Scratch&) { return n0(p); }
};
}  // namespace rm

extern "C" __global__ void __launch_bounds__(RM_BLOCK_THREADS) rm_render_jit(const rm::RenderParams prm) {
    rm::render_body<rm::JitScene>(prm);
}

extern "C" __global__ void __launch_bounds__(RM_BLOCK_THREADS)
rm_eval_points_jit(const rm::RenderParams prm, const float* __restrict__ xyz, uint32_t count,
                   float* __restrict__ dist_out, float* __restrict__ norm_out) {
    const uint32_t i = blockIdx.x * RM_BLOCK_THREADS + threadIdx.x;
    if (i >= count) return;
    rm::P3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    dist_out[i] = rm::d0(p);
    if (norm_out) {
        rm::P3 N = rm::n0(p);
        norm_out[3 * i] = N.x; norm_out[3 * i + 1] = N.y; norm_out[3 * i + 2] = N.z;
    }
}
)";
    return s;
}

namespace {

struct CacheKey {
    int device; uint64_t hash;
    bool operator<(const CacheKey& o) const { return device != o.device ? device < o.device : hash < o.hash; }
};
std::mutex g_mutex;
std::map<CacheKey, JitKernels> g_cache;      // modules live until the process ends

hipError_t load_module(const std::vector<char>& code, JitKernels* out) {
    hipError_t e = hipModuleLoadData(&out->module, code.data());
    if (e != hipSuccess) return e;
    e = hipModuleGetFunction(&out->render, out->module, "rm_render_jit");
    if (e == hipSuccess) e = hipModuleGetFunction(&out->eval_points, out->module, "rm_eval_points_jit");
    if (e == hipSuccess) {
        (void)hipFuncGetAttribute(&out->vgprs, HIP_FUNC_ATTRIBUTE_NUM_REGS, out->render);
        (void)hipFuncGetAttribute(&out->scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, out->render);
        int occ = 0;
        if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, out->render, RM_BLOCK_THREADS, 0) == hipSuccess && occ > 0)
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

hipError_t jit_compile(int device, const std::string& scene_source, JitKernels* out, std::string* log) {
    hipDeviceProp_t props;
    hipError_t e = hipGetDeviceProperties(&props, device);
    if (e != hipSuccess) return e;
    const std::string arch = props.gcnArchName;
    const std::string source = jit_full_source(scene_source);
    const char* flags = std::getenv("RM_JIT_FLAGS");
    const uint64_t hash = fnv1a(flags ? flags : "", fnv1a(arch, fnv1a(source)));

    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_cache.find(CacheKey{device, hash});
    if (it != g_cache.end()) { *out = it->second; return hipSuccess; }

    std::vector<char> code;
    const std::string path = disk_path(hash, arch);
    if (!path.empty()) {
        if (std::FILE* f = std::fopen(path.c_str(), "rb")) {
            char buf[65536]; size_t got;
            while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + got);
            std::fclose(f);
        }
    }
    if (code.empty()) {
        (void)jit_build_code(arch, source, &code, log);
        if (code.empty()) return hipErrorInvalidSource;
        if (!path.empty()) {
            if (std::FILE* f = std::fopen((path + ".tmp").c_str(), "wb")) {
                const bool ok = std::fwrite(code.data(), 1, code.size(), f) == code.size();
                if (std::fclose(f) == 0 && ok) std::rename((path + ".tmp").c_str(), path.c_str());
            }
        }
    }
    JitKernels k;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = load_module(code, &k);
    if (e != hipSuccess) { if (k.module) (void)hipModuleUnload(k.module); return e; }
    g_cache[CacheKey{device, hash}] = k;
    *out = k;
    return hipSuccess;
}

hipError_t jit_launch_render(const JitKernels& k, const RenderParams& prm, unsigned blocks, hipStream_t stream) {
    RenderParams copy = prm;
    void* args[] = { &copy };
    return hipModuleLaunchKernel(k.render, blocks, 1, 1, RM_BLOCK_THREADS, 1, 1, 0, stream, args, nullptr);
}

hipError_t jit_launch_eval_points(const JitKernels& k, const RenderParams& prm, const float* xyz, uint32_t count,
                                  float* dist_out, float* norm_out, hipStream_t stream) {
    RenderParams copy = prm;
    void* args[] = { &copy, &xyz, &count, &dist_out, &norm_out };
    const unsigned blocks = (count + RM_BLOCK_THREADS - 1) / RM_BLOCK_THREADS;
    return hipModuleLaunchKernel(k.eval_points, blocks, 1, 1, RM_BLOCK_THREADS, 1, 1, 0, stream, args, nullptr);
}

}  // namespace rm
