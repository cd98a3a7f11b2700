// rays.hpp - host-side mirror of the reference's backend interface, so that the
// call sequence of its front end (Source/main.cpp:67-138,269-270,292-293) compiles
// against this backend unchanged:
//
//   null::raymarching   abstract interface            (Include/rays.h:15-69)
//   hip::raymarching    concrete backend over the C ABI, in the place of
//                       cuda::raymarching             (Include/cuda_rays.cuh:14-46)
//   primitives::TYPE::create_from({...})  node builders (Include/defines.h:17-27,
//                       Include/objects_list.h:8-30, Include/types.h:56-72)
//
// Same names, argument meaning and return conventions: every method returns 0 on
// success or the backend's status, except Process(), which returns the kernel
// time in whole milliseconds (Include/rays.h:31-35, Source/cuda_rays.cu:215).
// SetTexture takes a GL texture name in the reference; this backend is headless,
// so it accepts the name and renders to its own linear RGBA8 framebuffer, which
// ReadPixels() copies out.
#pragma once
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/raymarching_amd.h"

// ---- data model (Include/types.h:3-19) ----------------------------------------
typedef unsigned char raw_byte;
typedef long long     coord;
#ifdef RM_COUNTER_32
typedef int           counter;   // the reference's own LLP64 target: 4-byte child links
#else
typedef long          counter;   // Include/types.h:5 as it compiles on LP64
#endif
typedef float         scalar;
typedef rm_point      point;
typedef rm_ray        ray;
typedef rm_rays_info  rays_info;

#define RAYS_BLOCK_2D_x             16   // Include/defines.h:82-87: main.cpp:89 sizes its
#define RAYS_BLOCK_2D_y             8    // reservation with these
#define RAYS_PRIMITIVES_PER_THREAD  2

namespace primitives {

enum object_type : uint32_t {            // Include/types.h:29-54
    type_nenio = RM_TYPE_NENIO,
    type_portanta_sfero = RM_TYPE_PORTANTA_SFERO, type_sfero = RM_TYPE_SFERO, type_kubo = RM_TYPE_KUBO,
    type_cilindro = RM_TYPE_CILINDRO, type_ebeno = RM_TYPE_EBENO,
    type_kunigajo_2 = RM_TYPE_KUNIGAJO_2, type_kunigajo_3 = RM_TYPE_KUNIGAJO_3, type_kunigajo_4 = RM_TYPE_KUNIGAJO_4,
    type_komunajo_2 = RM_TYPE_KOMUNAJO_2, type_komunajo_3 = RM_TYPE_KOMUNAJO_3, type_komunajo_4 = RM_TYPE_KOMUNAJO_4,
    type_komplemento = RM_TYPE_KOMPLEMENTO,
    type_glata_kunigajo_2 = RM_TYPE_GLATA_KUNIGAJO_2, type_glata_komunajo_2 = RM_TYPE_GLATA_KOMUNAJO_2,
    type_movo = RM_TYPE_MOVO, type_rotacioX = RM_TYPE_ROTACIOX, type_rotacioY = RM_TYPE_ROTACIOY,
    type_rotacioZ = RM_TYPE_ROTACIOZ, type_rotacioQ = RM_TYPE_ROTACIOQ, type_senfina_ripeto = RM_TYPE_SENFINA_RIPETO
};

// Include/types.h:56-61.  dist/norm are kept for layout only: this backend binds
// node behaviour while flattening, not through function pointers.
struct bazo {
    object_type type;
    raw_byte    data[RM_PRIMITIVE_PAYLOAD];
    void*       dist;
    void*       norm;
};
static_assert(sizeof(bazo) == RM_BAZO_BYTES, "bazo is a 48-byte record");

// One node type: its payload struct and the create_from builder.  Unlike
// Include/types.h:64-72 the unused payload bytes are zeroed, and a payload that
// does not fit the 24 bytes (kunigajo_4 / komunajo_4 with 8-byte counters) is a
// compile-time error instead of a silent overrun into `dist`.
template <object_type Type, typename Payload>
struct node_type {
    typedef Payload data_struct;
    static bazo create_from(data_struct data) {
        static_assert(sizeof(data_struct) <= RM_PRIMITIVE_PAYLOAD,
                      "payload exceeds PRIMITIVE_PAYLOAD: build with -DRM_COUNTER_32 for the *_4 combinators");
        bazo b;
        std::memset(&b, 0, sizeof b);
        b.type = Type;
        std::memcpy(b.data, &data, sizeof data);
        return b;
    }
};

// Include/objects_list.h:8-30
struct portanta_sfero_data { counter o; point t; scalar r; };
struct sfero_data          { scalar r; };
struct kubo_data           { point b; };
struct cilindro_data       { scalar r; scalar h; };
struct ebeno_data          { point n; };
template <int N> struct links_data { counter o[N]; };
struct komplemento_data    { counter o; };
struct glata_data          { counter o[2]; scalar k; };
struct movo_data           { counter o; point t; };
struct rotacio_data        { counter o; scalar cos_phi; scalar sin_phi; };
struct rotacioQ_data       { counter o; scalar q_w; point q; };
struct ripeto_data         { counter o; point a; };

typedef node_type<type_portanta_sfero, portanta_sfero_data> portanta_sfero;
typedef node_type<type_sfero, sfero_data>                   sfero;
typedef node_type<type_kubo, kubo_data>                     kubo;
typedef node_type<type_cilindro, cilindro_data>             cilindro;
typedef node_type<type_ebeno, ebeno_data>                   ebeno;
typedef node_type<type_kunigajo_2, links_data<2>>           kunigajo_2;
typedef node_type<type_kunigajo_3, links_data<3>>           kunigajo_3;
typedef node_type<type_kunigajo_4, links_data<4>>           kunigajo_4;
typedef node_type<type_komunajo_2, links_data<2>>           komunajo_2;
typedef node_type<type_komunajo_3, links_data<3>>           komunajo_3;
typedef node_type<type_komunajo_4, links_data<4>>           komunajo_4;
typedef node_type<type_komplemento, komplemento_data>       komplemento;
typedef node_type<type_glata_kunigajo_2, glata_data>        glata_kunigajo_2;
typedef node_type<type_glata_komunajo_2, glata_data>        glata_komunajo_2;
typedef node_type<type_movo, movo_data>                     movo;
typedef node_type<type_rotacioX, rotacio_data>              rotacioX;
typedef node_type<type_rotacioY, rotacio_data>              rotacioY;
typedef node_type<type_rotacioZ, rotacio_data>              rotacioZ;
typedef node_type<type_rotacioQ, rotacioQ_data>             rotacioQ;
typedef node_type<type_senfina_ripeto, ripeto_data>         senfina_ripeto;

}  // namespace primitives

namespace null {

// Include/rays.h:15-69
class raymarching {
protected:
    size_t Width = 0, Height = 0;
    std::string SceneName;
    std::vector<primitives::bazo> Primitives_h;
    size_t PrimitivesNum = 0;

public:
    virtual ~raymarching() {}
    virtual int Init(const size_t& width, const size_t& height) = 0;
    virtual int Process() = 0;
    virtual int Quit() = 0;
    virtual int SetInfo(const rays_info& info) = 0;
    virtual int SetTexture(unsigned int texture) = 0;
    virtual int UnsetTexture() = 0;
    virtual int SetPrimitives() = 0;
    virtual int SetRays() = 0;

    inline void SetSceneName(const char* name) { SceneName = name; }
    inline void ReservePrimitives(size_t new_capacity) { Primitives_h.reserve(new_capacity); PrimitivesNum = 0; }
    inline void AddPrimitive(primitives::bazo new_primitive) { Primitives_h.push_back(new_primitive); PrimitivesNum++; }
    inline void ClearPrimitives() { Primitives_h.clear(); PrimitivesNum = 0; }
};

}  // namespace null

namespace hip {

// One context per GPU.  With one device this is the reference's backend as it stands; with several, the frame is
// dealt to them in bands of RM_FRONT_END_BAND_ROWS rows, round-robin (rm_set_bands: the split SURVEY.md 8e asks
// for; the reference itself is single-device, Source/cuda_rays.cu:20-21).  Every GPU renders its own rows of
// every frame from the same scene and camera block, nothing is exchanged between them, and ReadPixels() puts
// the bands back in frame order on the host.  A device may be named more than once (several contexts on one
// GPU): that is how the path is tested on a one-GPU box.
constexpr size_t RM_FRONT_END_BAND_ROWS = 8;      // one row of tiles: the finest deal

class raymarching : public null::raymarching {
    std::vector<rm_ctx*> ctx_;
    std::vector<int> devices_;
    int last_ = RM_OK;
    float last_ms_ = 0.f;
    size_t batch_ = 1;         // frames the contexts' launches have room for (rm_set_batch)
    size_t last_slot_ = 0;     // the plane of the last launch that holds the frame rendered last

    int each(int (*call)(rm_ctx*)) {
        for (rm_ctx* c : ctx_)
            if ((last_ = call(c)) != RM_OK) return last_;
        return last_ = RM_OK;
    }
    void release() {
        for (rm_ctx* c : ctx_) if (c) rm_destroy(c);
        ctx_.clear();
    }

public:
    explicit raymarching(int device = 0) : devices_{ device } {}
    explicit raymarching(const std::vector<int>& devices) : devices_(devices.empty() ? std::vector<int>{ 0 } : devices) {}
    ~raymarching() override { release(); }

    int Init(const size_t& width, const size_t& height) override {
        Width = width; Height = height;
        release();
        const int n = static_cast<int>(devices_.size());
        for (int rank = 0; rank < n; ++rank) {
            rm_ctx* c = nullptr;
            if ((last_ = rm_create(devices_[rank], width, height, &c)) != RM_OK) { release(); return last_; }
            ctx_.push_back(c);
            if (n > 1 && (last_ = rm_set_bands(c, RM_FRONT_END_BAND_ROWS, rank, n)) != RM_OK) { release(); return last_; }
        }
        return last_ = RM_OK;
    }
    // whole milliseconds like the reference; ProcessTime() has the float.  Several GPUs: all are started
    // before any is waited for, and the time is the host's, from the first launch to the last GPU's end.
    int Process() override {
        if (ctx_.empty()) { last_ = RM_ERR_NOT_READY; return 0; }
        if (ctx_.size() == 1) last_ = rm_process(ctx_[0], &last_ms_);
        else {
            const auto t0 = std::chrono::steady_clock::now();
            if (each(rm_process_async) != RM_OK) return 0;
            if (each(rm_synchronize) != RM_OK) return 0;
            last_ms_ = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        return last_ == RM_OK ? static_cast<int>(static_cast<uint32_t>(last_ms_)) : 0;
    }
    int Quit() override {
        int rc = RM_OK;
        for (rm_ctx*& c : ctx_) { int r = c ? rm_destroy(c) : RM_OK; c = nullptr; if (rc == RM_OK) rc = r; }
        ctx_.clear();
        return last_ = rc;
    }
    int SetInfo(const rays_info& info) override {                     // SetRays is fused
        // (one frame for the next Process(), also after a ProcessBatch has made room for launches of several)
        for (rm_ctx* c : ctx_) if ((last_ = rm_set_info_batch(c, &info, 1)) != RM_OK) return last_;
        last_slot_ = 0;
        return last_ = ctx_.empty() ? RM_ERR_NOT_READY : RM_OK;
    }
    // Beyond the reference interface: the frames of a camera path that is known ahead (what Source/main.cpp:261-270
    // does K times: SetInfo, Process), `n` rays_info blocks rendered in launches of several frames each
    // (rm_set_batch): the tiles of a launch's frames are one pool for the persistent wavefronts, so the few long tiles
    // of one frame run beside the short ones of the others - a frame per Process() is as long as its longest ray
    // (chair.sce 3840x2160: 0.33 ms a frame one by one, 0.15 ms in launches of 20).  Same pixels as n SetInfo/Process
    // pairs; ReadPixels() then returns the LAST frame, as it would after them.  Returns the kernel time of all the
    // launches in whole milliseconds (ProcessTime(): the float); LastStatus() tells whether it worked.
    int ProcessBatch(const rays_info* infos, size_t n) {
        if (ctx_.empty()) { last_ = RM_ERR_NOT_READY; return 0; }
        if (!infos || n == 0) { last_ = RM_ERR_INVALID_ARG; return 0; }
        // A launch of ONE frame first: it learns in what order the tiles of this camera path should go out (the
        // costliest first) and how many wavefronts a launch needs, and a launch of F frames behind a launch of one
        // takes that order for each of its frames (raymarching_amd.h, rm_set_batch).  Then launches of nearly equal
        // size, the larger ones first (a launch of more frames than the one before it, and not a multiple, finds the
        // extra frames' tiles in no list), of at most 2^23 tiles per context: more gain nothing.
        const size_t tiles = ((Width + 7) / 8) * ((Height + 7) / 8) / ctx_.size() + 1;
        size_t cap = (size_t(1) << 23) / tiles;
        cap = cap < 1 ? 1 : cap > RM_MAX_BATCH ? RM_MAX_BATCH : cap;
        const size_t lead = (cap > 1 && n > 2) ? 1 : 0, rest = n - lead;
        const size_t launches = lead + (rest + cap - 1) / cap, small = rest / (launches - lead), larger = rest % (launches - lead);
        const size_t need = small + (larger ? 1 : 0);
        if (need > batch_) {                          // room is only ever added (rm_set_batch waits and may allocate)
            for (rm_ctx* c : ctx_) if ((last_ = rm_set_batch(c, need, 0, 0)) != RM_OK) return 0;
            batch_ = need;
        }
        last_ms_ = 0.f;
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t k = 0, first = 0; k < launches; ++k) {
            const size_t take = (lead && k == 0) ? 1 : small + (k - lead < larger ? 1 : 0);
            for (rm_ctx* c : ctx_) if ((last_ = rm_set_info_batch(c, infos + first, take)) != RM_OK) return 0;
            if (ctx_.size() == 1) {
                float ms = 0.f;
                if ((last_ = rm_process(ctx_[0], &ms)) != RM_OK) return 0;
                last_ms_ += ms;
            } else if (each(rm_process_async) != RM_OK || each(rm_synchronize) != RM_OK) return 0;
            first += take;
            last_slot_ = take - 1;
        }
        if (ctx_.size() > 1) last_ms_ = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return static_cast<int>(static_cast<uint32_t>(last_ms_));
    }
    int SetTexture(unsigned int) override {
        for (rm_ctx* c : ctx_) if ((last_ = rm_set_output(c, nullptr, 0)) != RM_OK) return last_;
        return last_ = RM_OK;
    }
    int UnsetTexture() override { return last_ = RM_OK; }
    int SetPrimitives() override {
        PrimitivesNum = Primitives_h.size();
        for (rm_ctx* c : ctx_)
            if ((last_ = rm_set_primitives(c, Primitives_h.data(), PrimitivesNum,
                                           sizeof(counter) == 8 ? RM_BAZO_LP64 : RM_BAZO_LLP64)) != RM_OK) return last_;
        return last_ = ctx_.empty() ? RM_ERR_NOT_READY : RM_OK;
    }
    int SetRays() override { return last_ = RM_OK; }   // rays are derived inside Process()

    // ---- beyond the reference interface: what a headless caller needs
    int LoadScene(const char* sce_path) {              // the `--input` file, Source/main.cpp:29-31
        size_t n = 0;
        if ((last_ = rm_load_sce(sce_path, nullptr, 0, &n)) != RM_OK) return last_;
        std::vector<rm_node> nodes(n);
        if ((last_ = rm_load_sce(sce_path, nodes.data(), n, &n)) != RM_OK) return last_;
        SceneName = sce_path;
        PrimitivesNum = n;
        for (rm_ctx* c : ctx_) if ((last_ = rm_set_nodes(c, nodes.data(), n)) != RM_OK) return last_;
        return last_ = ctx_.empty() ? RM_ERR_NOT_READY : RM_OK;
    }
    int ReadPixels(std::vector<unsigned char>& rgba) {
        rgba.resize(Width * Height * 4);
        if (ctx_.empty()) return last_ = RM_ERR_NOT_READY;
        if (ctx_.size() == 1) return last_ = rm_read_pixels_of_frame(ctx_[0], last_slot_, rgba.data(), Width * 4);
        const size_t n = ctx_.size(), row = Width * 4;
        std::vector<unsigned char> part;
        for (size_t rank = 0; rank < n; ++rank) {
            size_t rows = 0;
            if ((last_ = rm_local_rows(ctx_[rank], &rows)) != RM_OK) return last_;
            part.resize(rows * row);
            if (rows && (last_ = rm_read_pixels_of_frame(ctx_[rank], last_slot_, part.data(), row)) != RM_OK) return last_;
            // the rank's rows are its bands one after the other (include/raymarching_amd.h, rm_set_bands)
            size_t local = 0;
            for (size_t band = rank, y = rank * RM_FRONT_END_BAND_ROWS; y < Height; band += n, y += n * RM_FRONT_END_BAND_ROWS) {
                const size_t take = Height - y < RM_FRONT_END_BAND_ROWS ? Height - y : RM_FRONT_END_BAND_ROWS;
                std::memcpy(rgba.data() + y * row, part.data() + local * row, take * row);
                local += take;
            }
            if (local != rows) return last_ = RM_ERR_INVALID_ARG;
        }
        return last_ = RM_OK;
    }
    float ProcessTime() const { return last_ms_; }
    int LastStatus() const { return last_; }
    size_t Gpus() const { return ctx_.size(); }
    rm_ctx* Context(size_t rank = 0) { return rank < ctx_.size() ? ctx_[rank] : nullptr; }
};

}  // namespace hip
