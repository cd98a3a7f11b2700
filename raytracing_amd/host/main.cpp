// main.cpp - headless front end: the reference's `--width/--height/--input`
// command line (Source/main.cpp:12-39) driving the HIP backend through the same
// call sequence (Source/main.cpp:67-138, 261-270, 292-293), with a PPM dump in
// the place of the SDL window.
//
//   rm_render --width W --height H [--input scene.sce] [--frames K] [--drag-x PX] [--drag-y PX] [--script events.txt]
//             [--out frame.ppm] [--device D] [--gpus N] [--devices D0,D1,...] [--frame-by-frame]
//
// The cameras of a run are known before its first frame (--frames K with a fixed drag per frame, or a script), so
// the front end works the whole camera path out first and hands it to the backend in one call
// (hip::raymarching::ProcessBatch: launches of several frames, rays.hpp); --frame-by-frame renders it as the
// reference's loop does, one SetInfo and one Process() per frame (same pixels, 2x the time per frame).
//
// --gpus N: the frame is dealt to N GPUs in 8-row bands (hip::raymarching over several devices, rays.hpp; D,
// D+1, ... unless --devices names them, and a device named twice gets two contexts); every frame is rendered by
// all of them and the PPM is the frame put together from their rows.
//
// --script: the reference's event loop (Source/main.cpp:144-270) driven from a file, one event per line:
//   frame | left | right | space | pause | resume | wheel N | pan DX DY | drag DX DY
// (SDLK_LEFT/RIGHT/SPACE/p, the mouse wheel, mouse motion with the middle or the right button down);
// every `frame` line is one pass of the loop body: the light turns unless paused, SetInfo, Process.
//
// Without --input the scene hard-coded in Source/main.cpp:86-123 is built with
// the same create_from calls.  Without --width/--height the reference takes the
// display size (Source/main.cpp:46-47); headless, 1920x1080 stands in for it.
// As in the reference a size flag given twice toggles itself off again
// (`correctInit ^= ...`, Source/main.cpp:25,28) and falls back to that default.
#include <bitset>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "rays.hpp"

constexpr float M_SQRT1_5f{ .447213595499957939281f };   // Source/main.cpp:7
#ifndef M_PI_2f
constexpr float M_PI_2f{ 1.57079632679489661923f };
#endif

using namespace std;

static int write_ppm(const string& path, const vector<unsigned char>& rgba, size_t w, size_t h) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return -1;
    fprintf(f, "P6\n%zu %zu\n255\n", w, h);
    vector<unsigned char> row(w * 3);
    for (size_t y = 0; y < h; ++y) {
        for (size_t x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) row[x * 3 + c] = rgba[(y * w + x) * 4 + c];
        fwrite(row.data(), 1, row.size(), f);
    }
    return fclose(f);
}

int main(int argc, char** argv) {
    size_t Width = 0, Height = 0, Frames = 1;
    int DragX = 0, DragY = 0;          // scripted right-button drag per frame, pixels (Source/main.cpp:240-251)
    int Device = 0, Gpus = 1;
    bool FrameByFrame = false;
    for (int I = 1; I < argc; ++I) if (string(argv[I]) == "--frame-by-frame") FrameByFrame = true;
    string File, Out, Script, Devices;

    string arg;
    bitset<3> correctInit;
    for (int I = 1; I < argc; ++I) {
        arg = argv[I];
        if (arg.substr(0, 2) != "--" || I + 1 >= argc) continue;
        arg = arg.substr(2);
        if (arg == "width")        { Width = atoi(argv[++I]);  if (Width)  correctInit ^= 1 << 0; }
        else if (arg == "height")  { Height = atoi(argv[++I]); if (Height) correctInit ^= 1 << 1; }
        else if (arg == "input")   { File = argv[++I];         if (File.size()) correctInit ^= 1 << 2; }
        else if (arg == "frames")  { Frames = strtoul(argv[++I], nullptr, 10); }
        else if (arg == "drag-x")  { DragX = atoi(argv[++I]); }
        else if (arg == "drag-y")  { DragY = atoi(argv[++I]); }
        else if (arg == "out")     { Out = argv[++I]; }
        else if (arg == "script")  { Script = argv[++I]; }
        else if (arg == "device")  { Device = atoi(argv[++I]); }
        else if (arg == "gpus")    { Gpus = atoi(argv[++I]); }
        else if (arg == "devices") { Devices = argv[++I]; }
    }
    if (!correctInit[0]) Width = 1920;
    if (!correctInit[1]) Height = 1080;

    vector<int> devices;
    if (!Devices.empty()) {
        istringstream list(Devices);
        for (string d; getline(list, d, ',');) {
            char* end = nullptr;
            long v = strtol(d.c_str(), &end, 10);
            if (d.empty() || *end || v < 0) { cerr << "--devices: '" << d << "' is not a device number" << endl; return 1; }
            devices.push_back(int(v));
        }
        if (Gpus > 1 && size_t(Gpus) != devices.size()) {
            cerr << "--gpus " << Gpus << " but --devices names " << devices.size() << endl;
            return 1;
        }
    } else {
        if (Gpus < 1) { cerr << "--gpus " << Gpus << ": at least one" << endl; return 1; }
        for (int g = 0; g < Gpus; ++g) devices.push_back(Device + g);
    }

    null::raymarching* HIP = new hip::raymarching(devices);
    hip::raymarching* backend = static_cast<hip::raymarching*>(HIP);
    if (int rc = HIP->Init(Width, Height)) {
        cerr << "Init failed: " << rc << " (" << rm_status_string(rc) << ")" << endl;
        return 1;
    }
    HIP->SetTexture(0);

    if (correctInit[2]) {
        if (int rc = backend->LoadScene(File.c_str())) {
            cerr << "cannot load " << File << ": " << rc << " (" << rm_status_string(rc) << ")" << endl;
            return 1;
        }
    } else {
        // Source/main.cpp:86-123
        point d{ 1.f, 30.f, 1.f };
        float alpha = -1.8f, w = cosf(alpha / 2.f), r = sinf(alpha / 2.f) / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
        HIP->ReservePrimitives(RAYS_BLOCK_2D_x * RAYS_BLOCK_2D_y * RAYS_PRIMITIVES_PER_THREAD);
        HIP->AddPrimitive(primitives::senfina_ripeto::create_from({ 1, { 0.f, 500.f, 100.f } }));
        HIP->AddPrimitive(primitives::movo::create_from({ 1, { 200.f, 0.f, 0.f } }));
        HIP->AddPrimitive(primitives::rotacioQ::create_from({ 1, w, { r * d.x, r * d.y, r * d.z } }));
        HIP->AddPrimitive(primitives::komunajo_3::create_from({ { 1, 2, 5 } }));
        HIP->AddPrimitive(primitives::kubo::create_from({ { 50.f, 50.f, 50.f } }));
        HIP->AddPrimitive(primitives::komplemento::create_from({ 1 }));
        HIP->AddPrimitive(primitives::movo::create_from({ 1, { 0.f, 0.f, -50.f } }));
        HIP->AddPrimitive(primitives::sfero::create_from({ 60.f }));
        HIP->AddPrimitive(primitives::komplemento::create_from({ 1 }));
        HIP->AddPrimitive(primitives::movo::create_from({ 1, { 0.f, 0.f, 50.f } }));
        HIP->AddPrimitive(primitives::sfero::create_from({ 40.f }));
        if (int rc = HIP->SetPrimitives()) {
            cerr << "SetPrimitives failed: " << rc << " (" << rm_status_string(rc) << ")" << endl;
            return 1;
        }
    }

    // Source/main.cpp:127-138
    float scale = powf(2.f, -6.1f), theta = 0.f, cos_theta = 1.f, sin_theta = 0.f, phi = 0.f, cos_phi = 1.f, sin_phi = 0.f;
    float cos_1 = cosf(M_PI / 180.f), sin_1 = sinf(M_PI / 180.f), deg = M_PI / 360.f;
    rays_info Info_h{};
    Info_h.Depth = 1000;
    Info_h.LightSource = point{ -2.f * M_SQRT1_5f, 0.f, M_SQRT1_5f };
    Info_h.StartPos = point{ 0.f, 0.f, 0.f };
    Info_h.StartDir = point{ scale * cos_theta * cos_phi, scale * cos_theta * sin_phi, scale * sin_theta };
    Info_h.StartWVec = point{ scale * sin_phi, -scale * cos_phi, 0.f };
    Info_h.StartHVec = point{ scale * sin_theta * cos_phi, scale * sin_theta * sin_phi, -scale * cos_theta };
    HIP->SetInfo(Info_h);

    double kernel_ms = 0.0;
    auto t0 = chrono::steady_clock::now();
    vector<rays_info> path;                                      // the camera blocks of the frames, in order
    // one pass of the reference's loop body (Source/main.cpp:269-270): now, or collected for ProcessBatch
    auto render = [&]() -> bool {
        if (!FrameByFrame) { path.push_back(Info_h); return true; }
        HIP->SetInfo(Info_h);
        HIP->Process();
        if (backend->LastStatus()) {
            cerr << "\nProcess failed: " << backend->LastStatus() << " (" << rm_status_string(backend->LastStatus()) << ")" << endl;
            return false;
        }
        kernel_ms += backend->ProcessTime();
        return true;
    };
    auto set_vectors = [&]() {                                   // Source/main.cpp:253-255
        Info_h.StartDir = point{ scale * cos_theta * cos_phi, scale * cos_theta * sin_phi, scale * sin_theta };
        Info_h.StartWVec = point{ scale * sin_phi, -scale * cos_phi, 0.f };
        Info_h.StartHVec = point{ scale * sin_theta * cos_phi, scale * sin_theta * sin_phi, -scale * cos_theta };
    };
    if (!Script.empty()) {
        ifstream in(Script);
        if (!in) { cerr << "cannot read " << Script << endl; return 1; }
        bool pause = false;
        Frames = 0;
        string line;
        while (getline(in, line)) {
            line = line.substr(0, line.find('#'));
            istringstream tok(line);
            string what;
            int a = 0, b = 0;
            if (!(tok >> what)) continue;
            tok >> a >> b;
            if (what == "left") {                                 // Source/main.cpp:158-164
                Info_h.LightSource = point{
                    cos_1 * Info_h.LightSource.x - sin_1 * Info_h.LightSource.y,
                    sin_1 * Info_h.LightSource.x + cos_1 * Info_h.LightSource.y,
                    Info_h.LightSource.z };
            } else if (what == "right") {                         // :165-171
                Info_h.LightSource = point{
                    cos_1 * Info_h.LightSource.x + sin_1 * Info_h.LightSource.y,
                    -sin_1 * Info_h.LightSource.x + cos_1 * Info_h.LightSource.y,
                    Info_h.LightSource.z };
            } else if (what == "space") {                         // :172-186
                scale = powf(2.f, -2.1f);
                theta = 0.f; cos_theta = 1.f; sin_theta = 0.f;
                phi = 0.f; cos_phi = 1.f; sin_phi = 0.f;
                Info_h.StartPos = point{ 0.f, 0.f, 0.f };
                set_vectors();
            } else if (what == "pause") {                         // :187-188
                pause = true;
            } else if (what == "resume") {                        // :195-196
                pause = false;
            } else if (what == "wheel") {                         // :219-226
                scale *= powf(2.f, float(a) * .1f);
                set_vectors();
            } else if (what == "pan") {                           // :231-238
                float S = 2.f / scale;
                Info_h.StartPos.x -= float(a) * Info_h.StartWVec.x * S;
                Info_h.StartPos.y -= float(a) * Info_h.StartWVec.y * S;
                Info_h.StartPos.z -= float(a) * Info_h.StartWVec.z * S;
                Info_h.StartPos.x -= float(b) * Info_h.StartHVec.x * S;
                Info_h.StartPos.y -= float(b) * Info_h.StartHVec.y * S;
                Info_h.StartPos.z -= float(b) * Info_h.StartHVec.z * S;
                set_vectors();
            } else if (what == "drag") {                          // :239-255
                theta += float(b) * deg;
                if (theta > M_PI_2f) theta = M_PI_2f;
                else if (theta < -M_PI_2f) theta = -M_PI_2f;
                cos_theta = cosf(theta);
                sin_theta = sinf(theta);
                phi += float(a) * deg;
                cos_phi = cosf(phi);
                sin_phi = sinf(phi);
                set_vectors();
            } else if (what == "frame") {                         // :261-270
                if (!pause)
                    Info_h.LightSource = point{
                        cos_1 * Info_h.LightSource.x - sin_1 * Info_h.LightSource.y,
                        sin_1 * Info_h.LightSource.x + cos_1 * Info_h.LightSource.y,
                        Info_h.LightSource.z };
                if (!render()) return 1;
                ++Frames;
            } else {
                cerr << Script << ": unknown event '" << what << "'" << endl;
                return 1;
            }
        }
        if (Frames == 0) { cerr << Script << ": no frame line" << endl; return 1; }
    } else
    for (size_t frame = 0; frame < Frames; ++frame) {
        auto f0 = chrono::steady_clock::now();
        // Source/main.cpp:240-255, scripted: from the second frame on the mouse has moved (DragX, DragY)
        // pixels with the right button down since the frame before (--drag-x 1)
        if (frame > 0 && (DragX || DragY)) {
            theta += float(DragY) * deg;
            if (theta > M_PI_2f) theta = M_PI_2f;
            else if (theta < -M_PI_2f) theta = -M_PI_2f;
            cos_theta = cosf(theta);
            sin_theta = sinf(theta);
            phi += float(DragX) * deg;
            cos_phi = cosf(phi);
            sin_phi = sinf(phi);
            Info_h.StartDir = point{ scale * cos_theta * cos_phi, scale * cos_theta * sin_phi, scale * sin_theta };
            Info_h.StartWVec = point{ scale * sin_phi, -scale * cos_phi, 0.f };
            Info_h.StartHVec = point{ scale * sin_theta * cos_phi, scale * sin_theta * sin_phi, -scale * cos_theta };
        }
        // Source/main.cpp:261-266: the light turns one degree before every render
        Info_h.LightSource = point{
            cos_1 * Info_h.LightSource.x - sin_1 * Info_h.LightSource.y,
            sin_1 * Info_h.LightSource.x + cos_1 * Info_h.LightSource.y,
            Info_h.LightSource.z };
        if (!render()) return 1;
        if (FrameByFrame) {
            auto f1 = chrono::steady_clock::now();
            cout << "\rFrame time : " << chrono::duration_cast<chrono::milliseconds>(f1 - f0).count()
                 << "ms  \t\tTask execution time : " << size_t(backend->ProcessTime()) << "ms  " << flush;   // Source/main.cpp:287
        }
    }
    if (!path.empty()) {
        backend->ProcessBatch(path.data(), path.size());
        if (backend->LastStatus()) {
            cerr << "ProcessBatch failed: " << backend->LastStatus() << " (" << rm_status_string(backend->LastStatus()) << ")" << endl;
            return 1;
        }
        kernel_ms = backend->ProcessTime();
    }
    double secs = chrono::duration<double>(chrono::steady_clock::now() - t0).count();
    cout << endl << Frames << " frame(s) " << Width << "x" << Height << " on " << backend->Gpus() << " GPU context(s): "
         << (Width * Height * Frames / secs / 1e6) << " Mrays/s wall, kernel " << (kernel_ms / Frames) << " ms/frame" << endl;

    if (!Out.empty()) {
        vector<unsigned char> rgba;
        if (backend->ReadPixels(rgba) || write_ppm(Out, rgba, Width, Height)) {
            cerr << "cannot write " << Out << endl;
            return 1;
        }
    }
    HIP->UnsetTexture();
    HIP->Quit();
    delete HIP;
    return 0;
}
