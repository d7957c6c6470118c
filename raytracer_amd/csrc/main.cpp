// main.cpp -- command-line renderer with the reference CLI's contract
// (reference src/main.cu:7-17): one optional argument, the scene file
// (default ../assets/scenes/default.json); the image goes to image.png.
// Extra flags select what the reference fixes at compile time, and the frame
// loop its README describes (README.md:24) is built in: --frames N with
// --rotate / --move camera steps and the frame number added to the seed.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>

#include "scene.hpp"

namespace {

const char* kUsage =
    "usage: render [scene.json] [--direct] [--no-aa] [--no-bbox] [--exhaustive] [--fast] [--depth N] [--seed N]\n"
    "              [-o image.png|image.pfm] [--devices N]\n"
    "              [--frames N --rotate RX RY RZ --move DX DY DZ --first K --stride S --writers T]\n"
    "  with --frames, -o is a pattern such as frames/frame_%04d.png\n"
    "  --devices N splits every frame over HIP devices 0..N-1 of this node\n";

bool take3(int argc, char** argv, int& i, Vec3f& out) {
    if (i + 3 >= argc) return false;
    for (int k = 0; k < 3; k++) out[k] = std::strtof(argv[i + 1 + k], nullptr);
    i += 3;
    return true;
}

} // namespace

int main(int argc, char** argv) {
    std::string path = "../assets/scenes/default.json";
    std::string out = "image.png";
    size_t seed = 1;
    unsigned frames = 0, first = 0, stride = 1, writers = 4;
    Vec3f rotate, move;
    Scene s;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--direct") s.settings.indirect_lighting = false;
        else if (a == "--no-aa") s.settings.antialiasing = false;
        else if (a == "--no-bbox") s.settings.bounding_box = false;
        else if (a == "--exhaustive") s.settings.cluster_culling = false;
        else if (a == "--fast") s.settings.fast_arithmetic = true;
        else if (a == "--depth" && i + 1 < argc) s.settings.max_ray_depth = std::atoi(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 10);
        else if (a == "-o" && i + 1 < argc) out = argv[++i];
        else if (a == "--devices" && i + 1 < argc) s.settings.devices = static_cast<unsigned>(std::strtoul(argv[++i], nullptr, 10));
        else if (a == "--frames" && i + 1 < argc) frames = static_cast<unsigned>(std::strtoul(argv[++i], nullptr, 10));
        else if (a == "--first" && i + 1 < argc) first = static_cast<unsigned>(std::strtoul(argv[++i], nullptr, 10));
        else if (a == "--stride" && i + 1 < argc) stride = static_cast<unsigned>(std::strtoul(argv[++i], nullptr, 10));
        else if (a == "--writers" && i + 1 < argc) writers = static_cast<unsigned>(std::strtoul(argv[++i], nullptr, 10));
        else if (a == "--rotate") { if (!take3(argc, argv, i, rotate)) { std::fputs(kUsage, stderr); return 2; } }
        else if (a == "--move") { if (!take3(argc, argv, i, move)) { std::fputs(kUsage, stderr); return 2; } }
        else if (a == "--help") { std::fputs(kUsage, stdout); return 0; }
        else path = a;
    }
    try {
        s.load_scene(path);
        if (frames > 0) {
            if (out == "image.png") out = "frame_%04d.png";
            s.render_frames(out, seed, frames, rotate, move, first, stride, writers);
        } else {
            s.render(out, seed);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
