// main.cpp -- command-line renderer with the reference CLI's contract
// (reference src/main.cu:7-17): one optional argument, the scene file
// (default ../assets/scenes/default.json); the image goes to image.png.
// Extra flags select what the reference fixes at compile time.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>

#include "scene.hpp"

int main(int argc, char** argv) {
    std::string path = "../assets/scenes/default.json";
    std::string out = "image.png";
    size_t seed = 1;
    Scene s;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--direct") s.settings.indirect_lighting = false;
        else if (a == "--no-aa") s.settings.antialiasing = false;
        else if (a == "--no-bbox") s.settings.bounding_box = false;
        else if (a == "--depth" && i + 1 < argc) s.settings.max_ray_depth = std::atoi(argv[++i]);
        else if (a == "--seed" && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 10);
        else if (a == "-o" && i + 1 < argc) out = argv[++i];
        else if (a == "--help") {
            std::printf("usage: render [scene.json] [--direct] [--no-aa] [--no-bbox] [--depth N] [--seed N] [-o image.png]\n");
            return 0;
        } else path = a;
    }
    try {
        s.load_scene(path);
        s.render(out, seed);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
