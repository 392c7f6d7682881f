// The platform's own libstdc++ <random>, used exactly as the reference declares its two streams (main.cpp:91-92:
// `default_random_engine generator(seed); uniform_real_distribution<> distribution(-0.5f, 0.5f);`  material.h:17-19:
// `std::default_random_engine generator(seed); std::uniform_real_distribution<float> distribution(0.0, 1.0);`) -- test infrastructure:
// prints, for a seed, n raw engine values, n float draws and n double draws as bit patterns, each from a fresh engine.
// tests/test_oracle_known_answers.py compares them with the oracle's restatement of those streams (and, through
// tests/reference_restatements.py, with the numpy one).  libstdc++ is part of this image's toolchain, not of the reference.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>

int main(int argc, char **argv) {
    if (argc > 2 && std::strcmp(argv[1], "floats") == 0) {
        // `parse` mode: every `v` / `vn` / `Kd` / `Ke` / `Ks` / `Ns` number of a model or material file as `ifstream >> float` reads it
        // (scene.cpp:56-66, 73-82): the text-to-float conversion of the loader, done by the real library
        std::ifstream file(argv[2]);
        std::string input;
        while (!file.eof()) {
            file >> input;
            if (file.eof()) break;
            const int k = (input == "v" || input == "vn" || input == "Kd" || input == "Ke" || input == "Ks") ? 3 : input == "Ns" ? 1 : 0;
            for (int i = 0; i < k; ++i) {
                float v = 0;
                file >> v;
                uint32_t b;
                std::memcpy(&b, &v, 4);
                std::printf("%s %08x\n", input.c_str(), b);
            }
        }
        return 0;
    }
    const unsigned seed = argc > 1 ? static_cast<unsigned>(std::strtoul(argv[1], nullptr, 10)) : 42u;
    const int n = argc > 2 ? std::atoi(argv[2]) : 8;
    {
        std::default_random_engine generator(seed);
        for (int i = 0; i < n; ++i) std::printf("raw %lu\n", static_cast<unsigned long>(generator()));
    }
    {
        std::default_random_engine generator(seed);
        std::uniform_real_distribution<float> distribution(0.0, 1.0);
        for (int i = 0; i < n; ++i) {
            const float v = distribution(generator);
            uint32_t b;
            std::memcpy(&b, &v, 4);
            std::printf("unit %08x\n", b);
        }
    }
    {
        std::default_random_engine generator(seed);
        std::uniform_real_distribution<> distribution(-0.5f, 0.5f);
        for (int i = 0; i < n; ++i) {
            const double v = distribution(generator);
            uint64_t b;
            std::memcpy(&b, &v, 8);
            std::printf("jitter %016llx\n", static_cast<unsigned long long>(b));
        }
    }
    return 0;
}
