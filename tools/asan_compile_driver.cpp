// Host-side sanitizer run of the compiler (parser, lowering, passes, kernel generator): reads .mm texts separated by a
// line "====" from stdin and compiles each like runtime.cpp compile_source does, without the HIP runtime.  Built with
// g++ -fsanitize=address,undefined by tools/asan_compile.sh (sanitizers run on the CPU build only).
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>

#include "front.h"
#include "hipgen.h"
#include "passes.h"

using namespace mm;

int main() {
    std::stringstream all;
    all << std::cin.rdbuf();
    std::string text = all.str(), sep = "\n====\n";
    size_t pos = 0;
    int ok = 0, refused = 0;
    while (pos < text.size()) {
        size_t e = text.find(sep, pos);
        std::string src = text.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        pos = e == std::string::npos ? text.size() : e + sep.size();
        if (src.find("filter") == std::string::npos) continue;
        try {
            Module m;
            parse_module(m, src);
            auto code = lower_filter(m, m.main, nullptr);
            optimize(*code);
            analyze_frame_constants(*code);
            for (auto &sub : code->closure_renders) {
                optimize(*sub);
                eliminate_dead_cycles(*sub);
                analyze_frame_constants(*sub);
            }
            for (auto &fn : code->functions) optimize(*fn);
            KernelOptions ko;
            KernelSource ks = generate_hip(*code, ko);
            for (auto &sub : code->closure_renders) generate_hip(*sub, ko, code.get());
            ++ok;
        } catch (const std::exception &ex) {
            ++refused;
        }
    }
    printf("compiled %d, refused %d\n", ok, refused);
    return 0;
}
