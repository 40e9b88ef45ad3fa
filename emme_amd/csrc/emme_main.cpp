// emme_main.cpp -- command-line driver with the reference's file conventions
// (src/main.cpp:182-338): reads ./input.json, writes ./output.json and ./eigenMatrics/*.bin.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "../../include/emme_hip.h"

int main(int argc, char** argv) {
    const std::string in_path = argc > 1 ? argv[1] : "input.json";
    const std::string out_path = argc > 2 ? argv[2] : "output.json";
    const std::string mat_dir = argc > 3 ? argv[3] : "eigenMatrics";
    std::ifstream f(in_path);
    if (!f) {
        std::cerr << "cannot open " << in_path << "\n";
        return 1;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    char* out = nullptr;
    const int rc = emme_run_json(ss.str().c_str(), mat_dir.c_str(), &out);
    if (rc != EMME_OK) {
        std::cerr << "emme: " << emme_last_error() << "\n";
        return 2;
    }
    std::ofstream o(out_path);
    o << out;
    emme_free(out);
    return 0;
}
