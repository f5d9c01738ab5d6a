// meshGen: twin of the reference's src/meshgen/main_all.cpp (same 13 positional arguments).
#include <cstdlib>
#include <iostream>
#include <string>

#include "mesh_io.hpp"

int main(int argc, char **argv)
{
    if (argc != 14) {
        std::cout << "usage: " << argv[0]
                  << " type nx ny min_x min_y max_x max_y bcids factor loading ul_lr dead-axis filename\n"
                  << "type: Q|q for Quad-4, T|t for Tri-3\n"
                  << "nx, ny: no. elements on primary / secondary axis\n"
                  << "min_x min_y max_x max_y: extent on the primary / secondary axis\n"
                  << "bcids: comma-separated boundary ids top,bottom,left,right (-1 = none), e.g. 2,0,20,21\n"
                  << "factor: global factor on all force entries\n"
                  << "loading: 0 none, 1 unit load on the central node, 2 uniform load\n"
                  << "ul_lr: 1 = hypotenuse faces the lower right corner, 0 = rotated by 90 degrees\n"
                  << "dead-axis: x|y|z\n"
                  << "filename: mesh name without extension\n";
        return -1;
    }
    femshell_host::MeshGenArgs a;
    a.type = argv[1][0] == 'Q' ? 'q' : (argv[1][0] == 'T' ? 't' : argv[1][0]);
    a.nx = std::atoi(argv[2]);
    a.ny = std::atoi(argv[3]);
    a.min_x = std::atof(argv[4]);
    a.min_y = std::atof(argv[5]);
    a.max_x = std::atof(argv[6]);
    a.max_y = std::atof(argv[7]);
    {
        std::string s = argv[8];
        int ids[4] = {-1, -1, -1, -1};
        size_t pos = 0;
        for (int i = 0; i < 4; i++) {
            const size_t c = s.find(',', pos);
            const std::string tok = s.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
            if (!tok.empty()) ids[i] = std::atoi(tok.c_str());
            if (c == std::string::npos) break;
            pos = c + 1;
        }
        a.bc_top = ids[0];
        a.bc_bottom = ids[1];
        a.bc_left = ids[2];
        a.bc_right = ids[3];
    }
    a.factor = std::atof(argv[9]);
    a.loading = std::atoi(argv[10]);
    a.ul_lr = std::atoi(argv[11]) == 1;
    a.dead_axis = argv[12][0];
    if (const char *e = std::getenv("FEMSHELL_MESHGEN_PRECISION")) a.precision = std::atoi(e) > 0 ? std::atoi(e) : 6;
    try {
        femshell_host::write_meshgen_files(a, argv[13]);
    } catch (const std::exception &e) {
        std::cout << e.what() << "\n";
        return -1;
    }
    return 0;
}
