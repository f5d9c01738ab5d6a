// meshConvert: reads a mesh in any format the solver programs accept (*.xda ASCII, *.xdr binary, *.msh Gmsh 2.x) and
// writes it as *.xda or *.xdr, chosen by extension -- the part of libMesh's `meshtool -i in -o out` that the reference's
// users need to move between the two libMesh formats fem-shell.cpp:35-37 reads.  usage: meshConvert in out [digits]
#include <cstdlib>
#include <iostream>
#include <string>

#include "mesh_io.hpp"

int main(int argc, char **argv)
{
    if (argc < 3 || argc > 4) {
        std::cerr << "usage: " << argv[0] << " in.{xda,xdr,msh} out.{xda,xdr} [significant digits of an ASCII output, default 17]\n";
        return -1;
    }
    const std::string out = argv[2];
    auto ends_with = [&](const char *ext) { const std::string e(ext); return out.size() >= e.size() && out.compare(out.size() - e.size(), e.size(), e) == 0; };
    try {
        const femshell_host::ShellMesh m = femshell_host::read_mesh(argv[1]);
        if (ends_with(".xdr")) femshell_host::write_xdr(m, out);
        else if (ends_with(".xda")) femshell_host::write_xda(m, out, argc == 4 ? std::atoi(argv[3]) : 17);
        else throw std::runtime_error("output must be *.xda or *.xdr");
    } catch (const std::exception &e) {
        std::cerr << "ERROR: " << e.what() << "\n";
        return -1;
    }
    return 0;
}
