// meshConvert: reads a mesh in any format the solver programs accept (*.xda ASCII, *.xdr binary, *.msh Gmsh 2.x) and
// writes it as *.xda or *.xdr, chosen by extension -- the part of libMesh's `meshtool -i in -o out` that the reference's
// users need to move between the two libMesh formats fem-shell.cpp:35-37 reads.  usage: meshConvert in out [digits]
// An output named *.e is an ExodusII file of the mesh (write_exodus, the format of the solver's -out file) with zero
// nodal fields, or, with a third argument "ramp", the fields u_v(node) = 1e-3 (node + 1)(v + 1): what the CPU test of
// the writer reads back.
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

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
        else if (ends_with(".e")) {
            std::vector<double> u((size_t)m.n_nodes() * 6, 0.0);
            if (argc == 4 && std::string(argv[3]) == "ramp")
                for (int32_t n = 0; n < m.n_nodes(); n++)
                    for (int v = 0; v < 6; v++) u[6 * (size_t)n + v] = 1e-3 * (n + 1) * (v + 1);
            femshell_host::write_exodus(m, u, out);
        }
        else throw std::runtime_error("output must be *.xda, *.xdr or *.e");
    } catch (const std::exception &e) {
        std::cerr << "ERROR: " << e.what() << "\n";
        return -1;
    }
    return 0;
}
