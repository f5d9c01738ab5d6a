// mesh_io.hpp -- host-side data formats either side of the hot path (C++17, no GPU code).
//
// Mirrors what the reference gets from libMesh and its own tools:
//   * ASCII XDA mesh files ("libMesh-0.7.0+" header, TRI3 = type 3, QUAD4 = type 5, side
//     boundary ids)                         -- mesh.read(), fem-shell.cpp:35-37;
//                                              format: doc/implementation.tex:76-130
//   * "<mesh>_f" nodal force files          -- fem-shell.cpp:44-67; doc/implementation.tex:131-146
//   * the structured generator meshGen      -- src/meshgen/main_all.cpp:15-390, including its
//     side-BC numbering (:283-338) and its n-1 force rows (:352, :377-384)
//   * boundary-id semantics                 -- fem-shell.cpp:90-120: ids 0/20 fix u,v,w,
//     ids 1/21 fix all six dofs, on BOTH nodes of every flagged element side
#pragma once

#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace femshell_host {

struct SideBC {
    int32_t elem; // element index in file order
    int32_t side; // side s joins local nodes s and (s+1) % n
    int32_t id;
};

struct ShellMesh {
    std::vector<double> xyz;      // n_nodes x 3
    std::vector<int32_t> tri;     // n_tri x 3
    std::vector<int32_t> quad;    // n_quad x 4
    std::vector<std::pair<char, int32_t>> order; // file order: ('t'|'q', index into tri/quad)
    std::vector<SideBC> bcs;
    std::vector<std::pair<int32_t, int32_t>> node_bcs; // (node, boundary id): Gmsh point elements (doc/implementation.tex:103-124)
    std::vector<double> loads;    // n_nodes x 6 (already scaled by the file's factor)

    int32_t n_nodes() const { return (int32_t)(xyz.size() / 3); }
    int32_t n_tri() const { return (int32_t)(tri.size() / 3); }
    int32_t n_quad() const { return (int32_t)(quad.size() / 4); }
    // nodes of element e in file order
    std::vector<int32_t> element_nodes(int32_t e) const;
    // per-node Dirichlet bit mask (bit v = dof v fixed)
    std::vector<uint8_t> dirichlet_mask() const;
    // nodes on sides carrying one of the given ids (coupling interface: 2, 20, 21), ascending
    std::vector<int32_t> nodes_with_ids(const std::vector<int32_t> &ids) const;
};

// Throw std::runtime_error with a message on malformed input.
ShellMesh read_xda(const std::string &path);
// Gmsh ASCII format 2.x as libMesh's importer treats it (doc/implementation.tex:103-124): triangles (type 2) and
// quadrangles (type 3) are the mesh; lower-dimensional elements define boundary conditions through their first tag
// (the physical entity): 2-node lines (type 1) flag the element side they coincide with, points (type 15) the node.
ShellMesh read_msh(const std::string &path);
// binary XDR form of the XDA records (fem-shell.cpp:35-37, :203: "*.xda/*.xdr"): big-endian 32-bit integers and IEEE
// doubles, length-prefixed padded strings, headers libMesh-0.7.0+ and libMesh-0.9.2+ (the latter adds nodesets).
// write_xdr is the writer the round-trip test uses; no libMesh-written file exists here to pin the layout (mesh_io.cpp)
ShellMesh read_xdr(const std::string &path);
void write_xdr(const ShellMesh &m, const std::string &path);
// what mesh.read(in_filename) does for the formats the reference program documents (fem-shell.cpp:37, :203): *.xda,
// *.xdr and *.msh, chosen by extension
ShellMesh read_mesh(const std::string &path);
// Reads "<n> <factor> n x 6"; rows missing at the end stay zero (the reference's stream
// extraction leaves them zero, fem-shell.cpp:59-66).  Returns n_nodes x 6, scaled.
std::vector<double> read_forces(const std::string &path, int32_t n_nodes);
// "<mesh>.xda" -> "<mesh>_f" (fem-shell.cpp:45-50)
std::string force_file_name(const std::string &mesh_path);

// precision: significant digits of the coordinates; the reference's meshGen prints through a default ostream
// (6 digits, main_all.cpp:226-339), which is the default here so that the twin's files are byte-identical
void write_xda(const ShellMesh &m, const std::string &path, int precision = 6);

struct MeshGenArgs {
    char type = 't';         // 't' | 'q'
    int nx = 1, ny = 1;
    double min_x = 0, min_y = 0, max_x = 1, max_y = 1;
    int bc_top = -1, bc_bottom = -1, bc_left = -1, bc_right = -1;
    double factor = 1.0;
    int loading = 0;         // 0 none, 1 unit load on node n_nodes/2, 2 uniform
    bool ul_lr = true;
    char dead_axis = 'z';
    bool meshgen_quirk = true; // write/apply only n_nodes-1 force rows like the reference tool
    int precision = 6;         // digits of the coordinates in the .xda file (FEMSHELL_MESHGEN_PRECISION=17 for round-trip exact files)
};
ShellMesh generate_structured(const MeshGenArgs &a);
// writes "<name>.xda" and, if loading > 0, "<name>_f" in the reference tool's format
void write_meshgen_files(const MeshGenArgs &a, const std::string &name);

// the reference's output (fem-shell.cpp:1240-1251, "<out>.e"): ExodusII file of the displaced mesh with the nodal variables
// u, v, w, tx, ty, tz, written as a classic netCDF (CDF-2) file by hand -- the image has no netCDF library (mesh_io.cpp)
void write_exodus(const ShellMesh &m, const std::vector<double> &u6, const std::string &path);
// legacy-VTK dump of the same content (kept beside the ExodusII file: every viewer reads it)
void write_vtk(const ShellMesh &m, const std::vector<double> &u6, const std::string &path);
// <stem>.pvtu + <stem>_0.vtu: the VTK XML pair libMesh's VTKIO writes per time step of a serial coupled run
// (fem-shell_precice.cpp:1552-1559); path must end in .pvtu
void write_pvtu(const ShellMesh &m, const std::vector<double> &u6, const std::string &path);

} // namespace femshell_host
