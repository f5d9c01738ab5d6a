// shell_system.hpp -- host-side mirror of the reference's program surface above the C ABI.
//
// The reference is a libMesh application: parameters are parsed into globals
// (fem-shell.h:42-52), an EquationSystems object owns a LinearImplicitSystem "Elasticity"
// with the callback assemble_elasticity attached (fem-shell.cpp:70-85), and
// equation_systems.solve() runs callback + Krylov solve (fem-shell.cpp:138).  libMesh, PETSc
// and preCICE do not exist in this image, so this C++ layer keeps the same names, argument
// meaning and error behaviour on top of libfemshell:
//
//   reference                                   here
//   read_parameters(argc, argv)   SA:194-267    femshell_host::read_parameters
//   mesh.read + "_f" file         SA:35-67      femshell_host::read_xda / read_forces
//   DirichletBoundary {0,20},{1,21} SA:90-120   ShellMesh::dirichlet_mask
//   assemble_elasticity(es, name) SA:1160-1233  ShellSystem::assemble_elasticity
//   equation_systems.solve()      SA:138        ShellSystem::solve
//   build_solution_vector(sols)   SA:141        ShellSystem::build_solution_vector
//
// The adaptor a libMesh build would use instead (same C ABI underneath) is in
// libmesh_adaptor.hpp and INTEGRATION.md.
#pragma once

#include <iosfwd>
#include <string>
#include <vector>

#include "femshell.h"
#include "mesh_io.hpp"

namespace femshell_host {

// fem-shell.h:42-52 (same names)
struct Parameters {
    std::string in_filename;
    std::string out_filename;
    bool debug = false;
    double nu = 0.3;
    double em = 1.0e6;
    double thickness = 1.0;
    bool isOutfileSet = false;
    // extensions (not in the reference): solver controls that libMesh/PETSc take from
    // equation_systems.parameters / -ksp_* options
    double tol = 1e-12;       // libMesh "linear solver tolerance" default (TOLERANCE^2)
    int max_it = 100000;
};

// Same flags, defaults, messages and return convention as SA:194-267.
bool read_parameters(int argc, char **argv, Parameters &p, std::ostream &out, std::ostream &err);

struct SolveResult {
    unsigned int iterations = 0; // what LinearSolver::solve returns first
    double final_residual = 0.0; // ... and second
    bool converged = false;
    femshell_solve_info info{};
};

class ShellSystem {
  public:
    // throws std::runtime_error on any library error (libMesh code throws / asserts likewise)
    ShellSystem(const Parameters &p, int device = -1, int rank = 0, int world_size = 1, unsigned flags = FEMSHELL_REF_DEFAULT);
    ~ShellSystem();
    ShellSystem(const ShellSystem &) = delete;
    ShellSystem &operator=(const ShellSystem &) = delete;

    void comm_init(const unsigned char id[128]);
    // mesh + boundary ids + nodal forces (what main() sets up before init(), SA:35-125)
    void set_mesh(const ShellMesh &m);
    void set_forces(const std::vector<double> &f6); // n_nodes x 6, replaces the `forces` global
    // the assembly callback (SA:1160-1233); the name argument is checked like SA:1163
    void assemble_elasticity(const std::string &system_name = "Elasticity");
    // equation_systems.solve(): runs the callback if K is not current, then the Krylov solve
    SolveResult solve(double tol, int max_it);
    // sols[6*node + var] on every rank (SA:141, 163-169)
    const std::vector<double> &build_solution_vector();
    femshell_ctx *handle() { return ctx_; }

  private:
    femshell_ctx *ctx_ = nullptr;
    int n_nodes_ = 0;
    std::vector<double> sols_;
};

// The stand-alone program (SA:14-185); returns the process exit code.
int fem_shell_main(int argc, char **argv, std::ostream &out, std::ostream &err);

} // namespace femshell_host
