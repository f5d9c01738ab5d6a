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
#include <utility>
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
    double tol = 1e-12;       // libMesh "linear solver tolerance" default (TOLERANCE^2); also -ksp_rtol
    int max_it = 5000;        // libMesh's "linear solver maximum iterations" default; also -ksp_max_it
    // the reference passes -ksp_type / -pc_type through to PETSc (doc/implementation.tex:68-72).  K is SPD and the
    // library's Krylov method is CG: -ksp_type cg is accepted, anything else is reported and replaced by cg;
    // -pc_type jacobi|bjacobi|pbjacobi|none -> 6x6 block-Jacobi, gamg|amg|ml|hypre|mg -> the multigrid preconditioner
    std::string ksp_type = "cg";
    // default: the multigrid.  The reference's default (PETSc's GMRES + ILU) is a stronger preconditioner than point-block
    // Jacobi, whose iteration count grows with the element count (no convergence at 4M triangles); -pc_type bjacobi opts in
    std::string pc_type = "gamg";
};

// One process per GPU.  How a rank learns its place: FEMSHELL_RANK / FEMSHELL_WORLD_SIZE, else the
// launcher's variables (torchrun: RANK / WORLD_SIZE / LOCAL_RANK; Open MPI: OMPI_COMM_WORLD_*; MPICH/Slurm: PMI_RANK /
// PMI_SIZE), else a single rank.  The 128-byte RCCL id travels through a file: rank 0 writes FEMSHELL_UID_FILE
// (default: $XDG_RUNTIME_DIR or /tmp/femshell-<uid>, 0700, file femshell_uid_<MASTER_PORT or parent pid>), the others wait for it -- the role MPI plays for the
// reference (LibMeshInit, fem-shell.cpp:28).
// FEMSHELL_TIMING=1: the wall time of the program's phases, one line on stderr when rank 0 is done (the reference's runs end
// with libMesh's performance log; this is the twin's short form of it)
class PhaseClock {
public:
    PhaseClock();
    void done(const std::string &phase); // closes the phase that ran since the last call (or since construction)
    void note(const std::string &text) { notes_.push_back(text); }
    void report(std::ostream &err) const;

private:
    bool enabled_ = false;
    double start_ = 0.0, last_ = 0.0;
    std::vector<std::pair<std::string, double>> phases_;
    std::vector<std::string> notes_;
};

struct Launch {
    int rank = 0, world_size = 1, device = -1;
    std::string uid_file;
    static Launch from_environment();
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

    // the ShellSystem of this process as the launch environment describes it: context on the rank's GPU, RCCL id
    // exchanged, preconditioner chosen from p.pc_type
    ShellSystem(const Parameters &p, const Launch &launch, unsigned flags = FEMSHELL_REF_DEFAULT);
    void comm_init(const unsigned char id[128]);
    int rank() const { return rank_; }
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
    void choose_preconditioner(const Parameters &p);
    femshell_ctx *ctx_ = nullptr;
    int n_nodes_ = 0, rank_ = 0;
    bool solved_once_ = false; // later solves start from the solution before (libMesh's initial guess: ShellSystem::solve)
    std::vector<double> sols_;
};

// The stand-alone program (SA:14-185); returns the process exit code.
int fem_shell_main(int argc, char **argv, std::ostream &out, std::ostream &err);

} // namespace femshell_host
