// libmesh_adaptor.hpp -- the binding a fem-shell maintainer adds to run the hot path on
// libfemshell from the ORIGINAL libMesh program.  It needs libMesh headers, which do not exist
// in this image, so everything is inside FEMSHELL_HAVE_LIBMESH; the same C-ABI calls are
// exercised by shell_system.cpp and the tests.  tests/test_host_tools.py compiles this header
// (-fsyntax-only) against tests/helpers/libmesh_mock, a test-only stand-in for the handful of
// libMesh declarations it touches -- a syntax check, NOT a reference build.  See INTEGRATION.md.
//
// Two hooks, matching the two drop-in boundaries of the reference:
//   (1) femshell_assemble_elasticity: same signature as assemble_elasticity
//       (fem-shell.h:75), registered with system.attach_assemble_function (fem-shell.cpp:85).
//       It extracts flat arrays from the EquationSystems, runs the device assembly and, in
//       "compat" mode, adds the device-assembled rows into system.matrix / system.rhs so that
//       any libMesh/PETSc solver can be used unchanged.
//   (2) FemShellLinearSolver: a libMesh::LinearSolver<Number> that keeps K in HBM and runs the
//       CG there; assigned to system.linear_solver before equation_systems.solve()
//       (fem-shell.cpp:138; fem-shell_precice.cpp:271).  The binding selects the multigrid
//       preconditioner unless FEMSHELL_PC says otherwise: equation_systems.solve() passes libMesh's
//       "linear solver tolerance" / "maximum iterations" (fem-shell.cpp:130-133 leaves them at the
//       library defaults, tight tolerance and 5000 iterations as far as the survey could tell), and
//       6x6 block-Jacobi CG needs 5010 iterations for 1e-12 on the reference's own 64x64 Test-G-family
//       panel (run_examples.sh:47-48) -- the multigrid 60.  get_converged_reason() reports what
//       femshell_solve_info says, DIVERGED_ITS when the iteration limit was hit.
#pragma once

#ifdef FEMSHELL_HAVE_LIBMESH

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "femshell.h"
#include "libmesh/boundary_info.h"
#include "libmesh/dof_map.h"
#include "libmesh/elem.h"
#include "libmesh/equation_systems.h"
#include "libmesh/linear_implicit_system.h"
#include "libmesh/linear_solver.h"
#include "libmesh/mesh_base.h"
#include "libmesh/numeric_vector.h"
#include "libmesh/sparse_matrix.h"

namespace femshell_libmesh {

using namespace libMesh;

// globals of the reference program (fem-shell.h:42-52) that the callback reads
extern Real nu, em, thickness;
extern std::vector<DenseVector<Real>> forces;

struct Binding {
    femshell_ctx *ctx = nullptr;
    bool mesh_sent = false;
    bool compat_copy_back = true; // add K,F into libMesh's matrix/rhs after the device assembly
    EquationSystems *es = nullptr; // set by the assembly callback; the solver hook maps node dofs through it
    femshell_solve_info last_info{}; // of the last FemShellLinearSolver::solve
    int last_rc = FEMSHELL_OK;
};
inline Binding &binding()
{
    static Binding b;
    return b;
}

inline void check(int rc)
{
    if (rc != FEMSHELL_OK) libmesh_error_msg(femshell_last_error());
}

// flat arrays from the libMesh objects the callback sees (fem-shell.cpp:1166-1205)
inline void send_mesh(EquationSystems &es)
{
    const MeshBase &mesh = es.get_mesh();
    Binding &b = binding();
    if (!b.ctx) {
        femshell_config cfg{};
        cfg.nu = nu;
        cfg.E = em;
        cfg.thickness = thickness;
        cfg.flags = FEMSHELL_REF_DEFAULT;
        cfg.device = -1;
        cfg.rank = (int32_t)mesh.processor_id();
        cfg.world_size = (int32_t)mesh.n_processors();
        check(femshell_create(&cfg, &b.ctx));
        if (!std::getenv("FEMSHELL_PC")) { // (femshell_create has applied FEMSHELL_PC=amg|jacobi when it is set)
            femshell_pc_options pc;
            check(femshell_pc_defaults(FEMSHELL_PC_AMG, &pc));
            check(femshell_set_preconditioner(b.ctx, &pc));
        }
        if (cfg.world_size > 1) {
            // mpirun -n N (Test G, run_examples.sh:47-48): one GPU per MPI rank; rank 0 creates the RCCL id, libMesh's
            // communicator distributes it (replaces LibMeshInit's MPI setup for the device path, fem-shell.cpp:28, 35)
            std::vector<uint8_t> id(128, 0);
            if (cfg.rank == 0) check(femshell_comm_unique_id(id.data()));
            mesh.comm().broadcast(id, 0);
            check(femshell_comm_init(b.ctx, id.data()));
        }
    }
    const dof_id_type n_nodes = mesh.n_nodes();
    std::vector<double> xyz(3 * n_nodes);
    for (const Node *nd : mesh.node_ptr_range())
        for (int d = 0; d < 3; d++) xyz[3 * nd->id() + d] = (*nd)(d);
    std::vector<int32_t> tri, quad;
    std::vector<uint8_t> mask(n_nodes, 0);
    const BoundaryInfo &bi = mesh.get_boundary_info();
    for (const Elem *elem : mesh.active_element_ptr_range()) { // replicated mesh: every rank sees all
        std::vector<int32_t> &dst = elem->type() == TRI3 ? tri : quad;
        for (unsigned i = 0; i < elem->n_nodes(); i++) dst.push_back((int32_t)elem->node_id(i));
        for (unsigned s = 0; s < elem->n_sides(); s++) {
            std::vector<boundary_id_type> ids;
            bi.boundary_ids(elem, s, ids);
            for (boundary_id_type id : ids) {
                const uint8_t m = (id == 0 || id == 20) ? 0x07 : ((id == 1 || id == 21) ? 0x3F : 0);
                mask[elem->node_id(s)] |= m;                         // both nodes of the flagged side
                mask[elem->node_id((s + 1) % elem->n_sides())] |= m; // (fem-shell.cpp:90-120)
            }
        }
    }
    check(femshell_set_mesh(b.ctx, (int32_t)n_nodes, xyz.data(), (int32_t)(tri.size() / 3), tri.data(),
                            (int32_t)(quad.size() / 4), quad.data()));
    check(femshell_set_dirichlet(b.ctx, (int32_t)n_nodes, nullptr, mask.data()));
    b.mesh_sent = true;
}

inline void send_forces(const MeshBase &mesh)
{
    std::vector<double> f(6 * mesh.n_nodes(), 0.0);
    for (dof_id_type n = 0; n < forces.size() && n < mesh.n_nodes(); n++)
        for (int i = 0; i < 6; i++) f[6 * n + i] = forces[n](i);
    check(femshell_set_loads(binding().ctx, (int32_t)mesh.n_nodes(), nullptr, f.data()));
}

// (1) drop-in for assemble_elasticity: system.attach_assemble_function(femshell_assemble_elasticity)
inline void femshell_assemble_elasticity(EquationSystems &es, const std::string &system_name)
{
    libmesh_assert_equal_to(system_name, "Elasticity");
    LinearImplicitSystem &system = es.get_system<LinearImplicitSystem>("Elasticity");
    Binding &b = binding();
    b.es = &es;
    if (!b.mesh_sent) send_mesh(es);
    send_forces(es.get_mesh());
    check(femshell_assemble(b.ctx));
    if (!b.compat_copy_back) return; // K and F stay in HBM for FemShellLinearSolver
    // compat mode: hand the assembled block rows to libMesh (ADD semantics, fem-shell.cpp:1230-1231).  Every rank
    // exports the node rows it assembled (femshell_owned_nodes: [row_begin, row_end) unless the library renumbered the
    // nodes; global column ids); libMesh/PETSc route the entries to the owner of each dof when the matrix is closed, as
    // they do for the reference's own add_matrix calls.
    const int64_t nb = femshell_nnz_blocks(b.ctx);
    const int32_t n_rows = femshell_owned_nodes(b.ctx, nullptr);
    std::vector<int32_t> own((size_t)n_rows);
    femshell_owned_nodes(b.ctx, own.data());
    std::vector<int32_t> rowptr(n_rows + 1), colidx(nb);
    std::vector<double> vals(36 * nb), F(6 * (size_t)n_rows);
    check(femshell_export_bsr(b.ctx, rowptr.data(), colidx.data(), vals.data(), F.data()));
    auto dof = [&](int32_t node, unsigned var) { return es.get_mesh().node_ref(node).dof_number(system.number(), var, 0); };
    DenseMatrix<Number> blk(6, 6);
    std::vector<dof_id_type> rows(6), cols(6);
    for (int32_t a = 0; a < n_rows; a++) {
        for (unsigned v = 0; v < 6; v++) rows[v] = dof(own[(size_t)a], v);
        for (int32_t q = rowptr[a]; q < rowptr[a + 1]; q++) {
            for (unsigned v = 0; v < 6; v++) cols[v] = dof(colidx[q], v);
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) blk(i, j) = vals[36 * (size_t)q + 6 * i + j];
            system.matrix->add_matrix(blk, rows, cols);
        }
        for (unsigned v = 0; v < 6; v++) system.rhs->add(rows[v], F[6 * (size_t)a + v]);
    }
}

// (2) drop-in for the PETSc KSP solve: system.linear_solver.reset(new FemShellLinearSolver(comm))
class FemShellLinearSolver : public LinearSolver<Number> {
  public:
    explicit FemShellLinearSolver(const Parallel::Communicator &comm) : LinearSolver<Number>(comm) {}
    void clear() override {}
    void init(const char * = nullptr) override { this->_is_initialized = true; }
    std::pair<unsigned int, Real> solve(SparseMatrix<Number> &, NumericVector<Number> &solution, NumericVector<Number> &,
                                        const double tol, const unsigned int m_its) override
    {
        femshell_solve_info info{};
        Binding &b = binding();
        if (!b.es) libmesh_error_msg("FemShellLinearSolver: attach femshell_assemble_elasticity first");
        const MeshBase &mesh = b.es->get_mesh();
        const unsigned sys = b.es->get_system<LinearImplicitSystem>("Elasticity").number();
        std::vector<double> u(6 * (size_t)mesh.n_nodes());
        // `solution` arrives holding the initial guess, as it does for PETSc (PetscLinearSolver: KSPSetInitialGuessNonzero): zero on
        // the first solve of a system, the last solution afterwards -- every rank gets the whole vector, as femshell_set_initial_guess
        // takes it; an all-zero guess is the solve from zero
        {
            std::vector<Number> all;
            solution.localize(all);
            bool any = false;
            for (const Node *nd : mesh.node_ptr_range())
                for (unsigned var = 0; var < 6; var++) {
                    const double v = all[nd->dof_number(sys, var, 0)];
                    u[6 * (size_t)nd->id() + var] = v;
                    any = any || v != 0.0;
                }
            if (any) check(femshell_set_initial_guess(b.ctx, u.data()));
        }
        b.last_rc = femshell_solve(b.ctx, tol, (int32_t)m_its, u.data(), &info); // full vector on every rank, 6*node+var
        b.last_info = info;
        // a breakdown (K or the preconditioner not positive definite) is a solver outcome libMesh asks about through
        // get_converged_reason(), like PETSc's KSP_DIVERGED_BREAKDOWN; every other failure is an error
        if (b.last_rc == FEMSHELL_ERR_BREAKDOWN) return {(unsigned)info.iterations, info.rel_residual};
        check(b.last_rc);
        // libMesh numbers dofs per processor and (by default) variable-major: go through dof_number(), like the
        // reference does when it reads the solution back (fem-shell.cpp:163-169); each rank sets the dofs it owns
        for (const Node *nd : mesh.local_node_ptr_range())
            for (unsigned var = 0; var < 6; var++) solution.set(nd->dof_number(sys, var, 0), u[6 * (size_t)nd->id() + var]);
        solution.close();
        return {(unsigned)info.iterations, info.rel_residual};
    }
    std::pair<unsigned int, Real> solve(SparseMatrix<Number> &A, SparseMatrix<Number> &, NumericVector<Number> &x,
                                        NumericVector<Number> &b, const double tol, const unsigned int its) override
    {
        return solve(A, x, b, tol, its);
    }
    std::pair<unsigned int, Real> solve(const ShellMatrix<Number> &, NumericVector<Number> &, NumericVector<Number> &,
                                        const double, const unsigned int) override
    {
        libmesh_not_implemented();
    }
    std::pair<unsigned int, Real> solve(const ShellMatrix<Number> &, const SparseMatrix<Number> &, NumericVector<Number> &,
                                        NumericVector<Number> &, const double, const unsigned int) override
    {
        libmesh_not_implemented();
    }
    void print_converged_reason() const override
    {
        const Binding &b = binding();
        const LinearConvergenceReason r = get_converged_reason();
        std::printf("FemShellLinearSolver: %s after %d iterations, ||r||/||b|| = %.3e (%s, estimated relative error %.1e)\n",
                    r == CONVERGED_RTOL_NORMAL ? "converged" : r == DIVERGED_ITS ? "iteration limit reached" : "breakdown",
                    (int)b.last_info.iterations, b.last_info.rel_residual,
                    b.last_info.pc_type == FEMSHELL_PC_AMG ? "multigrid" : "block-Jacobi", b.last_info.error_estimate);
    }
    LinearConvergenceReason get_converged_reason() const override
    {
        const Binding &b = binding();
        if (b.last_rc == FEMSHELL_ERR_BREAKDOWN) return DIVERGED_BREAKDOWN;
        return b.last_info.converged == 1 ? CONVERGED_RTOL_NORMAL : DIVERGED_ITS;
    }
};

} // namespace femshell_libmesh

#endif // FEMSHELL_HAVE_LIBMESH
