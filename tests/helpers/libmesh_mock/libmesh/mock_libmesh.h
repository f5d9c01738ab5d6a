// mock_libmesh.h -- TEST-ONLY stand-in for the handful of libMesh declarations fem-shell_amd/host/libmesh_adaptor.hpp
// touches, so that the adaptor can be syntax-checked (g++ -fsyntax-only) in an image without libMesh.
// This is NOT libMesh and NOT a reference build: nothing here is linked, run, or used to produce a number; the
// signatures follow libMesh's public headers as documented (mesh_base.h, elem.h, node.h, dof_object.h,
// boundary_info.h, numeric_vector.h, sparse_matrix.h, linear_solver.h, parallel communicator).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#define libmesh_error_msg(msg) do { (void)(msg); std::abort(); } while (0)
#define libmesh_assert_equal_to(a, b) ((void)((a) == (b)))
#define libmesh_not_implemented() std::abort()

namespace libMesh {
typedef double Real;
typedef double Number;
typedef uint32_t dof_id_type;
typedef uint32_t numeric_index_type;
typedef int16_t boundary_id_type;
enum ElemType { TRI3 = 3, QUAD4 = 5 };
enum LinearConvergenceReason { CONVERGED_RTOL_NORMAL = 1, DIVERGED_ITS = -3, DIVERGED_BREAKDOWN = -5 }; // values of libMesh's enum_convergence_flags.h (= PETSc's KSPConvergedReason)
template <class T> struct DenseVector { T operator()(unsigned) const; };
template <class T> struct DenseMatrix { DenseMatrix(unsigned, unsigned); T &operator()(unsigned, unsigned); };
template <class T> struct Range { T *begin() const; T *end() const; };
struct Node { dof_id_type id() const; Real operator()(unsigned) const; dof_id_type dof_number(unsigned sys, unsigned var, unsigned comp) const; };
struct Elem { ElemType type() const; unsigned n_nodes() const; unsigned n_sides() const; dof_id_type node_id(unsigned) const; };
struct BoundaryInfo { void boundary_ids(const Elem *, unsigned short side, std::vector<boundary_id_type> &ids) const; };
namespace Parallel { struct Communicator { template <class T> void broadcast(std::vector<T> &, unsigned root = 0) const; }; }
struct MeshBase {
    dof_id_type n_nodes() const; unsigned processor_id() const; unsigned n_processors() const;
    Range<const Node *const> node_ptr_range() const; Range<const Node *const> local_node_ptr_range() const;
    Range<const Elem *const> active_element_ptr_range() const;
    const BoundaryInfo &get_boundary_info() const; const Node &node_ref(dof_id_type) const;
    const Parallel::Communicator &comm() const;
};
template <class T> struct NumericVector {
    numeric_index_type size() const, first_local_index() const, last_local_index() const;
    void set(numeric_index_type, T); void add(numeric_index_type, T); void close();
    void localize(std::vector<T> &v_local) const; // (libMesh: the whole vector on every processor)
};
template <class T> struct SparseMatrix { void add_matrix(const DenseMatrix<T> &, const std::vector<dof_id_type> &rows, const std::vector<dof_id_type> &cols); };
template <class T> struct ShellMatrix {};
template <class T> struct LinearSolver {
    explicit LinearSolver(const Parallel::Communicator &);
    virtual ~LinearSolver();
    virtual void clear() = 0;
    virtual void init(const char *name = nullptr) = 0;
    virtual std::pair<unsigned int, Real> solve(SparseMatrix<T> &, NumericVector<T> &, NumericVector<T> &, const double, const unsigned int) = 0;
    virtual std::pair<unsigned int, Real> solve(SparseMatrix<T> &, SparseMatrix<T> &, NumericVector<T> &, NumericVector<T> &, const double, const unsigned int) = 0;
    virtual std::pair<unsigned int, Real> solve(const ShellMatrix<T> &, NumericVector<T> &, NumericVector<T> &, const double, const unsigned int) = 0;
    virtual std::pair<unsigned int, Real> solve(const ShellMatrix<T> &, const SparseMatrix<T> &, NumericVector<T> &, NumericVector<T> &, const double, const unsigned int) = 0;
    virtual void print_converged_reason() const = 0;
    virtual LinearConvergenceReason get_converged_reason() const = 0;
    bool _is_initialized;
};
struct LinearImplicitSystem {
    unsigned number() const;
    SparseMatrix<Number> *matrix; NumericVector<Number> *rhs; std::unique_ptr<LinearSolver<Number>> linear_solver;
};
struct EquationSystems { const MeshBase &get_mesh() const; template <class S> S &get_system(const std::string &); };
} // namespace libMesh
