// coupling.hpp -- twin of the reference's preCICE adapter (fem-shell_precice.cpp, "PC") above the C ABI.
//
// run_coupled_structure<Interface>() keeps the reference's call sequence on the coupling library
// (PC:47-170 initialisation, PC:256-412 loop) and is a template on the interface type, so it
// compiles against precice::SolverInterface (pre-1.0 API, the one the reference uses) where preCICE
// exists, and against InProcessCoupling below where it does not (this image).
//
//   reference                                              here
//   interface-node discovery by ids 2/20/21   PC:60-72     ShellMesh::nodes_with_ids({2,20,21})
//   2-D <-> 3-D dead-axis mapping of positions PC:113-135  dead_axis_components()
//   id_map: mesh node -> interface vertex     PC:150-157   CoupledStructure::id_map
//   contribRHS: forces -> nodal loads         PC:1399-1432 CoupledStructure::loads_from_forces
//   incremental displacements sols - preSols  PC:286-320   CoupledStructure::displacement_increments
//   preSols update on a finished time step    PC:343-381   CoupledStructure::accept_time_step
//   re-solve on every coupling iteration      PC:271       ShellSystem::solve (K and the block-Jacobi
//        factors stay in HBM; FEMSHELL_REASSEMBLE_EACH_SOLVE reproduces the reference's re-assembly)
//
// InProcessCoupling stands in for the preCICE library + the reference's dummy fluid participant
// (preCICE/fluid_solver.cpp:84-239): a serial-implicit scheme with the FLUID participant first,
// nearest-neighbour consistent mapping both ways, relative convergence measure on the
// displacements (precice_config.xml:57-78; no IQN-ILS acceleration: the dummy fluid's forces do
// not depend on the displacements, so the fixed point is reached on the second iteration).
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <ostream>
#include <string>
#include <vector>

#include "mesh_io.hpp"
#include "shell_system.hpp"

namespace femshell_host {

// the two mesh axes that are alive when `dead_axis` is dead (PC:113-135, 1399-1432)
inline std::array<int, 2> dead_axis_components(char dead_axis)
{
    if (dead_axis == 'z') return {0, 1};
    if (dead_axis == 'y') return {0, 2};
    return {1, 2};
}

// ---- the dummy fluid of preCICE/fluid_solver.cpp --------------------------------------------
struct DummyFluid {
    int dimensions = 2;
    std::vector<double> grid;  // N x dimensions
    std::vector<double> f;     // N x dimensions
    int t = 0;                 // finished time steps
    // fluid_solver.cpp:95-118: 21 left-edge, 21 right-edge vertices and one on top of the tower
    static DummyFluid tower(int dimensions);
    // the same load pattern on any interface (BASELINE config 5, the perpendicular flap): fluid vertices
    // coincide with the given interface vertices; the forced ones are those on the minimal first coordinate
    static DummyFluid left_edge(int dimensions, const std::vector<double> &interface_positions);
    std::vector<int> forced; // vertices that carry f_x = 1 + sin(t/25.01)
    // fluid_solver.cpp:187-199: f_x = 1 + sin(t/25.01) on the 21 left-edge vertices
    void compute_forces();
    int n() const { return (int)(grid.size() / dimensions); }
};

// ---- in-process stand-in for precice::SolverInterface (STRUCTURE side) ------------------------
class InProcessCoupling {
  public:
    struct Scheme {
        double max_time = 4.0;       // precice_config.xml:60
        double timestep = 0.01;      // :61
        int max_iterations = 40;     // :63
        double rel_limit = 1e-5;     // :67
    };
    InProcessCoupling(DummyFluid fluid, Scheme s) : fluid_(std::move(fluid)), scheme_(s) {}

    // the subset of the SolverInterface API the reference calls, same names and meaning
    int getDimensions() const { return fluid_.dimensions; }
    int getMeshID(const std::string &) const { return 0; }
    int getDataID(const std::string &name, int) const { return name == "Displacements" ? 0 : 1; }
    void setMeshVertices(int, int n, const double *positions, int *ids);
    double initialize();
    bool isActionRequired(const std::string &action) const;
    void fulfilledAction(const std::string &action);
    void initializeData();
    bool isReadDataAvailable() const { return true; }
    void writeBlockVectorData(int, int n, const int *ids, const double *values);
    void readBlockVectorData(int, int n, const int *ids, double *values) const;
    double advance(double dt);
    bool isCouplingOngoing() const { return time_ < scheme_.max_time - 1e-12; }
    void finalize() {}

    int time_steps_done() const { return steps_; }
    int iterations_total() const { return iterations_total_; }

  private:
    void map_forces_to_structure();
    DummyFluid fluid_;
    Scheme scheme_;
    int dim_ = 2;
    std::vector<double> spos_;     // structure vertex positions
    std::vector<int> nearest_;     // structure vertex -> fluid vertex (consistent nearest neighbour)
    std::vector<double> forces_;   // on structure vertices
    std::vector<double> displ_, displ_prev_, displ_base_;
    double time_ = 0.0;
    int steps_ = 0, iter_ = 0, iterations_total_ = 0;
    bool need_write_cp_ = true, need_read_cp_ = false, need_init_data_ = true;
};

inline const std::string &actionWriteInitialData() { static const std::string s = "write-initial-data"; return s; }
inline const std::string &actionWriteIterationCheckpoint() { static const std::string s = "write-iteration-checkpoint"; return s; }
inline const std::string &actionReadIterationCheckpoint() { static const std::string s = "read-iteration-checkpoint"; return s; }

// ---- structure-side state of the adapter ---------------------------------------------------------
struct CoupledStructure {
    int dimensions = 2;
    char deadAxis = 'y';
    std::vector<int32_t> interface_nodes;  // mesh node ids, ascending (PC:60-72)
    std::map<int32_t, int> id_map;         // mesh node -> interface vertex (PC:150-157)
    std::vector<double> grid;              // vertex positions handed to the coupling library
    std::vector<double> forces, displ;     // n x dimensions
    std::vector<double> preSols;           // 6 x n_nodes, solution of the last finished time step

    void init(const ShellMesh &m, int dims, char dead_axis);
    // PC:1399-1432: interface forces -> n_nodes x 6 nodal loads
    std::vector<double> loads_from_forces(int32_t n_nodes) const;
    // PC:286-320
    void displacement_increments(const std::vector<double> &sols);
    // PC:343-381
    void accept_time_step(const std::vector<double> &sols);
};

struct CoupledRunLog {
    int time_steps = 0, coupling_iterations = 0;
    std::vector<double> tip_displacement; // per finished time step: displacement of `probe_node` along `probe_dof`
    double solve_seconds = 0.0, assemble_seconds = 0.0;
    long cg_iterations = 0;
    int pc_type = 0;    // femshell_solve_info::pc_type of the solves
    int assemblies = 0; // solves that assembled K (the reference: every one, PC:271; here: the first, K is constant)
};

// The coupling loop of PC:256-412 on any interface type with the SolverInterface method names.
// (on_time_step(t, sols): called after every converged time step -- the reference's writeOutput(mesh, es, t), PC:393)
template <class Interface>
CoupledRunLog run_coupled_structure(Interface &interface, ShellSystem &system, const ShellMesh &mesh, char deadAxis,
                                    double deltaT, double tol, int max_it, int32_t probe_node, int probe_dof,
                                    int max_time_steps, std::ostream &out, bool debug = false,
                                    const std::function<void(int, const std::vector<double> &)> &on_time_step = nullptr)
{
    CoupledStructure cs;
    cs.init(mesh, interface.getDimensions(), deadAxis);
    const int n_nodes = (int)cs.interface_nodes.size();
    const int meshID = interface.getMeshID("Structure_Nodes");
    const int displID = interface.getDataID("Displacements", meshID);
    const int forceID = interface.getDataID("Stresses", meshID);
    std::vector<int> vertexIDs((size_t)n_nodes);
    interface.setMeshVertices(meshID, n_nodes, cs.grid.data(), vertexIDs.data());
    out << "init preCICE..." << std::endl;
    interface.initialize();
    if (interface.isActionRequired(actionWriteInitialData())) {
        interface.writeBlockVectorData(displID, n_nodes, vertexIDs.data(), cs.displ.data());
        interface.fulfilledAction(actionWriteInitialData());
    }
    interface.initializeData();
    if (interface.isReadDataAvailable()) interface.readBlockVectorData(forceID, n_nodes, vertexIDs.data(), cs.forces.data());

    CoupledRunLog log;
    int t = 0;
    while (interface.isCouplingOngoing() && (max_time_steps < 0 || t < max_time_steps)) {
        if (interface.isActionRequired(actionWriteIterationCheckpoint()))
            interface.fulfilledAction(actionWriteIterationCheckpoint()); // quasi-static: nothing to save (PC:260-265)
        // "the magic": new displacements for the current interface forces (PC:271)
        system.set_forces(cs.loads_from_forces(mesh.n_nodes()));
        const SolveResult res = system.solve(tol, max_it);
        log.solve_seconds += res.info.solve_seconds;
        log.assemble_seconds += res.info.assemble_seconds;
        log.assemblies += res.info.assemble_seconds > 0.0 ? 1 : 0;
        log.pc_type = res.info.pc_type;
        log.cg_iterations += res.iterations;
        const std::vector<double> &sols = system.build_solution_vector();
        cs.displacement_increments(sols);
        if (debug) out << "Displacements sent to preCICE: " << n_nodes << " vertices" << std::endl;
        interface.writeBlockVectorData(displID, n_nodes, vertexIDs.data(), cs.displ.data());
        interface.advance(deltaT);
        interface.readBlockVectorData(forceID, n_nodes, vertexIDs.data(), cs.forces.data());
        log.coupling_iterations++;
        if (interface.isActionRequired(actionReadIterationCheckpoint())) {
            out << "Iterate" << std::endl;
            interface.fulfilledAction(actionReadIterationCheckpoint());
        } else {
            out << "Advancing in time, finished timestep: " << t << std::endl;
            if (on_time_step) on_time_step(t, sols); // write output files (if desired), PC:392-393
            t++;
            cs.accept_time_step(sols);
            log.tip_displacement.push_back(sols[6 * (size_t)probe_node + probe_dof]);
        }
    }
    log.time_steps = t;
    interface.finalize();
    out << "Exiting Structure Solver" << std::endl;
    return log;
}

// The coupled program (PC:18-419): flags of the stand-alone program plus -config -dt [-axis] (PC:428-525).
int fem_shell_precice_main(int argc, char **argv, std::ostream &out, std::ostream &err);

} // namespace femshell_host
