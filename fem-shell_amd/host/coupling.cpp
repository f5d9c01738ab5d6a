// coupling.cpp -- see coupling.hpp
#include "coupling.hpp"

#include <cstdio>

#ifdef FEMSHELL_HAVE_PRECICE
// build with -DFEMSHELL_HAVE_PRECICE -lprecice: the coupled program then talks to the real coupling library (pre-1.0
// SolverInterface API, the one the reference uses, PC:15, 50-52); -inprocess keeps the built-in stand-in
#include "precice/SolverInterface.hpp"
#endif

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <regex>
#include <sstream>
#include <stdexcept>

namespace femshell_host {

// ---- dummy fluid ------------------------------------------------------------------------------

DummyFluid DummyFluid::tower(int dimensions)
{
    DummyFluid d;
    d.dimensions = dimensions;
    const int N = 43; // fluid_solver.cpp:45-47: the count is hard-wired, the N argument is ignored
    d.grid.assign((size_t)N * dimensions, 0.0);
    d.f.assign((size_t)N * dimensions, 0.0);
    for (int k = 0; k < 21; k++) { // left edge of the tower
        d.grid[(size_t)k * dimensions] = 3.0;
        d.grid[(size_t)k * dimensions + 1] = k * 0.1;
    }
    for (int k = 21; k < 42; k++) { // right edge
        d.grid[(size_t)k * dimensions] = 3.25;
        d.grid[(size_t)k * dimensions + 1] = (k - 21.0) * 0.1;
    }
    d.grid[(size_t)42 * dimensions] = 3.125; // top
    d.grid[(size_t)42 * dimensions + 1] = 2.0;
    for (int i = 0; i < 21; i++) d.forced.push_back(i); // fluid_solver.cpp:190-195
    return d;
}

DummyFluid DummyFluid::left_edge(int dimensions, const std::vector<double> &pos)
{
    DummyFluid d;
    d.dimensions = dimensions;
    d.grid = pos;
    d.f.assign(pos.size(), 0.0);
    const int n = (int)(pos.size() / dimensions);
    double xmin = 1e300;
    for (int i = 0; i < n; i++) xmin = std::min(xmin, pos[(size_t)i * dimensions]);
    for (int i = 0; i < n; i++)
        if (pos[(size_t)i * dimensions] <= xmin + 1e-12 * (1.0 + std::fabs(xmin))) d.forced.push_back(i);
    return d;
}

void DummyFluid::compute_forces()
{
    for (int i : forced) {
        f[(size_t)i * dimensions] = 1.0 + std::sin(t / 25.01);
        if (dimensions == 3) f[(size_t)i * dimensions + 1] = 0.0;
    }
}

// ---- in-process coupling ------------------------------------------------------------------------

void InProcessCoupling::setMeshVertices(int, int n, const double *positions, int *ids)
{
    dim_ = fluid_.dimensions;
    spos_.assign(positions, positions + (size_t)n * dim_);
    for (int i = 0; i < n; i++) ids[i] = i;
    // consistent nearest-neighbour mapping Fluid_Nodes -> Structure_Nodes (precice_config.xml:45)
    nearest_.assign((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        double best = 1e300;
        for (int k = 0; k < fluid_.n(); k++) {
            double d2 = 0.0;
            for (int d = 0; d < dim_; d++) {
                const double dx = spos_[(size_t)i * dim_ + d] - fluid_.grid[(size_t)k * dim_ + d];
                d2 += dx * dx;
            }
            if (d2 < best) {
                best = d2;
                nearest_[(size_t)i] = k;
            }
        }
    }
    forces_.assign((size_t)n * dim_, 0.0);
    displ_.assign((size_t)n * dim_, 0.0);
    displ_prev_ = displ_;
    displ_base_ = displ_;
}

void InProcessCoupling::map_forces_to_structure()
{
    for (size_t i = 0; i < nearest_.size(); i++)
        for (int d = 0; d < dim_; d++) forces_[i * dim_ + d] = fluid_.f[(size_t)nearest_[i] * dim_ + d];
}

double InProcessCoupling::initialize()
{
    // serial-implicit, FLUID first (precice_config.xml:58): the fluid has computed its first forces
    // before the structure's first solve
    fluid_.compute_forces();
    map_forces_to_structure();
    iter_ = 0;
    need_write_cp_ = true;
    return scheme_.timestep;
}

bool InProcessCoupling::isActionRequired(const std::string &action) const
{
    if (action == actionWriteInitialData()) return need_init_data_;
    if (action == actionWriteIterationCheckpoint()) return need_write_cp_;
    if (action == actionReadIterationCheckpoint()) return need_read_cp_;
    return false;
}

void InProcessCoupling::fulfilledAction(const std::string &action)
{
    if (action == actionWriteInitialData()) need_init_data_ = false;
    if (action == actionWriteIterationCheckpoint()) need_write_cp_ = false;
    if (action == actionReadIterationCheckpoint()) need_read_cp_ = false;
}

void InProcessCoupling::initializeData() { need_init_data_ = false; }

void InProcessCoupling::writeBlockVectorData(int, int n, const int *ids, const double *values)
{
    for (int i = 0; i < n; i++)
        for (int d = 0; d < dim_; d++) displ_[(size_t)ids[i] * dim_ + d] = values[(size_t)i * dim_ + d];
}

void InProcessCoupling::readBlockVectorData(int, int n, const int *ids, double *values) const
{
    for (int i = 0; i < n; i++)
        for (int d = 0; d < dim_; d++) values[(size_t)i * dim_ + d] = forces_[(size_t)ids[i] * dim_ + d];
}

double InProcessCoupling::advance(double dt)
{
    // relative convergence measure on the displacement data (precice_config.xml:67)
    iter_++;
    iterations_total_++;
    double diff = 0.0, norm = 0.0;
    for (size_t i = 0; i < displ_.size(); i++) {
        diff += (displ_[i] - displ_prev_[i]) * (displ_[i] - displ_prev_[i]);
        norm += displ_[i] * displ_[i];
    }
    const bool converged = (iter_ > 1 && std::sqrt(diff) <= scheme_.rel_limit * std::sqrt(norm)) ||
                           (iter_ > 1 && norm == 0.0 && diff == 0.0) || iter_ >= scheme_.max_iterations;
    displ_prev_ = displ_;
    if (!converged) {
        need_read_cp_ = true; // the structure prints "Iterate" and repeats the step (PC:325-330)
        // the fluid repeats its step with the new displacements: its forces do not depend on them
        fluid_.compute_forces();
    } else {
        need_read_cp_ = false;
        need_write_cp_ = true;
        time_ += dt;
        steps_++;
        iter_ = 0;
        fluid_.t++; // fluid_solver.cpp:226
        fluid_.compute_forces();
    }
    map_forces_to_structure();
    return scheme_.timestep;
}

// ---- structure-side adapter state -----------------------------------------------------------------

void CoupledStructure::init(const ShellMesh &m, int dims, char dead_axis)
{
    dimensions = dims;
    deadAxis = dead_axis;
    if (dimensions == 2 && deadAxis != 'x' && deadAxis != 'y' && deadAxis != 'z')
        throw std::runtime_error("Error: preCICE expects 2D mesh, but mesh file does not provide this requirement. "
                                 "Allowed values: 'x', 'y' or 'z'"); // PC:92-99
    interface_nodes = m.nodes_with_ids({2, 20, 21});
    const int n = (int)interface_nodes.size();
    grid.assign((size_t)n * dimensions, 0.0);
    forces.assign((size_t)n * dimensions, 0.0);
    displ.assign((size_t)n * dimensions, 0.0);
    preSols.assign((size_t)m.n_nodes() * 6, 0.0);
    const std::array<int, 2> ax = dead_axis_components(deadAxis);
    for (int i = 0; i < n; i++) {
        const int32_t id = interface_nodes[(size_t)i];
        id_map[id] = i;
        if (dimensions == 3) {
            for (int d = 0; d < 3; d++) grid[(size_t)i * 3 + d] = m.xyz[3 * (size_t)id + d];
        } else {
            grid[(size_t)i * 2] = m.xyz[3 * (size_t)id + ax[0]];
            grid[(size_t)i * 2 + 1] = m.xyz[3 * (size_t)id + ax[1]];
        }
    }
}

std::vector<double> CoupledStructure::loads_from_forces(int32_t n_nodes) const
{
    std::vector<double> f6((size_t)n_nodes * 6, 0.0);
    const std::array<int, 2> ax = dead_axis_components(deadAxis);
    for (const auto &kv : id_map) {
        const size_t node = (size_t)kv.first, v = (size_t)kv.second;
        if (dimensions == 3) {
            for (int i = 0; i < 3; i++) f6[6 * node + i] = forces[v * 3 + i];
        } else {
            f6[6 * node + ax[0]] = forces[v * 2];
            f6[6 * node + ax[1]] = forces[v * 2 + 1];
        }
    }
    return f6;
}

void CoupledStructure::displacement_increments(const std::vector<double> &sols)
{
    const std::array<int, 2> ax = dead_axis_components(deadAxis);
    for (size_t i = 0; i < interface_nodes.size(); i++) {
        const size_t id = (size_t)interface_nodes[i];
        if (dimensions == 3) {
            for (int d = 0; d < 3; d++) displ[i * 3 + d] = sols[6 * id + d] - preSols[6 * id + d];
        } else {
            displ[i * 2] = sols[6 * id + ax[0]] - preSols[6 * id + ax[0]];
            displ[i * 2 + 1] = sols[6 * id + ax[1]] - preSols[6 * id + ax[1]];
        }
    }
}

void CoupledStructure::accept_time_step(const std::vector<double> &sols)
{
    const std::array<int, 2> ax = dead_axis_components(deadAxis);
    for (int32_t idn : interface_nodes) {
        const size_t id = (size_t)idn;
        if (dimensions == 3) {
            for (int j = 0; j < 3; j++) preSols[6 * id + j] = sols[6 * id + j];
        } else {
            preSols[6 * id + ax[0]] = sols[6 * id + ax[0]];
            preSols[6 * id + ax[1]] = sols[6 * id + ax[1]];
        }
    }
}

// ---- the coupled program ------------------------------------------------------------------------------

namespace {

const char *arg_after(int argc, char **argv, const char *flag)
{
    for (int i = 1; i + 1 < argc; i++)
        if (std::strcmp(argv[i], flag) == 0) return argv[i + 1];
    return nullptr;
}

// the four scheme values of the reference's precice_config.xml, if the file is readable
InProcessCoupling::Scheme scheme_from_xml(const std::string &path, int *dimensions)
{
    InProcessCoupling::Scheme s;
    std::ifstream in(path);
    if (!in) return s;
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string x = ss.str();
    std::smatch m;
    if (std::regex_search(x, m, std::regex("<max-time\\s+value=\"([^\"]+)\""))) s.max_time = std::atof(m[1].str().c_str());
    if (std::regex_search(x, m, std::regex("<timestep-length\\s+value=\"([^\"]+)\""))) s.timestep = std::atof(m[1].str().c_str());
    if (std::regex_search(x, m, std::regex("<max-iterations\\s+value=\"([^\"]+)\""))) s.max_iterations = std::atoi(m[1].str().c_str());
    if (std::regex_search(x, m, std::regex("<relative-convergence-measure\\s+data=\"Displacements\"[^>]*limit=\"([^\"]+)\"")))
        s.rel_limit = std::atof(m[1].str().c_str());
    if (dimensions && std::regex_search(x, m, std::regex("<solver-interface\\s+dimensions=\"([^\"]+)\"")))
        *dimensions = std::atoi(m[1].str().c_str());
    return s;
}

} // namespace

int fem_shell_precice_main(int argc, char **argv, std::ostream &out, std::ostream &err)
{
    PhaseClock clock;
    out << "Starting Structure Solver..." << std::endl;
    if (argc < 7) {
        err << "Error, must choose valid parameters.\n"
            << "Usage: " << argv[0] << " -nu -e -t -mesh -config -dt [-axis] [-out] [-d]\n"
            << "-config:\t (preCICE) configuration file (required; without preCICE its coupling-scheme values drive\n"
            << "\t\t the in-process stand-in, the dummy fluid of fluid_solver.cpp is built in)\n"
            << "-dt:\t (preCICE) max time step length (required, recommended to set same as in config XML)\n"
            << "-axis:\t (preCICE) dead axis ([x,y,z] optional)\n"
            << "-steps:\t stop after this many time steps (optional)\n"
            << "-fluid:\t dummy fluid geometry: tower (fluid_solver.cpp, default) or edge (forces on the interface's\n"
            << "\t\t minimal-first-coordinate edge, for other meshes)\n";
        out << "Read command-line arguments.......FAILED" << std::endl;
        return -1;
    }
    Parameters p;
    if (!read_parameters(argc, argv, p, out, err)) {
        out << "Read command-line arguments.......FAILED" << std::endl;
        return -1;
    }
    const char *config = arg_after(argc, argv, "-config");
    const char *dtv = arg_after(argc, argv, "-dt");
    if (!config) err << "ERROR: preCICE configuration file not specified!\n";
    if (!dtv) err << "ERROR: preCICE max time step length not specified!\n";
    if (!config || !dtv) {
        out << "Read command-line arguments.......FAILED" << std::endl;
        return -1;
    }
    const char *axis = arg_after(argc, argv, "-axis");
    const char deadAxis = axis ? axis[0] : '0';
    const double deltaT = std::atof(dtv);
    const char *stepsv = arg_after(argc, argv, "-steps");
    out << "Read command-line arguments.......OK" << std::endl;
    try {
        ShellMesh mesh = read_mesh(p.in_filename);
        clock.done("read mesh");
        int dims = 2;
        const InProcessCoupling::Scheme scheme = scheme_from_xml(config, &dims);
        const char *fluid = arg_after(argc, argv, "-fluid"); // extension: "tower" (default) | "edge"
        DummyFluid dummy = DummyFluid::tower(dims);
        if (fluid && std::string(fluid) == "edge") {
            CoupledStructure probe_cs;
            probe_cs.init(mesh, dims, deadAxis);
            dummy = DummyFluid::left_edge(dims, probe_cs.grid);
        }
        InProcessCoupling interface(dummy, scheme);
#ifdef FEMSHELL_HAVE_PRECICE
        bool in_process = false; // built against preCICE: the real library unless -inprocess is given
        for (int i = 1; i < argc; i++) in_process = in_process || std::string(argv[i]) == "-inprocess";
#else
        const bool in_process = true;
#endif
        if (in_process)
            out << "preCICE configured... (in-process stand-in, " << dims << "D, " << scheme.max_time / scheme.timestep
                << " time steps)" << std::endl;
        // probe: the highest interface node, displacement along the first live axis
        const std::vector<int32_t> ifn = mesh.nodes_with_ids({2, 20, 21});
        if (ifn.empty()) throw std::runtime_error("mesh has no coupling interface (boundary ids 2, 20, 21)");
        out << "preCICE dimensions = " << dims << ", dead axis = " << deadAxis << ", coupling interface nodes = " << ifn.size()
            << std::endl; // PC:74-77
        // one process per GPU: the in-process coupling stand-in and its dummy fluid run replicated on every rank (they are
        // deterministic and see the full solution vector, PC:274-280), the structure solve is row-partitioned
        const Launch launch = Launch::from_environment();
        ShellSystem system(p, launch);
        clock.done("coupling set-up, context (device, ranks)");
        mesh.loads.assign((size_t)mesh.n_nodes() * 6, 0.0);
        system.set_mesh(mesh);
        clock.done("symbolic phase, boundary conditions");
        const std::array<int, 2> ax = dead_axis_components(deadAxis == '0' ? 'z' : deadAxis);
        int32_t probe = ifn[0];
        for (int32_t n : ifn)
            if (mesh.xyz[3 * (size_t)n + ax[1]] > mesh.xyz[3 * (size_t)probe + ax[1]]) probe = n;
        std::ostream quiet(nullptr); // ranks other than 0 compute the same coupling steps silently
        CoupledRunLog log;
        // per converged time step, named as the reference names them (PC:1526-1560): <out>_NNNN.e when several processes run,
        // <out>_NNN.pvtu (+ its piece) in a serial run; rank 0 writes (every rank holds the gathered solution)
        std::function<void(int, const std::vector<double> &)> write_step;
        if (p.isOutfileSet && launch.rank == 0)
            write_step = [&](int t, const std::vector<double> &sols) {
                char tag[16];
                if (launch.world_size > 1) {
                    snprintf(tag, sizeof tag, "_%04d", t);
                    write_exodus(mesh, sols, p.out_filename + tag + ".e");
                } else {
                    snprintf(tag, sizeof tag, "_%03d", t);
                    write_pvtu(mesh, sols, p.out_filename + tag + ".pvtu");
                }
            };
#ifdef FEMSHELL_HAVE_PRECICE
        if (!in_process) {
            // the reference's participant: a real coupling library and a real fluid solver on the other side (PC:50-52);
            // every rank joins with its rank / size like global_processor_id() / global_n_processors() there
            precice::SolverInterface real("STRUCTURE", config, launch.rank, launch.world_size);
            out << "preCICE configured..." << std::endl;
            log = run_coupled_structure(real, system, mesh, deadAxis, deltaT, p.tol, p.max_it, probe, ax[0],
                                        stepsv ? std::atoi(stepsv) : -1, launch.rank == 0 ? out : quiet, p.debug, write_step);
        } else
#endif
            log = run_coupled_structure(interface, system, mesh, deadAxis, deltaT, p.tol, p.max_it, probe, ax[0],
                                        stepsv ? std::atoi(stepsv) : -1, launch.rank == 0 ? out : quiet, p.debug, write_step);
        if (launch.rank != 0) return 0;
        clock.done("coupling loop (assembly, preconditioner setup, solves, per-step output)");
        {
            char note[120];
            snprintf(note, sizeof note, "of which assembly %.4f, solves %.3f", log.assemble_seconds, log.solve_seconds);
            clock.note(note);
        }
        out << "Linear solver: " << (log.pc_type == FEMSHELL_PC_AMG ? "multigrid-preconditioned" : "6x6 block-Jacobi") << " CG on MI355X";
        if (launch.world_size > 1) out << " (" << launch.world_size << " ranks)";
        out << std::endl;
        out << "Coupled run: " << log.time_steps << " time steps, " << log.coupling_iterations << " coupling iterations, "
            << log.cg_iterations << " CG iterations, " << log.assemblies << " assemblies of K, assembly " << log.assemble_seconds
            << " s, solves " << log.solve_seconds << " s" << std::endl;
        for (size_t i = 0; i < log.tip_displacement.size(); i++)
            out << "tip[" << i << "] node " << probe << " = " << log.tip_displacement[i] << "\n";
        if (p.isOutfileSet) {
            const std::vector<double> sols = system.build_solution_vector();
            write_exodus(mesh, sols, p.out_filename + ".e");
            write_vtk(mesh, sols, p.out_filename + ".vtk");
            clock.done("final output files");
        }
        out << "All done :)\n";
        clock.report(err);
        return 0;
    } catch (const std::exception &e) {
        err << "ERROR: " << e.what() << std::endl;
        return 1;
    }
}

} // namespace femshell_host
