// shell_system.cpp -- see shell_system.hpp
#include "shell_system.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <thread>

#include <fcntl.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

namespace femshell_host {

namespace {

void check(int rc, const char *what)
{
    if (rc != FEMSHELL_OK) throw std::runtime_error(std::string(what) + ": " + femshell_last_error());
}

// GetPot-like lookup: value following `flag`, if present
const char *arg_after(int argc, char **argv, const char *flag)
{
    for (int i = 1; i + 1 < argc; i++)
        if (std::strcmp(argv[i], flag) == 0) return argv[i + 1];
    return nullptr;
}

bool has_flag(int argc, char **argv, const char *flag)
{
    for (int i = 1; i < argc; i++)
        if (std::strcmp(argv[i], flag) == 0) return true;
    return false;
}

} // namespace

bool read_parameters(int argc, char **argv, Parameters &p, std::ostream &out, std::ostream &err)
{
    if (argc < 5) {
        err << "Error, must choose valid parameters.\n"
            << "Usage: " << argv[0] << " -nu -e -t -mesh [-out] [-d]\n"
            << "-nu:\t Possion's ratio (required)\n"
            << "-e:\t Elastic/Young's modulus E (required)\n"
            << "-t:\t Thickness (required)\n"
            << "-mesh:\t Input mesh file (*.xda, required)\n"
            << "-out:\t Output file name (without extension, optional)\n"
            << "-d:\t Additional (debug) messages (1=on, 0=off (default))\n"
            << "-tol:\t relative residual tolerance of the CG solve (optional, default 1e-12)\n"
            << "-max_it:\t iteration limit of the CG solve (optional, default 5000)\n"
            << "-pc_type:\t gamg (multigrid, default) | bjacobi (6x6 block-Jacobi)\n";
        return false;
    }
    bool failed = false;
    if (const char *v = arg_after(argc, argv, "-d")) p.debug = std::atoi(v) == 1;
    if (has_flag(argc, argv, "-nu")) {
        const char *v = arg_after(argc, argv, "-nu");
        p.nu = v ? std::atof(v) : 0.3;
    } else {
        err << "ERROR: Poisson's ratio nu not specified!\n";
        failed = true;
    }
    if (has_flag(argc, argv, "-e")) {
        const char *v = arg_after(argc, argv, "-e");
        p.em = v ? std::atof(v) : 1.0e6;
    } else {
        err << "ERROR: Elastic modulus E not specified!\n";
        failed = true;
    }
    if (has_flag(argc, argv, "-t")) {
        const char *v = arg_after(argc, argv, "-t");
        p.thickness = v ? std::atof(v) : 1.0;
    } else {
        err << "ERROR: Mesh thickness t not specified!\n";
        failed = true;
    }
    if (has_flag(argc, argv, "-mesh")) {
        const char *v = arg_after(argc, argv, "-mesh");
        p.in_filename = v ? v : "mesh.xda";
    } else {
        err << "ERROR: Mesh file not specified!\n";
        failed = true;
    }
    if (has_flag(argc, argv, "-out")) {
        const char *v = arg_after(argc, argv, "-out");
        p.out_filename = v ? v : "out";
        p.isOutfileSet = true;
    } else {
        p.isOutfileSet = false;
    }
    if (const char *v = arg_after(argc, argv, "-tol")) p.tol = std::atof(v);
    if (const char *v = arg_after(argc, argv, "-max_it")) p.max_it = std::atoi(v);
    // PETSc-style options the reference's users pass on the same command line (doc/implementation.tex:68-72)
    if (const char *v = arg_after(argc, argv, "-ksp_rtol")) p.tol = std::atof(v);
    if (const char *v = arg_after(argc, argv, "-ksp_max_it")) p.max_it = std::atoi(v);
    if (const char *v = arg_after(argc, argv, "-ksp_type")) {
        p.ksp_type = v;
        if (p.ksp_type != "cg") {
            err << "NOTE: -ksp_type " << p.ksp_type << " is not available: K is symmetric positive definite and the solve on the GPU"
                << " is a conjugate-gradient method; using -ksp_type cg\n";
            p.ksp_type = "cg";
        }
    }
    if (const char *v = arg_after(argc, argv, "-pc_type")) {
        p.pc_type = v;
        static const char *known[] = {"jacobi", "bjacobi", "pbjacobi", "gamg", "amg", "ml", "hypre", "mg"};
        bool ok = false;
        for (const char *k : known) ok = ok || p.pc_type == k;
        if (p.pc_type == "none" || p.pc_type == "ilu" || p.pc_type == "icc" || p.pc_type == "sor" || p.pc_type == "asm" || p.pc_type == "lu") {
            // PETSc's own defaults (ilu serial, bjacobi + ilu parallel) and other host-side preconditioners: handled like an
            // unavailable -ksp_type -- say what runs instead, do not fail
            err << "NOTE: -pc_type " << p.pc_type << " is not available on the GPU; using the 6x6 block-Jacobi preconditioner"
                << " (-pc_type gamg selects the multigrid)\n";
            p.pc_type = "pbjacobi";
        } else if (!ok) {
            err << "ERROR: -pc_type " << p.pc_type << " is not available (jacobi|bjacobi|pbjacobi -> 6x6 block-Jacobi, gamg|amg|ml|hypre|mg -> multigrid)\n";
            failed = true;
        }
    }

    out << "Run program with parameters:"
        << " debug messages = " << (p.debug ? "true" : "false") << ", nu = " << p.nu << ", E = " << p.em
        << ", t = " << p.thickness << ", mesh file = " << p.in_filename;
    if (p.isOutfileSet) out << ", out-file = " << p.out_filename;
    out << std::endl;
    return !failed;
}

ShellSystem::ShellSystem(const Parameters &p, int device, int rank, int world_size, unsigned flags)
{
    femshell_config cfg{};
    cfg.nu = p.nu;
    cfg.E = p.em;
    cfg.thickness = p.thickness;
    cfg.flags = flags;
    cfg.device = device;
    cfg.rank = rank;
    cfg.world_size = world_size;
    rank_ = rank;
    check(femshell_create(&cfg, &ctx_), "femshell_create");
}

Launch Launch::from_environment()
{
    auto env_int = [](std::initializer_list<const char *> names, int fallback) {
        for (const char *n : names)
            if (const char *v = std::getenv(n)) return std::atoi(v);
        return fallback;
    };
    Launch l;
    l.rank = env_int({"FEMSHELL_RANK", "RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK"}, 0);
    l.world_size = env_int({"FEMSHELL_WORLD_SIZE", "WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE"}, 1);
    l.device = env_int({"FEMSHELL_DEVICE", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK"}, -1);
    if (std::getenv("FEMSHELL_SINGLE") && std::atoi(std::getenv("FEMSHELL_SINGLE")) == 1) { // ignore a launcher's RANK / WORLD_SIZE
        l.rank = 0;
        l.world_size = 1;
    }
    if (const char *f = std::getenv("FEMSHELL_UID_FILE")) l.uid_file = f;
    else {
        // per-user directory, not a name in /tmp anyone can squat: $XDG_RUNTIME_DIR, else /tmp/femshell-<uid> (0700, ours)
        std::string dir;
        if (const char *x = std::getenv("XDG_RUNTIME_DIR")) dir = x;
        if (dir.empty()) {
            dir = "/tmp/femshell-" + std::to_string((long)getuid());
            (void)mkdir(dir.c_str(), 0700);
            struct stat st;
            if (lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 077) != 0)
                throw std::runtime_error("launch environment: " + dir + " is not a private directory of this user (set FEMSHELL_UID_FILE)");
        }
        const char *port = std::getenv("MASTER_PORT");
        l.uid_file = dir + "/femshell_uid_" + (port ? std::string(port) : std::to_string((long)getppid()));
    }
    if (l.world_size < 1 || l.rank < 0 || l.rank >= l.world_size) throw std::runtime_error("launch environment: invalid rank / world size");
    if (l.world_size > 1) // a launcher's generic RANK / WORLD_SIZE turn a plain run into rank k of N: say so
        std::cerr << "fem-shell: rank " << l.rank << " of " << l.world_size << " (from the launch environment; RCCL id through " << l.uid_file
                  << "; FEMSHELL_SINGLE=1 runs single-process regardless)" << std::endl;
    return l;
}

namespace {

// The RCCL id travels through a file: 8 bytes of magic, then the 128-byte id.  Rank 0 removes whatever an earlier run
// left under the name, writes a temporary created with O_EXCL | O_NOFOLLOW and renames it into place; the others accept
// only a regular file of this user that is not older than a minute before their own start (a crashed run's leftover).
constexpr char kUidMagic[8] = {'F', 'S', 'H', 'L', 'U', 'I', 'D', '1'};

void publish_uid(const std::string &path, const unsigned char id[128])
{
    (void)unlink(path.c_str());
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) throw std::runtime_error("cannot create " + tmp);
    const bool ok = write(fd, kUidMagic, 8) == 8 && write(fd, id, 128) == 128;
    (void)close(fd);
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) {
        (void)unlink(tmp.c_str());
        throw std::runtime_error("cannot publish " + path);
    }
}

bool read_uid(const std::string &path, time_t not_before, unsigned char id[128])
{
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    char magic[8];
    const bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == getuid() && st.st_mtime >= not_before &&
                    read(fd, magic, 8) == 8 && std::memcmp(magic, kUidMagic, 8) == 0 && read(fd, id, 128) == 128;
    (void)close(fd);
    return ok;
}

} // namespace

ShellSystem::ShellSystem(const Parameters &p, const Launch &launch, unsigned flags)
    : ShellSystem(p, launch.device, launch.rank, launch.world_size, flags)
{
    if (launch.world_size > 1) {
        unsigned char id[128];
        const time_t started = time(nullptr);
        if (launch.rank == 0) {
            check(femshell_comm_unique_id(id), "femshell_comm_unique_id");
            publish_uid(launch.uid_file, id);
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            while (!read_uid(launch.uid_file, started - 60, id)) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                    throw std::runtime_error("rank " + std::to_string(launch.rank) + ": no fresh RCCL id in " + launch.uid_file + " after 120 s");
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        comm_init(id);
        if (launch.rank == 0 && !std::getenv("FEMSHELL_KEEP_UID_FILE")) std::remove(launch.uid_file.c_str()); // every rank has joined
    }
    choose_preconditioner(p);
}

void ShellSystem::choose_preconditioner(const Parameters &p)
{
    const bool amg = p.pc_type == "gamg" || p.pc_type == "amg" || p.pc_type == "ml" || p.pc_type == "hypre" || p.pc_type == "mg";
    femshell_pc_options o;
    check(femshell_pc_defaults(amg ? FEMSHELL_PC_AMG : FEMSHELL_PC_BLOCK_JACOBI, &o), "femshell_pc_defaults");
    check(femshell_set_preconditioner(ctx_, &o), "femshell_set_preconditioner");
}

ShellSystem::~ShellSystem() { femshell_destroy(ctx_); }

void ShellSystem::comm_init(const unsigned char id[128]) { check(femshell_comm_init(ctx_, id), "femshell_comm_init"); }

void ShellSystem::set_mesh(const ShellMesh &m)
{
    n_nodes_ = m.n_nodes();
    solved_once_ = false;
    check(femshell_set_mesh(ctx_, m.n_nodes(), m.xyz.data(), m.n_tri(), m.tri.data(), m.n_quad(), m.quad.data()),
          "femshell_set_mesh");
    const std::vector<uint8_t> mask = m.dirichlet_mask();
    check(femshell_set_dirichlet(ctx_, m.n_nodes(), nullptr, mask.data()), "femshell_set_dirichlet");
    if (!m.loads.empty()) set_forces(m.loads);
}

void ShellSystem::set_forces(const std::vector<double> &f6)
{
    if ((int)(f6.size() / 6) != n_nodes_) throw std::runtime_error("set_forces: need one row of 6 per mesh node");
    check(femshell_set_loads(ctx_, n_nodes_, nullptr, f6.data()), "femshell_set_loads");
}

void ShellSystem::assemble_elasticity(const std::string &system_name)
{
    // libmesh_assert_equal_to (system_name, "Elasticity"), SA:1163
    if (system_name != "Elasticity") throw std::runtime_error("assemble_elasticity: system_name must be \"Elasticity\"");
    check(femshell_assemble(ctx_), "femshell_assemble");
}

SolveResult ShellSystem::solve(double tol, int max_it)
{
    SolveResult r;
    sols_.assign((size_t)n_nodes_ * 6, 0.0);
    // libMesh hands system.solution to KSPSolve as the initial guess (PetscLinearSolver: KSPSetInitialGuessNonzero): the first
    // solve of a system starts from zero, every later one -- the coupling iterations of fem-shell_precice.cpp:271 -- from the
    // displacements of the solve before.  FEMSHELL_WARM_START=0: every solve from zero (A/B runs)
    static const bool warm = !(std::getenv("FEMSHELL_WARM_START") && std::atoi(std::getenv("FEMSHELL_WARM_START")) == 0);
    if (warm && solved_once_) check(femshell_set_initial_guess(ctx_, nullptr), "femshell_set_initial_guess");
    check(femshell_solve(ctx_, tol, max_it, sols_.data(), &r.info), "femshell_solve");
    solved_once_ = true;
    r.iterations = (unsigned)r.info.iterations;
    r.final_residual = r.info.rel_residual;
    r.converged = r.info.converged == 1;
    return r;
}

const std::vector<double> &ShellSystem::build_solution_vector() { return sols_; }

namespace {
double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
} // namespace

PhaseClock::PhaseClock()
{
    const char *e = std::getenv("FEMSHELL_TIMING");
    enabled_ = e && std::atoi(e) != 0;
    start_ = last_ = wall_now();
}

void PhaseClock::done(const std::string &phase)
{
    const double t = wall_now();
    phases_.emplace_back(phase, t - last_);
    last_ = t;
}

void PhaseClock::report(std::ostream &err) const
{
    if (!enabled_) return;
    char buf[64];
    err << "Times [s]:";
    for (const auto &p : phases_) {
        snprintf(buf, sizeof buf, " %.3f", p.second);
        err << " " << p.first << buf << " |";
    }
    snprintf(buf, sizeof buf, " %.3f", last_ - start_);
    err << " total" << buf;
    for (const std::string &n : notes_) err << " | " << n;
    err << std::endl;
}

int fem_shell_main(int argc, char **argv, std::ostream &out, std::ostream &err)
{
    PhaseClock clock;
    Parameters p;
    if (read_parameters(argc, argv, p, out, err)) {
        out << "Read command-line arguments.......OK" << std::endl;
    } else {
        out << "Read command-line arguments.......FAILED" << std::endl;
        return -1;
    }
    try {
        ShellMesh mesh = read_mesh(p.in_filename);
        out << " Mesh Information:\n  n_nodes()=" << mesh.n_nodes() << "\n  n_elem()=" << mesh.n_tri() + mesh.n_quad()
            << "\n";
        // CONVENTION: force file = mesh file name without extension + "_f" (SA:42-50); a missing
        // file means no loads, as in the reference (SA:52)
        try {
            mesh.loads = read_forces(force_file_name(p.in_filename), mesh.n_nodes());
        } catch (const std::exception &) {
            mesh.loads.assign((size_t)mesh.n_nodes() * 6, 0.0);
        }
        clock.done("read mesh and loads");
        const Launch launch = Launch::from_environment();
        ShellSystem system(p, launch);
        clock.done("context (device, ranks)");
        system.set_mesh(mesh);
        clock.done("symbolic phase, boundary conditions, loads");
        const SolveResult res = system.solve(p.tol, p.max_it);
        clock.done("assembly, preconditioner setup, solve");
        {
            char note[160];
            snprintf(note, sizeof note, "of which assembly %.4f, preconditioner setup %.3f, iterations %.3f", res.info.assemble_seconds,
                     res.info.pc_setup_seconds, res.info.solve_seconds);
            clock.note(note);
        }
        const std::vector<double> &sols = system.build_solution_vector();
        if (launch.rank != 0) return res.converged ? 0 : 2; // every rank holds the solution (SA:141); rank 0 reports it
        out << "Linear solver: " << (res.info.pc_type == FEMSHELL_PC_AMG ? "multigrid-preconditioned" : "6x6 block-Jacobi")
            << " CG on MI355X";
        if (launch.world_size > 1) out << " (" << launch.world_size << " ranks)";
        out << ", " << res.iterations << " iterations, ||r||/||b|| = "
            << res.final_residual << (res.converged ? "" : " (NOT converged)") << std::endl;
        out << "Solution: u_vec = [";
        for (int32_t id = 0; id < mesh.n_nodes(); id++) {
            const double *s = &sols[6 * (size_t)id];
            out << "u= " << s[0] << ", v= " << s[1] << ", w= " << s[2];
            out << ", tx= " << s[3] << ", ty= " << s[4] << ", tz= " << s[5] << "]" << std::endl;
        }
        out << "]" << std::endl << std::endl;
        clock.done("print the solution");
        if (p.isOutfileSet) {
            write_exodus(mesh, sols, p.out_filename + ".e"); // the reference's file (fem-shell.cpp:1249)
            write_vtk(mesh, sols, p.out_filename + ".vtk");
            clock.done("output files");
        }
        out << "All done :)\n";
        clock.report(err);
        return res.converged ? 0 : 2;
    } catch (const std::exception &e) {
        err << "ERROR: " << e.what() << std::endl;
        return 1;
    }
}

} // namespace femshell_host
