// mesh_io.cpp -- see mesh_io.hpp
#include "mesh_io.hpp"

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <set>
#include <sstream>
#include <stdexcept>

namespace femshell_host {

namespace {

std::string strip_comment(const std::string &line)
{
    const size_t h = line.find('#');
    return h == std::string::npos ? line : line.substr(0, h);
}

std::vector<std::string> read_lines(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::vector<std::string> out;
    std::string l;
    while (std::getline(in, l)) out.push_back(strip_comment(l));
    return out;
}

} // namespace

std::vector<int32_t> ShellMesh::element_nodes(int32_t e) const
{
    char kind;
    int32_t idx;
    if (order.empty()) {
        kind = e < n_tri() ? 't' : 'q';
        idx = e < n_tri() ? e : e - n_tri();
    } else {
        kind = order.at((size_t)e).first;
        idx = order[(size_t)e].second;
    }
    if (kind == 't') return {tri[3 * idx], tri[3 * idx + 1], tri[3 * idx + 2]};
    return {quad[4 * idx], quad[4 * idx + 1], quad[4 * idx + 2], quad[4 * idx + 3]};
}

std::vector<uint8_t> ShellMesh::dirichlet_mask() const
{
    std::vector<uint8_t> mask((size_t)n_nodes(), 0);
    for (const SideBC &b : bcs) {
        const std::vector<int32_t> nd = element_nodes(b.elem);
        const int32_t a = nd[(size_t)b.side], c = nd[(size_t)(b.side + 1) % nd.size()];
        uint8_t m = 0;
        if (b.id == 0 || b.id == 20) m = 0x07;       // u, v, w
        else if (b.id == 1 || b.id == 21) m = 0x3F;  // all six
        mask[(size_t)a] |= m;
        mask[(size_t)c] |= m;
    }
    return mask;
}

std::vector<int32_t> ShellMesh::nodes_with_ids(const std::vector<int32_t> &ids) const
{
    std::set<int32_t> s;
    for (const SideBC &b : bcs)
        if (std::find(ids.begin(), ids.end(), b.id) != ids.end()) {
            const std::vector<int32_t> nd = element_nodes(b.elem);
            s.insert(nd[(size_t)b.side]);
            s.insert(nd[(size_t)(b.side + 1) % nd.size()]);
        }
    return std::vector<int32_t>(s.begin(), s.end());
}

ShellMesh read_xda(const std::string &path)
{
    const std::vector<std::string> L = read_lines(path);
    if (L.size() < 8 || L[0].rfind("libMesh", 0) != 0) throw std::runtime_error(path + ": not an ASCII XDA file");
    ShellMesh m;
    const long n_elem = std::stol(L[1]), n_nodes = std::stol(L[2]);
    size_t pos = 8;
    if (L.size() < pos + (size_t)n_elem + (size_t)n_nodes + 1) throw std::runtime_error(path + ": truncated XDA file");
    for (long e = 0; e < n_elem; e++, pos++) {
        std::istringstream is(L[pos]);
        int type;
        is >> type;
        if (type == 3) {
            int32_t a, b, c;
            is >> a >> b >> c;
            m.order.push_back({'t', m.n_tri()});
            m.tri.insert(m.tri.end(), {a, b, c});
        } else if (type == 5) {
            int32_t a, b, c, d;
            is >> a >> b >> c >> d;
            m.order.push_back({'q', m.n_quad()});
            m.quad.insert(m.quad.end(), {a, b, c, d});
        } else {
            throw std::runtime_error(path + ": unsupported element type " + std::to_string(type) +
                                     " (only TRI3 = 3 and QUAD4 = 5)");
        }
        if (!is) throw std::runtime_error(path + ": bad element line " + std::to_string(e));
    }
    for (long n = 0; n < n_nodes; n++, pos++) {
        std::istringstream is(L[pos]);
        double x, y, z;
        is >> x >> y >> z;
        if (!is) throw std::runtime_error(path + ": bad node line " + std::to_string(n));
        m.xyz.insert(m.xyz.end(), {x, y, z});
    }
    const long n_bc = std::stol(L[pos++]);
    for (long b = 0; b < n_bc && pos < L.size(); b++, pos++) {
        std::istringstream is(L[pos]);
        SideBC bc;
        is >> bc.elem >> bc.side >> bc.id;
        if (!is) throw std::runtime_error(path + ": bad boundary line " + std::to_string(b));
        if (bc.elem < 0 || bc.elem >= n_elem) throw std::runtime_error(path + ": boundary element out of range");
        m.bcs.push_back(bc);
    }
    m.loads.assign((size_t)n_nodes * 6, 0.0);
    return m;
}

std::string force_file_name(const std::string &mesh_path)
{
    std::string s = mesh_path;
    for (const char *ext : {".xda", ".xdr", ".msh"})
        if (s.find(ext) != std::string::npos) {
            s.resize(s.size() - 4);
            break;
        }
    return s + "_f";
}

std::vector<double> read_forces(const std::string &path, int32_t n_nodes)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    long n = 0;
    double factor = 1.0;
    in >> n >> factor;
    if (!in) throw std::runtime_error(path + ": bad force file header");
    std::vector<double> out((size_t)n_nodes * 6, 0.0);
    for (long i = 0; i < n && i < n_nodes; i++)
        for (int j = 0; j < 6; j++) {
            double v = 0.0;
            if (in >> v) out[(size_t)i * 6 + j] = v * factor;
        }
    return out;
}

void write_xda(const ShellMesh &m, const std::string &path)
{
    std::ofstream os(path);
    if (!os) throw std::runtime_error("cannot write " + path);
    os.precision(17);
    const long n_elem = m.n_tri() + m.n_quad();
    os << "libMesh-0.7.0+\n";
    os << n_elem << "      # number of elements\n";
    os << m.n_nodes() << "      # number of nodes\n";
    os << ".        # boundary condition specification file\n";
    os << "n/a      # subdomain id specification file\n";
    os << "n/a      # processor id specification file\n";
    os << "n/a      # p-level specification file\n";
    os << n_elem << "      # n_elem at level 0, [ type (n0 ... nN-1) ]\n";
    for (long e = 0; e < n_elem; e++) {
        const std::vector<int32_t> nd = m.element_nodes((int32_t)e);
        os << (nd.size() == 3 ? '3' : '5');
        for (int32_t v : nd) os << " " << v;
        os << "\n";
    }
    for (int32_t n = 0; n < m.n_nodes(); n++) os << m.xyz[3 * n] << " " << m.xyz[3 * n + 1] << " " << m.xyz[3 * n + 2] << "\n";
    os << m.bcs.size() << "        # number of boundary conditions\n";
    for (const SideBC &b : m.bcs) os << b.elem << " " << b.side << " " << b.id << "\n";
}

ShellMesh generate_structured(const MeshGenArgs &a)
{
    if (a.nx <= 0 || a.ny <= 0) throw std::runtime_error("meshgen: nx and ny must be positive");
    if (a.type != 't' && a.type != 'q') throw std::runtime_error("meshgen: type must be t or q");
    if (a.dead_axis != 'x' && a.dead_axis != 'y' && a.dead_axis != 'z') throw std::runtime_error("meshgen: dead axis must be x, y or z");
    ShellMesh m;
    const int nx = a.nx, ny = a.ny;
    const double fx = (a.max_x - a.min_x) / nx, fy = (a.max_y - a.min_y) / ny;
    for (int y = 0; y <= ny; y++)
        for (int x = 0; x <= nx; x++) {
            double p[3] = {0, 0, 0};
            const double prim = a.min_x + x * fx, sec = a.min_y + y * fy;
            if (a.dead_axis == 'z') { p[0] = prim; p[1] = sec; }
            else if (a.dead_axis == 'y') { p[0] = prim; p[2] = sec; }
            else { p[1] = prim; p[2] = sec; }
            m.xyz.insert(m.xyz.end(), p, p + 3);
        }
    const int up = nx + 1;
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) {
            const int n = x + y * up;
            if (a.type == 'q') {
                m.order.push_back({'q', m.n_quad()});
                m.quad.insert(m.quad.end(), {n, n + 1, n + up + 1, n + up});
            } else if (a.ul_lr) {
                m.order.push_back({'t', m.n_tri()});
                m.tri.insert(m.tri.end(), {n, n + 1, n + up});
                m.order.push_back({'t', m.n_tri()});
                m.tri.insert(m.tri.end(), {n + 1, n + up + 1, n + up});
            } else {
                m.order.push_back({'t', m.n_tri()});
                m.tri.insert(m.tri.end(), {n, n + up + 1, n + 1});
                m.order.push_back({'t', m.n_tri()});
                m.tri.insert(m.tri.end(), {n + up + 1, n, n + up});
            }
        }
    // side boundary ids, in the reference tool's order: top/bottom first, then left/right
    for (int i = 0; i < nx; i++) {
        if (a.type == 't') {
            if (a.bc_bottom >= 0) m.bcs.push_back({2 * i, a.ul_lr ? 0 : 2, a.bc_bottom});
            if (a.bc_top >= 0) m.bcs.push_back({2 * nx * ny - 2 * i - 1, a.ul_lr ? 1 : 2, a.bc_top});
        } else {
            if (a.bc_bottom >= 0) m.bcs.push_back({i, 0, a.bc_bottom});
            if (a.bc_top >= 0) m.bcs.push_back({nx * ny - 1 - i, 2, a.bc_top});
        }
    }
    for (int i = 0; i < ny; i++) {
        if (a.type == 't') {
            if (a.ul_lr) {
                if (a.bc_left >= 0) m.bcs.push_back({2 * nx * i, 2, a.bc_left});
                if (a.bc_right >= 0) m.bcs.push_back({2 * nx * (i + 1) - 1, 0, a.bc_right});
            } else {
                if (a.bc_left >= 0) m.bcs.push_back({2 * nx * i + 1, 1, a.bc_left});
                if (a.bc_right >= 0) m.bcs.push_back({2 * nx * (i + 1) - 2, 1, a.bc_right});
            }
        } else {
            if (a.bc_left >= 0) m.bcs.push_back({nx * i, 3, a.bc_left});
            if (a.bc_right >= 0) m.bcs.push_back({nx * (i + 1) - 1, 1, a.bc_right});
        }
    }
    const int32_t nn = m.n_nodes();
    m.loads.assign((size_t)nn * 6, 0.0);
    const int axis = a.dead_axis == 'x' ? 0 : (a.dead_axis == 'y' ? 1 : 2);
    const int32_t rows = a.meshgen_quirk ? nn - 1 : nn;
    if (a.loading == 1) {
        if (nn / 2 < rows) m.loads[(size_t)(nn / 2) * 6 + axis] = a.factor;
    } else if (a.loading == 2) {
        for (int32_t n = 0; n < rows; n++) m.loads[(size_t)n * 6 + axis] = a.factor * fx * fy;
    }
    return m;
}

void write_meshgen_files(const MeshGenArgs &a, const std::string &name)
{
    const ShellMesh m = generate_structured(a);
    write_xda(m, name + ".xda");
    if (a.loading <= 0) return;
    std::ofstream os(name + "_f");
    if (!os) throw std::runtime_error("cannot write " + name + "_f");
    const int32_t nn = m.n_nodes();
    const double fx = (a.max_x - a.min_x) / a.nx, fy = (a.max_y - a.min_y) / a.ny;
    const char *unit = a.dead_axis == 'x' ? "1 0 0 0 0 0\n" : (a.dead_axis == 'y' ? "0 1 0 0 0 0\n" : "0 0 1 0 0 0\n");
    os << nn << "\n";
    const int32_t rows = a.meshgen_quirk ? nn - 1 : nn;
    if (a.loading == 1) {
        os << a.factor << "\n";
        for (int32_t i = 0; i < rows; i++) os << (i == nn / 2 ? unit : "0 0 0 0 0 0\n");
    } else {
        os << a.factor * fx * fy << "\n";
        for (int32_t i = 0; i < rows; i++) os << unit;
    }
}

void write_vtk(const ShellMesh &m, const std::vector<double> &u, const std::string &path)
{
    std::ofstream os(path);
    if (!os) throw std::runtime_error("cannot write " + path);
    os.precision(12);
    const int32_t nn = m.n_nodes();
    os << "# vtk DataFile Version 3.0\nfem-shell displaced mesh\nASCII\nDATASET UNSTRUCTURED_GRID\n";
    os << "POINTS " << nn << " double\n";
    for (int32_t n = 0; n < nn; n++)
        os << m.xyz[3 * n] + u[6 * (size_t)n] << " " << m.xyz[3 * n + 1] + u[6 * (size_t)n + 1] << " "
           << m.xyz[3 * n + 2] + u[6 * (size_t)n + 2] << "\n";
    const long ne = m.n_tri() + m.n_quad();
    os << "CELLS " << ne << " " << 4L * m.n_tri() + 5L * m.n_quad() << "\n";
    for (long e = 0; e < ne; e++) {
        const std::vector<int32_t> nd = m.element_nodes((int32_t)e);
        os << nd.size();
        for (int32_t v : nd) os << " " << v;
        os << "\n";
    }
    os << "CELL_TYPES " << ne << "\n";
    for (long e = 0; e < ne; e++) os << (m.element_nodes((int32_t)e).size() == 3 ? 5 : 9) << "\n";
    os << "POINT_DATA " << nn << "\n";
    static const char *names[6] = {"u", "v", "w", "tx", "ty", "tz"};
    for (int v = 0; v < 6; v++) {
        os << "SCALARS " << names[v] << " double 1\nLOOKUP_TABLE default\n";
        for (int32_t n = 0; n < nn; n++) os << u[6 * (size_t)n + v] << "\n";
    }
}

} // namespace femshell_host
