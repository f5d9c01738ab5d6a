// mesh_io.cpp -- see mesh_io.hpp
#include "mesh_io.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
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
    for (const auto &nb : node_bcs) {
        if (nb.second == 0 || nb.second == 20) mask[(size_t)nb.first] |= 0x07;
        else if (nb.second == 1 || nb.second == 21) mask[(size_t)nb.first] |= 0x3F;
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
    for (const auto &nb : node_bcs)
        if (std::find(ids.begin(), ids.end(), nb.second) != ids.end()) s.insert(nb.first);
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
        if (bc.side < 0 || bc.side >= (int32_t)m.element_nodes(bc.elem).size())
            throw std::runtime_error(path + ": boundary line " + std::to_string(b) + " names side " + std::to_string(bc.side) +
                                     " of an element with " + std::to_string(m.element_nodes(bc.elem).size()) + " sides");
        m.bcs.push_back(bc);
    }
    if (L[0].rfind("libMesh-0.9.2+", 0) == 0 && pos < L.size() && !L[pos].empty() && L[pos].find_first_not_of(" \t\r") != std::string::npos) {
        const long n_ns = std::stol(L[pos++]); // nodesets: (node, boundary id)
        for (long b = 0; b < n_ns && pos < L.size(); b++, pos++) {
            std::istringstream is(L[pos]);
            int32_t node, id;
            is >> node >> id;
            if (!is || node < 0 || node >= n_nodes) throw std::runtime_error(path + ": bad nodeset line " + std::to_string(b));
            m.node_bcs.push_back({node, id});
        }
    }
    m.loads.assign((size_t)n_nodes * 6, 0.0);
    return m;
}

ShellMesh read_msh(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    ShellMesh m;
    std::vector<std::pair<long, int32_t>> id_map; // Gmsh node number -> index in file order
    auto node_index = [&](long id) -> int32_t {
        auto it = std::lower_bound(id_map.begin(), id_map.end(), std::make_pair(id, (int32_t)-1));
        if (it == id_map.end() || it->first != id) throw std::runtime_error(path + ": element references unknown node " + std::to_string(id));
        return it->second;
    };
    struct Low { int nn; int32_t n[2]; int32_t id; };
    std::vector<Low> lows;
    std::string tok;
    bool have_nodes = false, have_elems = false;
    while (in >> tok) {
        if (tok == "$MeshFormat") {
            double ver = 0; int type = 0, size = 0;
            in >> ver >> type >> size;
            if (!in || ver < 2.0 || ver >= 3.0 || type != 0) throw std::runtime_error(path + ": only Gmsh ASCII format 2.x is supported");
        } else if (tok == "$Nodes") {
            long n = 0;
            in >> n;
            for (long i = 0; i < n; i++) {
                long id; double x, y, z;
                in >> id >> x >> y >> z;
                if (!in) throw std::runtime_error(path + ": bad node line " + std::to_string(i));
                id_map.push_back({id, (int32_t)i});
                m.xyz.insert(m.xyz.end(), {x, y, z});
            }
            std::sort(id_map.begin(), id_map.end());
            have_nodes = true;
        } else if (tok == "$Elements") {
            if (!have_nodes) throw std::runtime_error(path + ": $Elements before $Nodes");
            long n = 0;
            in >> n;
            for (long i = 0; i < n; i++) {
                long idx; int type, ntags;
                in >> idx >> type >> ntags;
                if (!in || ntags < 2) throw std::runtime_error(path + ": element line " + std::to_string(i) + " needs at least two tags (libMesh's requirement)");
                long phys = 0;
                for (int t = 0; t < ntags; t++) { long v; in >> v; if (t == 0) phys = v; }
                const int nn = type == 2 ? 3 : type == 3 ? 4 : type == 1 ? 2 : type == 15 ? 1 : -1;
                if (nn < 0) throw std::runtime_error(path + ": unsupported Gmsh element type " + std::to_string(type) + " (points, lines, triangles and quadrangles only)");
                long ids[4];
                for (int k = 0; k < nn; k++) in >> ids[k];
                if (!in) throw std::runtime_error(path + ": bad element line " + std::to_string(i));
                if (nn == 3) {
                    m.order.push_back({'t', m.n_tri()});
                    for (int k = 0; k < 3; k++) m.tri.push_back(node_index(ids[k]));
                } else if (nn == 4) {
                    m.order.push_back({'q', m.n_quad()});
                    for (int k = 0; k < 4; k++) m.quad.push_back(node_index(ids[k]));
                } else {
                    Low l{nn, {node_index(ids[0]), nn == 2 ? node_index(ids[1]) : -1}, (int32_t)phys};
                    lows.push_back(l);
                }
            }
            have_elems = true;
        }
    }
    if (!have_nodes || !have_elems || m.order.empty()) throw std::runtime_error(path + ": no $Nodes / $Elements with triangles or quadrangles");
    // lower-dimensional elements -> boundary conditions.  Sides are looked up through one table of (min node, max node)
    // -> (element, side), first element in file order wins: a scan over all elements per boundary line is quadratic
    // (1e10 steps on a 4M-triangle mesh with its ~6000 boundary lines); libMesh's GmshIO goes through a node-to-element map too
    const int32_t n_elem = (int32_t)m.order.size();
    struct SideRef { int64_t key; int32_t elem, side; };
    std::vector<SideRef> sides;
    if (std::any_of(lows.begin(), lows.end(), [](const Low &l) { return l.nn == 2; })) {
        sides.reserve((size_t)m.n_tri() * 3 + (size_t)m.n_quad() * 4);
        for (int32_t e = 0; e < n_elem; e++) {
            const bool t = m.order[(size_t)e].first == 't';
            const int32_t *nd = t ? &m.tri[3 * (size_t)m.order[(size_t)e].second] : &m.quad[4 * (size_t)m.order[(size_t)e].second];
            const int nn = t ? 3 : 4;
            for (int sd = 0; sd < nn; sd++) {
                const int64_t a = nd[sd], b = nd[(sd + 1) % nn];
                sides.push_back({(std::min(a, b) << 32) | std::max(a, b), e, sd});
            }
        }
        std::sort(sides.begin(), sides.end(), [](const SideRef &x, const SideRef &y) {
            return x.key != y.key ? x.key < y.key : (x.elem != y.elem ? x.elem < y.elem : x.side < y.side);
        });
    }
    for (const Low &l : lows) {
        if (l.nn == 1) {
            m.node_bcs.push_back({l.n[0], l.id});
            continue;
        }
        const int64_t a = l.n[0], b = l.n[1], key = (std::min(a, b) << 32) | std::max(a, b);
        auto it = std::lower_bound(sides.begin(), sides.end(), key, [](const SideRef &x, int64_t k) { return x.key < k; });
        if (it == sides.end() || it->key != key) throw std::runtime_error(path + ": a boundary line is not a side of any element");
        m.bcs.push_back({it->elem, it->side, l.id});
    }
    m.loads.assign((size_t)m.n_nodes() * 6, 0.0);
    return m;
}

ShellMesh read_mesh(const std::string &path)
{
    auto ends_with = [&](const char *ext) { const std::string e(ext); return path.size() >= e.size() && path.compare(path.size() - e.size(), e.size(), e) == 0; };
    if (ends_with(".msh")) return read_msh(path);
    if (ends_with(".xdr")) return read_xdr(path);
    return read_xda(path);
}

// ---- binary XDR: the records of the XDA file in Sun XDR encoding (RFC 4506) -----------------------------------------
// libMesh's XdrIO writes the same sequence of fields to both formats (mesh.read() picks the codec by extension,
// fem-shell.cpp:35-37); in the binary one every integer is a big-endian 32-bit word, every coordinate a big-endian
// IEEE double, a string its length as a 32-bit word followed by the bytes padded with zeros to a multiple of four,
// and the comments of the ASCII file ("# number of elements") do not exist.  No libMesh exists in this image to
// produce such a file, so the reader is tested against write_xdr below (round trip with the XDA reader), not against a
// file libMesh wrote: format parity with libMesh's own writer is unpinned.
namespace {

struct XdrIn {
    std::ifstream in;
    std::string path;
    explicit XdrIn(const std::string &p) : in(p, std::ios::binary), path(p)
    {
        if (!in) throw std::runtime_error("cannot open " + p);
    }
    void bytes(void *dst, size_t n)
    {
        in.read(static_cast<char *>(dst), (std::streamsize)n);
        if ((size_t)in.gcount() != n) throw std::runtime_error(path + ": truncated XDR file");
    }
    uint32_t u32()
    {
        unsigned char b[4];
        bytes(b, 4);
        return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3];
    }
    double f64()
    {
        unsigned char b[8];
        bytes(b, 8);
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v = (v << 8) | b[i];
        double d;
        std::memcpy(&d, &v, 8);
        return d;
    }
    std::string str()
    {
        const uint32_t n = u32();
        if (n > 4096) throw std::runtime_error(path + ": implausible string length in XDR header (not an XDR mesh file?)");
        std::string s((size_t)((n + 3u) & ~3u), '\0');
        if (!s.empty()) bytes(&s[0], s.size());
        s.resize(n);
        return s;
    }
};

struct XdrOut {
    std::ofstream os;
    explicit XdrOut(const std::string &p) : os(p, std::ios::binary)
    {
        if (!os) throw std::runtime_error("cannot write " + p);
    }
    void u32(uint32_t v)
    {
        const unsigned char b[4] = {(unsigned char)(v >> 24), (unsigned char)(v >> 16), (unsigned char)(v >> 8), (unsigned char)v};
        os.write(reinterpret_cast<const char *>(b), 4);
    }
    void f64(double d)
    {
        uint64_t v;
        std::memcpy(&v, &d, 8);
        unsigned char b[8];
        for (int i = 7; i >= 0; i--, v >>= 8) b[i] = (unsigned char)v;
        os.write(reinterpret_cast<const char *>(b), 8);
    }
    void str(const std::string &s)
    {
        u32((uint32_t)s.size());
        os.write(s.data(), (std::streamsize)s.size());
        static const char zeros[4] = {0, 0, 0, 0};
        os.write(zeros, (std::streamsize)(((s.size() + 3) & ~(size_t)3) - s.size()));
    }
};

} // namespace

ShellMesh read_xdr(const std::string &path)
{
    XdrIn x(path);
    const std::string version = x.str();
    if (version.rfind("libMesh-0.7.0+", 0) != 0 && version.rfind("libMesh-0.9.2+", 0) != 0)
        throw std::runtime_error(path + ": XDR header \"" + version + "\" is not one this reader knows (libMesh-0.7.0+, libMesh-0.9.2+: 32-bit fields)");
    const bool nodesets = version.rfind("libMesh-0.9.2+", 0) == 0;
    const uint32_t n_elem = x.u32(), n_nodes = x.u32();
    const std::string bc_file = x.str(), subdomain_file = x.str(), partition_file = x.str(), plevel_file = x.str();
    if (subdomain_file != "n/a" || partition_file != "n/a" || plevel_file != "n/a")
        throw std::runtime_error(path + ": subdomain / processor / p-level records are not supported (the reference's meshes carry none)");
    const uint32_t n_level0 = x.u32();
    if (n_level0 != n_elem) throw std::runtime_error(path + ": refined meshes (elements above level 0) are not supported");
    ShellMesh m;
    for (uint32_t e = 0; e < n_elem; e++) {
        const uint32_t type = x.u32();
        if (type == 3) {
            m.order.push_back({'t', m.n_tri()});
            for (int k = 0; k < 3; k++) m.tri.push_back((int32_t)x.u32());
        } else if (type == 5) {
            m.order.push_back({'q', m.n_quad()});
            for (int k = 0; k < 4; k++) m.quad.push_back((int32_t)x.u32());
        } else {
            throw std::runtime_error(path + ": unsupported element type " + std::to_string(type) + " (only TRI3 = 3 and QUAD4 = 5)");
        }
    }
    for (int32_t v : m.tri)
        if (v < 0 || (uint32_t)v >= n_nodes) throw std::runtime_error(path + ": element references a node out of range");
    for (int32_t v : m.quad)
        if (v < 0 || (uint32_t)v >= n_nodes) throw std::runtime_error(path + ": element references a node out of range");
    m.xyz.resize((size_t)n_nodes * 3);
    for (double &c : m.xyz) c = x.f64();
    if (bc_file == ".") { // the boundary conditions follow in this file
        const uint32_t n_bc = x.u32();
        for (uint32_t b = 0; b < n_bc; b++) {
            SideBC bc;
            bc.elem = (int32_t)x.u32();
            bc.side = (int32_t)x.u32();
            bc.id = (int32_t)x.u32();
            if (bc.elem < 0 || (uint32_t)bc.elem >= n_elem) throw std::runtime_error(path + ": boundary element out of range");
            if (bc.side < 0 || bc.side >= (int32_t)m.element_nodes(bc.elem).size())
                throw std::runtime_error(path + ": boundary record " + std::to_string(b) + " names a side the element does not have");
            m.bcs.push_back(bc);
        }
        if (nodesets) {
            const uint32_t n_ns = x.u32();
            for (uint32_t b = 0; b < n_ns; b++) {
                const int32_t node = (int32_t)x.u32(), id = (int32_t)x.u32();
                if (node < 0 || (uint32_t)node >= n_nodes) throw std::runtime_error(path + ": nodeset node out of range");
                m.node_bcs.push_back({node, id});
            }
        }
    }
    m.loads.assign((size_t)n_nodes * 6, 0.0);
    return m;
}

void write_xdr(const ShellMesh &m, const std::string &path)
{
    XdrOut o(path);
    const uint32_t n_elem = (uint32_t)(m.n_tri() + m.n_quad());
    o.str(m.node_bcs.empty() ? "libMesh-0.7.0+" : "libMesh-0.9.2+");
    o.u32(n_elem);
    o.u32((uint32_t)m.n_nodes());
    o.str(".");
    o.str("n/a");
    o.str("n/a");
    o.str("n/a");
    o.u32(n_elem);
    for (uint32_t e = 0; e < n_elem; e++) {
        const std::vector<int32_t> nd = m.element_nodes((int32_t)e);
        o.u32(nd.size() == 3 ? 3u : 5u);
        for (int32_t v : nd) o.u32((uint32_t)v);
    }
    for (double c : m.xyz) o.f64(c);
    o.u32((uint32_t)m.bcs.size());
    for (const SideBC &b : m.bcs) {
        o.u32((uint32_t)b.elem);
        o.u32((uint32_t)b.side);
        o.u32((uint32_t)b.id);
    }
    if (!m.node_bcs.empty()) {
        o.u32((uint32_t)m.node_bcs.size());
        for (const auto &nb : m.node_bcs) {
            o.u32((uint32_t)nb.first);
            o.u32((uint32_t)nb.second);
        }
    }
    if (!o.os) throw std::runtime_error("cannot write " + path);
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

void write_xda(const ShellMesh &m, const std::string &path, int precision)
{
    std::ofstream os(path);
    if (!os) throw std::runtime_error("cannot write " + path);
    os.precision(precision);
    const long n_elem = m.n_tri() + m.n_quad();
    // (node boundary ids -- Gmsh point elements -- need the nodeset record libMesh added with the 0.9.2+ header; files
    //  without them keep the header and bytes of the reference's generator)
    os << (m.node_bcs.empty() ? "libMesh-0.7.0+\n" : "libMesh-0.9.2+\n");
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
    if (!m.node_bcs.empty()) {
        os << m.node_bcs.size() << "        # number of nodesets\n";
        for (const auto &nb : m.node_bcs) os << nb.first << " " << nb.second << "\n";
    }
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
    write_xda(m, name + ".xda", a.precision);
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
