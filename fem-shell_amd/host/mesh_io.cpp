// mesh_io.cpp -- see mesh_io.hpp
#include "mesh_io.hpp"

#include <algorithm>
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>
#include <stdexcept>

namespace femshell_host {

namespace {

// A text file held in memory as a whole, read line by line without copies: the readers of the large formats (ASCII XDA
// and the force file: 30 MB / 60 MB at 1M triangles) parse numbers straight out of the buffer with std::from_chars --
// a stream per line made reading the mesh the longest phase of a run of the coupled program.
struct TextFile {
    std::string buf;
    std::vector<size_t> starts; // offset of every line, plus one past the end

    explicit TextFile(const std::string &path)
    {
        std::ifstream in(path, std::ios::binary);
        if (!in) throw std::runtime_error("cannot open " + path);
        in.seekg(0, std::ios::end);
        const std::streamoff size = in.tellg();
        in.seekg(0, std::ios::beg);
        buf.resize(size > 0 ? (size_t)size : 0);
        if (size > 0) in.read(&buf[0], size);
        if (!in && size > 0) throw std::runtime_error("cannot read " + path);
        const char *b = buf.data(), *e = b + buf.size();
        for (const char *p = b; p < e;) {
            starts.push_back((size_t)(p - b));
            const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
            p = nl ? nl + 1 : e;
        }
        // one past the newline of the last line: behind the buffer when the file does not end with one (as if it did).  A file
        // that DOES end with a newline must not get the extra position: its last line would keep the newline character, and a
        // libMesh >= 0.9.2 file with a blank line behind the boundary-condition section then failed with "bad nodeset count"
        starts.push_back(buf.size() + (buf.empty() || buf.back() != '\n' ? 1 : 0));
    }
    size_t n_lines() const { return starts.size() - 1; }
    // line i without its newline and without a trailing comment
    void line(size_t i, const char **b, const char **e) const
    {
        *b = buf.data() + starts[i];
        *e = buf.data() + std::min(starts[i + 1] - 1, buf.size());
        if (const char *h = (const char *)memchr(*b, '#', (size_t)(*e - *b))) *e = h;
    }
    std::string line_text(size_t i) const
    {
        const char *b, *e;
        line(i, &b, &e);
        return std::string(b, e);
    }
};

// numbers of one line (or of a whole buffer), the way `stream >> value` reads them: blanks skipped, an optional sign
struct NumberCursor {
    const char *p, *e;
    void skip_blanks()
    {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\f' || *p == '\v')) p++;
    }
    template <class T> bool read(T *v)
    {
        skip_blanks();
        if (p < e && *p == '+' && p + 1 < e && p[1] != '-' && p[1] != '+') p++; // (from_chars takes no plus sign)
        const std::from_chars_result r = std::from_chars(p, e, *v);
        if (r.ec != std::errc()) return false;
        p = r.ptr;
        return true;
    }
};

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
    const TextFile F(path);
    if (F.n_lines() < 8 || F.line_text(0).rfind("libMesh", 0) != 0) throw std::runtime_error(path + ": not an ASCII XDA file");
    const std::string header = F.line_text(0);
    ShellMesh m;
    auto count_of = [&](const std::string &line, const char *what) -> long {
        try {
            const long v = std::stol(line);
            if (v < 0) throw std::out_of_range("negative");
            return v;
        } catch (const std::exception &) {
            throw std::runtime_error(path + ": bad " + what + " \"" + line.substr(0, 40) + "\"");
        }
    };
    const long n_elem = count_of(F.line_text(1), "element count"), n_nodes = count_of(F.line_text(2), "node count");
    size_t pos = 8;
    if (F.n_lines() < pos + (size_t)n_elem + (size_t)n_nodes + 1) throw std::runtime_error(path + ": truncated XDA file");
    m.order.reserve((size_t)n_elem);
    NumberCursor c{nullptr, nullptr};
    for (long e = 0; e < n_elem; e++, pos++) {
        F.line(pos, &c.p, &c.e);
        int type = 0;
        int32_t v[4] = {0, 0, 0, 0};
        bool ok = c.read(&type);
        if (ok && type == 3) {
            ok = c.read(&v[0]) && c.read(&v[1]) && c.read(&v[2]);
            m.order.push_back({'t', m.n_tri()});
            m.tri.insert(m.tri.end(), {v[0], v[1], v[2]});
        } else if (ok && type == 5) {
            ok = c.read(&v[0]) && c.read(&v[1]) && c.read(&v[2]) && c.read(&v[3]);
            m.order.push_back({'q', m.n_quad()});
            m.quad.insert(m.quad.end(), {v[0], v[1], v[2], v[3]});
        } else if (ok) {
            throw std::runtime_error(path + ": unsupported element type " + std::to_string(type) +
                                     " (only TRI3 = 3 and QUAD4 = 5)");
        }
        if (!ok) throw std::runtime_error(path + ": bad element line " + std::to_string(e));
    }
    m.xyz.resize((size_t)n_nodes * 3);
    for (long n = 0; n < n_nodes; n++, pos++) {
        F.line(pos, &c.p, &c.e);
        double *x = &m.xyz[3 * (size_t)n];
        if (!(c.read(&x[0]) && c.read(&x[1]) && c.read(&x[2]))) throw std::runtime_error(path + ": bad node line " + std::to_string(n));
    }
    const long n_bc = count_of(F.line_text(pos++), "boundary condition count");
    for (long b = 0; b < n_bc && pos < F.n_lines(); b++, pos++) {
        F.line(pos, &c.p, &c.e);
        SideBC bc;
        if (!(c.read(&bc.elem) && c.read(&bc.side) && c.read(&bc.id))) throw std::runtime_error(path + ": bad boundary line " + std::to_string(b));
        if (bc.elem < 0 || bc.elem >= n_elem) throw std::runtime_error(path + ": boundary element out of range");
        if (bc.side < 0 || bc.side >= (int32_t)m.element_nodes(bc.elem).size())
            throw std::runtime_error(path + ": boundary line " + std::to_string(b) + " names side " + std::to_string(bc.side) +
                                     " of an element with " + std::to_string(m.element_nodes(bc.elem).size()) + " sides");
        m.bcs.push_back(bc);
    }
    if (header.rfind("libMesh-0.9.2+", 0) == 0 && pos < F.n_lines() && F.line_text(pos).find_first_not_of(" \t\r\n") != std::string::npos) {
        const long n_ns = count_of(F.line_text(pos++), "nodeset count"); // nodesets: (node, boundary id)
        for (long b = 0; b < n_ns && pos < F.n_lines(); b++, pos++) {
            F.line(pos, &c.p, &c.e);
            int32_t node = 0, id = 0;
            if (!(c.read(&node) && c.read(&id)) || node < 0 || node >= n_nodes) throw std::runtime_error(path + ": bad nodeset line " + std::to_string(b));
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
    // bytes between the read position and the end of the file
    uint64_t remaining()
    {
        const std::streampos here = in.tellg();
        in.seekg(0, std::ios::end);
        const std::streampos end = in.tellg();
        in.seekg(here);
        return (here < 0 || end < here) ? 0 : (uint64_t)(end - here);
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
    // the counts come from the file: hold them against its size before anything is allocated for them (an element record is
    // at least 16 bytes, a node 24)
    if ((uint64_t)n_elem * 16u + (uint64_t)n_nodes * 24u > x.remaining())
        throw std::runtime_error(path + ": header announces " + std::to_string(n_elem) + " elements and " + std::to_string(n_nodes) +
                                 " nodes, more than the file holds (truncated, or not an XDR mesh file)");
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
    const TextFile F(path);
    NumberCursor c{F.buf.data(), F.buf.data() + F.buf.size()}; // (numbers in sequence, line breaks are blanks like any other)
    long n = 0;
    double factor = 1.0;
    if (!(c.read(&n) && c.read(&factor))) throw std::runtime_error(path + ": bad force file header");
    std::vector<double> out((size_t)n_nodes * 6, 0.0);
    bool more = true; // (a file that ends early, or holds something that is no number, leaves the rest zero)
    for (long i = 0; i < n && i < n_nodes && more; i++)
        for (int j = 0; j < 6 && more; j++) {
            double v = 0.0;
            more = c.read(&v);
            if (more) out[(size_t)i * 6 + j] = v * factor;
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

// VTK XML output as libMesh's VTKIO names it (fem-shell_precice.cpp:1552-1559: <out>_NNN.pvtu per converged time step of a
// serial run): the .pvtu index and one piece, <stem>_0.vtu, with the displaced nodes and the six nodal variables.  ASCII
// data arrays; no VTK library here to read it back -- the tests parse the XML and compare the numbers.
void write_pvtu(const ShellMesh &m, const std::vector<double> &u, const std::string &path)
{
    if (path.size() < 5 || path.substr(path.size() - 5) != ".pvtu") throw std::runtime_error("write_pvtu: the name must end in .pvtu");
    const std::string stem = path.substr(0, path.size() - 5), piece = stem + "_0.vtu";
    const std::string piece_name = piece.substr(piece.find_last_of('/') == std::string::npos ? 0 : piece.find_last_of('/') + 1);
    static const char *names[6] = {"u", "v", "w", "tx", "ty", "tz"};
    {
        std::ofstream os(path);
        if (!os) throw std::runtime_error("cannot write " + path);
        os << "<?xml version=\"1.0\"?>\n<VTKFile type=\"PUnstructuredGrid\" version=\"0.1\" byte_order=\"LittleEndian\">\n"
           << "  <PUnstructuredGrid GhostLevel=\"0\">\n    <PPointData>\n";
        for (const char *nm : names) os << "      <PDataArray type=\"Float64\" Name=\"" << nm << "\"/>\n";
        os << "    </PPointData>\n    <PPoints>\n      <PDataArray type=\"Float64\" NumberOfComponents=\"3\"/>\n    </PPoints>\n"
           << "    <Piece Source=\"" << piece_name << "\"/>\n  </PUnstructuredGrid>\n</VTKFile>\n";
    }
    std::ofstream os(piece);
    if (!os) throw std::runtime_error("cannot write " + piece);
    os.precision(17);
    const int32_t nn = m.n_nodes();
    const long ne = m.n_tri() + m.n_quad();
    os << "<?xml version=\"1.0\"?>\n<VTKFile type=\"UnstructuredGrid\" version=\"0.1\" byte_order=\"LittleEndian\">\n  <UnstructuredGrid>\n"
       << "    <Piece NumberOfPoints=\"" << nn << "\" NumberOfCells=\"" << ne << "\">\n      <PointData>\n";
    for (int v = 0; v < 6; v++) {
        os << "        <DataArray type=\"Float64\" Name=\"" << names[v] << "\" format=\"ascii\">\n";
        for (int32_t n = 0; n < nn; n++) os << u[6 * (size_t)n + v] << "\n";
        os << "        </DataArray>\n";
    }
    os << "      </PointData>\n      <Points>\n        <DataArray type=\"Float64\" NumberOfComponents=\"3\" format=\"ascii\">\n";
    for (int32_t n = 0; n < nn; n++)
        os << m.xyz[3 * n] + u[6 * (size_t)n] << " " << m.xyz[3 * n + 1] + u[6 * (size_t)n + 1] << " " << m.xyz[3 * n + 2] + u[6 * (size_t)n + 2] << "\n";
    os << "        </DataArray>\n      </Points>\n      <Cells>\n        <DataArray type=\"Int32\" Name=\"connectivity\" format=\"ascii\">\n";
    for (long e = 0; e < ne; e++) {
        for (int32_t v : m.element_nodes((int32_t)e)) os << v << " ";
        os << "\n";
    }
    os << "        </DataArray>\n        <DataArray type=\"Int32\" Name=\"offsets\" format=\"ascii\">\n";
    long off = 0;
    for (long e = 0; e < ne; e++) os << (off += (long)m.element_nodes((int32_t)e).size()) << "\n";
    os << "        </DataArray>\n        <DataArray type=\"UInt8\" Name=\"types\" format=\"ascii\">\n";
    for (long e = 0; e < ne; e++) os << (m.element_nodes((int32_t)e).size() == 3 ? 5 : 9) << "\n";
    os << "        </DataArray>\n      </Cells>\n    </Piece>\n  </UnstructuredGrid>\n</VTKFile>\n";
}

// ---- ExodusII output (fem-shell.cpp:1240-1251: ExodusII_IO(mesh).write_equation_systems(out + ".e", es)) ------------
// ExodusII is a set of conventions on a netCDF file; the image has no netCDF library, but the classic netCDF format
// (here "64-bit offset", CDF-2) is a header of dimensions, attributes and variables followed by the big-endian data, and
// a writer for the handful of variables an ExodusII mesh with nodal fields needs is short.  What is written is what
// libMesh's writer produces for this program in content: the displaced nodes (the reference adds the displacements to
// the nodes before writing, fem-shell.cpp:172-175), one element block per element type (TRI3, QUAD4; 1-based
// connectivity), the original element numbers as elem_num_map, and the six nodal variables u, v, w, tx, ty, tz of the
// "Elasticity" system (fem-shell.cpp:75-80) at one time step.  tests/test_host_tools.py reads the file back with
// scipy.io.netcdf_file; no libMesh / ParaView exists here to open it, which is all the pinning this format gets.
namespace {

struct NcWriter {
    std::vector<unsigned char> hdr;
    void i32(int32_t v)
    {
        for (int s = 24; s >= 0; s -= 8) hdr.push_back((unsigned char)((uint32_t)v >> s));
    }
    void i64(int64_t v)
    {
        for (int s = 56; s >= 0; s -= 8) hdr.push_back((unsigned char)((uint64_t)v >> s));
    }
    void name(const std::string &n)
    {
        i32((int32_t)n.size());
        hdr.insert(hdr.end(), n.begin(), n.end());
        while (hdr.size() % 4) hdr.push_back(0);
    }
};

enum { NC_CHAR = 2, NC_INT = 4, NC_FLOAT = 5, NC_DOUBLE = 6 };

struct NcAtt {
    std::string name;
    int type;
    std::string text;          // NC_CHAR
    std::vector<int32_t> ints; // NC_INT
    std::vector<float> floats; // NC_FLOAT
};

struct NcVar {
    std::string name;
    int type;
    std::vector<int> dims;      // dimension ids; a leading record dimension makes it a record variable
    std::vector<NcAtt> atts;
    std::vector<unsigned char> data; // big-endian payload (record variables: one record)
    int64_t begin = 0;
};

void put_be(std::vector<unsigned char> &d, int32_t v)
{
    for (int s = 24; s >= 0; s -= 8) d.push_back((unsigned char)((uint32_t)v >> s));
}
void put_be(std::vector<unsigned char> &d, double x)
{
    uint64_t v;
    std::memcpy(&v, &x, 8);
    for (int s = 56; s >= 0; s -= 8) d.push_back((unsigned char)(v >> s));
}

void write_atts(NcWriter &w, const std::vector<NcAtt> &atts)
{
    if (atts.empty()) {
        w.i32(0);
        w.i32(0);
        return;
    }
    w.i32(0x0C);
    w.i32((int32_t)atts.size());
    for (const NcAtt &a : atts) {
        w.name(a.name);
        w.i32(a.type);
        if (a.type == NC_CHAR) {
            w.i32((int32_t)a.text.size());
            w.hdr.insert(w.hdr.end(), a.text.begin(), a.text.end());
            while (w.hdr.size() % 4) w.hdr.push_back(0);
        } else if (a.type == NC_INT) {
            w.i32((int32_t)a.ints.size());
            for (int32_t v : a.ints) w.i32(v);
        } else {
            w.i32((int32_t)a.floats.size());
            for (float f : a.floats) {
                uint32_t v;
                std::memcpy(&v, &f, 4);
                w.i32((int32_t)v);
            }
        }
    }
}

} // namespace

void write_exodus(const ShellMesh &m, const std::vector<double> &u, const std::string &path)
{
    const int32_t nn = m.n_nodes(), nt = m.n_tri(), nq = m.n_quad(), ne = nt + nq;
    if ((int64_t)u.size() != 6ll * nn) throw std::runtime_error("write_exodus: solution vector has the wrong length");
    // dimensions
    struct Dim { std::string name; int32_t len; };
    std::vector<Dim> dims = {{"len_string", 33}, {"len_line", 81}, {"four", 4}, {"len_name", 33}, {"time_step", 0},
                             {"num_dim", 3}, {"num_nodes", nn}, {"num_elem", ne}, {"num_el_blk", (nt > 0) + (nq > 0)},
                             {"num_nod_var", 6}};
    enum { D_STRING = 0, D_LINE, D_FOUR, D_NAME, D_TIME, D_DIM, D_NODES, D_ELEM, D_BLK, D_NODVAR };
    std::vector<NcVar> vars;
    auto chars = [](const std::vector<std::string> &rows, int width) {
        std::vector<unsigned char> d;
        for (const std::string &r : rows)
            for (int i = 0; i < width; i++) d.push_back(i < (int)r.size() ? (unsigned char)r[(size_t)i] : 0);
        return d;
    };
    {
        NcVar v{"time_whole", NC_DOUBLE, {D_TIME}, {}, {}};
        put_be(v.data, 0.0);
        vars.push_back(v);
    }
    const int nblk = dims[D_BLK].len;
    {
        NcVar st{"eb_status", NC_INT, {D_BLK}, {}, {}}, pr{"eb_prop1", NC_INT, {D_BLK}, {{"name", NC_CHAR, "ID", {}, {}}}, {}};
        for (int b = 0; b < nblk; b++) {
            put_be(st.data, (int32_t)1);
            put_be(pr.data, (int32_t)(b + 1));
        }
        vars.push_back(st);
        vars.push_back(pr);
        NcVar names{"eb_names", NC_CHAR, {D_BLK, D_NAME}, {}, chars(std::vector<std::string>((size_t)nblk, ""), 33)};
        vars.push_back(names);
    }
    for (int d = 0; d < 3; d++) {
        NcVar c{std::string("coord") + "xyz"[d], NC_DOUBLE, {D_NODES}, {}, {}};
        c.data.reserve((size_t)nn * 8);
        for (int32_t n = 0; n < nn; n++) put_be(c.data, m.xyz[3 * (size_t)n + d] + u[6 * (size_t)n + d]); // displaced mesh
        vars.push_back(std::move(c));
    }
    vars.push_back(NcVar{"coor_names", NC_CHAR, {D_DIM, D_NAME}, {}, chars({"x", "y", "z"}, 33)});
    // element blocks: triangles first, then quadrilaterals; elem_num_map keeps the file-order numbers
    std::vector<int32_t> order_t, order_q;
    for (int32_t e = 0; e < ne; e++) {
        const bool tri = m.order.empty() ? e < nt : m.order[(size_t)e].first == 't';
        (tri ? order_t : order_q).push_back(e);
    }
    int blk = 0;
    for (int kind = 0; kind < 2; kind++) {
        const std::vector<int32_t> &lst = kind == 0 ? order_t : order_q;
        if (lst.empty()) continue;
        blk++;
        const int npe = kind == 0 ? 3 : 4;
        dims.push_back({"num_el_in_blk" + std::to_string(blk), (int32_t)lst.size()});
        dims.push_back({"num_nod_per_el" + std::to_string(blk), npe});
        NcVar c{"connect" + std::to_string(blk), NC_INT, {(int)dims.size() - 2, (int)dims.size() - 1},
                {{"elem_type", NC_CHAR, kind == 0 ? "TRI3" : "QUAD4", {}, {}}}, {}};
        c.data.reserve(lst.size() * (size_t)npe * 4);
        for (int32_t e : lst)
            for (int32_t nd : m.element_nodes(e)) put_be(c.data, (int32_t)(nd + 1));
        vars.push_back(std::move(c));
    }
    {
        NcVar em{"elem_num_map", NC_INT, {D_ELEM}, {}, {}}, nm{"node_num_map", NC_INT, {D_NODES}, {}, {}};
        for (int32_t e : order_t) put_be(em.data, (int32_t)(e + 1));
        for (int32_t e : order_q) put_be(em.data, (int32_t)(e + 1));
        for (int32_t n = 0; n < nn; n++) put_be(nm.data, (int32_t)(n + 1));
        vars.push_back(std::move(em));
        vars.push_back(std::move(nm));
    }
    vars.push_back(NcVar{"name_nod_var", NC_CHAR, {D_NODVAR, D_NAME}, {}, chars({"u", "v", "w", "tx", "ty", "tz"}, 33)});
    for (int v = 0; v < 6; v++) {
        NcVar nv{"vals_nod_var" + std::to_string(v + 1), NC_DOUBLE, {D_TIME, D_NODES}, {}, {}};
        nv.data.reserve((size_t)nn * 8);
        for (int32_t n = 0; n < nn; n++) put_be(nv.data, u[6 * (size_t)n + v]);
        vars.push_back(std::move(nv));
    }
    const std::vector<NcAtt> gatts = {{"api_version", NC_FLOAT, "", {}, {5.22f}}, {"version", NC_FLOAT, "", {}, {5.22f}},
                                      {"floating_point_word_size", NC_INT, "", {8}, {}}, {"file_size", NC_INT, "", {1}, {}},
                                      {"maximum_name_length", NC_INT, "", {32}, {}},
                                      {"title", NC_CHAR, "fem-shell displaced mesh (libfemshell, MI355X)", {}, {}}};
    auto padded = [](size_t n) { return (n + 3) & ~(size_t)3; };
    auto is_record = [&](const NcVar &v) { return !v.dims.empty() && v.dims[0] == D_TIME; };
    // two passes over the header: the first fixes its length, the second knows every variable's offset
    int64_t header_len = 0;
    std::vector<unsigned char> header;
    for (int pass = 0; pass < 2; pass++) {
        int64_t off = header_len;
        for (NcVar &v : vars)
            if (!is_record(v)) {
                v.begin = off;
                off += (int64_t)padded(v.data.size());
            }
        for (NcVar &v : vars)
            if (is_record(v)) {
                v.begin = off;
                off += (int64_t)padded(v.data.size());
            }
        NcWriter w;
        w.hdr = {'C', 'D', 'F', 2};
        w.i32(1); // one record (time step)
        w.i32(0x0A);
        w.i32((int32_t)dims.size());
        for (const Dim &d : dims) {
            w.name(d.name);
            w.i32(d.len);
        }
        write_atts(w, gatts);
        w.i32(0x0B);
        w.i32((int32_t)vars.size());
        for (const NcVar &v : vars) {
            w.name(v.name);
            w.i32((int32_t)v.dims.size());
            for (int d : v.dims) w.i32(d);
            write_atts(w, v.atts);
            w.i32(v.type);
            w.i32((int32_t)padded(v.data.size())); // vsize (record variables: one record)
            w.i64(v.begin);
        }
        header_len = (int64_t)w.hdr.size();
        header.swap(w.hdr);
    }
    std::ofstream os(path, std::ios::binary);
    if (!os) throw std::runtime_error("cannot write " + path);
    os.write(reinterpret_cast<const char *>(header.data()), (std::streamsize)header.size());
    static const char zeros[4] = {0, 0, 0, 0};
    for (int rec = 0; rec < 2; rec++)
        for (const NcVar &v : vars)
            if (is_record(v) == (rec == 1)) {
                os.write(reinterpret_cast<const char *>(v.data.data()), (std::streamsize)v.data.size());
                os.write(zeros, (std::streamsize)(padded(v.data.size()) - v.data.size()));
            }
    if (!os) throw std::runtime_error("cannot write " + path);
}

} // namespace femshell_host
