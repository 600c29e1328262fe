#include "trianglemesh.h"

#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>

namespace {

bool header_count(const std::string &hdr, const std::string &key, int *n) {
    const size_t p = hdr.find(key);
    if (p == std::string::npos) return false;
    std::istringstream ss(hdr.substr(p + key.size()));
    ss >> *n;
    return !ss.fail() && *n >= 0;
}

}  // namespace

bool TriangleMesh::loadPLY(const std::string &filename) {
    std::ifstream f(filename.c_str(), std::ios::in | std::ios::binary);
    if (!f.is_open()) return false;
    std::string data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const std::string endTag("end_header\n");
    const size_t e = data.find(endTag);
    if (e == std::string::npos) return false;
    const std::string hdr = data.substr(0, e + endTag.size());
    if (hdr.find("format binary_little_endian 1.0") == std::string::npos) return false;
    int nv = 0, nf = 0;
    if (!header_count(hdr, "element vertex ", &nv) || !header_count(hdr, "element face ", &nf)) return false;
    const bool colors = hdr.find("property uchar red\nproperty uchar green\nproperty uchar blue\n") != std::string::npos;
    const size_t vsize = 12 + (colors ? 3 : 0), fsize = 1 + 12;
    const size_t voff = hdr.size(), foff = voff + (size_t)nv * vsize;
    if (data.size() < foff + (size_t)nf * fsize) return false;

    std::vector<vmath::vec3> verts((size_t)nv), cols;
    if (colors) cols.resize((size_t)nv);
    for (int i = 0; i < nv; i++) {
        const char *p = data.data() + voff + (size_t)i * vsize;
        float xyz[3];
        memcpy(xyz, p, 12);
        verts[i] = vmath::vec3(xyz[0], xyz[1], xyz[2]);
        if (colors) {
            const unsigned char *c = (const unsigned char *)p + 12;
            cols[i] = vmath::vec3(c[0] / 255.0f, c[1] / 255.0f, c[2] / 255.0f);
        }
    }
    std::vector<Triangle> tris((size_t)nf);
    for (int i = 0; i < nf; i++) {
        const char *p = data.data() + foff + (size_t)i * fsize;
        if ((unsigned char)p[0] != 3) return false;
        int32_t idx[3];
        memcpy(idx, p + 1, 12);
        for (int q = 0; q < 3; q++)
            if (idx[q] < 0 || idx[q] >= nv) return false;
        tris[i] = Triangle(idx[0], idx[1], idx[2]);
    }
    vertices.swap(verts);
    vertexcolors.swap(cols);
    triangles.swap(tris);
    return true;
}

void TriangleMesh::writeMeshToPLY(const std::string &filename) const {
    std::ostringstream h;
    h << "ply\nformat binary_little_endian 1.0\nelement vertex " << vertices.size()
      << "\nproperty float x\nproperty float y\nproperty float z\nelement face " << triangles.size()
      << "\nproperty list uchar int vertex_index\nend_header\n";
    std::ofstream out(filename.c_str(), std::ios::out | std::ios::binary);
    const std::string hs = h.str();
    out.write(hs.data(), (std::streamsize)hs.size());
    if (!vertices.empty()) out.write((const char *)vertices.data(), (std::streamsize)(vertices.size() * 12));
    for (const Triangle &t : triangles) {
        const char three = 3;
        out.write(&three, 1);
        int32_t idx[3] = {t.tri[0], t.tri[1], t.tri[2]};
        out.write((const char *)idx, 12);
    }
}

void TriangleMesh::writeMeshToOBJ(const std::string &filename) const {
    std::ostringstream s;
    s << "# OBJ file format with ext .obj\n# vertex count = " << vertices.size() << "\n# face count = "
      << triangles.size() << "\n";
    for (const vmath::vec3 &p : vertices) s << "v " << p.x << " " << p.y << " " << p.z << "\n";
    if (normals.size() == vertices.size())
        for (const vmath::vec3 &n : normals) s << "vn " << n.x << " " << n.y << " " << n.z << "\n";
    for (const Triangle &t : triangles) {
        const int a = t.tri[0] + 1, b = t.tri[1] + 1, c = t.tri[2] + 1;
        s << "f " << a << "//" << a << " " << b << "//" << b << " " << c << "//" << c << "\n";
    }
    std::ofstream out(filename.c_str());
    out << s.str();
}

void TriangleMesh::translate(vmath::vec3 t) {
    for (vmath::vec3 &v : vertices) v += t;
}
