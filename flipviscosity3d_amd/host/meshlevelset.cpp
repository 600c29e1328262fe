#include "meshlevelset.h"

#include <algorithm>
#include <cassert>
#include <cmath>

namespace {

inline size_t flat(int i, int j, int k, int w, int h) { return (size_t)i + (size_t)w * ((size_t)j + (size_t)h * (size_t)k); }
inline bool inRange(int i, int j, int k, int w, int h, int d) { return i >= 0 && j >= 0 && k >= 0 && i < w && j < h && k < d; }
inline int clampi(int v, int lo, int hi) { return std::max(lo, std::min(v, hi)); }

// distance from x0 to segment x1-x2 (reference meshlevelset.cpp:432-446)
float pointSegmentDistance(const vmath::vec3 &x0, const vmath::vec3 &x1, const vmath::vec3 &x2) {
    const vmath::vec3 dx = x2 - x1;
    const double m2 = vmath::lengthsq(dx);
    float s12 = (float)(vmath::dot(x2 - x0, dx) / m2);
    if (s12 < 0) s12 = 0;
    else if (s12 > 1) s12 = 1;
    return vmath::length(x0 - (s12 * x1 + (1 - s12) * x2));
}

// distance from x0 to triangle x1-x2-x3 (reference meshlevelset.cpp:349-393)
float pointTriangleDistance(const vmath::vec3 &x0, const vmath::vec3 &x1, const vmath::vec3 &x2, const vmath::vec3 &x3) {
    const vmath::vec3 x13 = x1 - x3, x23 = x2 - x3, x03 = x0 - x3;
    const float m13 = vmath::lengthsq(x13), m23 = vmath::lengthsq(x23), d = vmath::dot(x13, x23);
    const float invdet = 1.0f / std::fmax(m13 * m23 - d * d, 1e-30f);
    const float a = vmath::dot(x13, x03), b = vmath::dot(x23, x03);
    const float w23 = invdet * (m23 * a - d * b);
    const float w31 = invdet * (m13 * b - d * a);
    const float w12 = 1 - w23 - w31;
    if (w23 >= 0 && w31 >= 0 && w12 >= 0) return vmath::length(x0 - (w23 * x1 + w31 * x2 + w12 * x3));
    if (w23 > 0) return std::fmin(pointSegmentDistance(x0, x1, x2), pointSegmentDistance(x0, x1, x3));
    if (w31 > 0) return std::fmin(pointSegmentDistance(x0, x1, x2), pointSegmentDistance(x0, x2, x3));
    return std::fmin(pointSegmentDistance(x0, x1, x3), pointSegmentDistance(x0, x2, x3));
}

// twice the signed area of (0,0)-(x1,y1)-(x2,y2) with a simulation-of-simplicity sign
// (reference meshlevelset.cpp:448-470)
int orientation(double x1, double y1, double x2, double y2, double *area2) {
    *area2 = y1 * x2 - x1 * y2;
    if (*area2 > 0) return 1;
    if (*area2 < 0) return -1;
    if (y2 > y1) return 1;
    if (y2 < y1) return -1;
    if (x1 > x2) return 1;
    if (x1 < x2) return -1;
    return 0;
}

// robust point-in-triangle test in 2-D with barycentric coordinates (reference meshlevelset.cpp:395-430)
bool barycentric(double x0, double y0, double x1, double y1, double x2, double y2, double x3, double y3, double *a,
                 double *b, double *c) {
    x1 -= x0; x2 -= x0; x3 -= x0;
    y1 -= y0; y2 -= y0; y3 -= y0;
    double oa, ob, oc;
    const int sa = orientation(x2, y2, x3, y3, &oa);
    if (sa == 0) return false;
    if (orientation(x3, y3, x1, y1, &ob) != sa) return false;
    if (orientation(x1, y1, x2, y2, &oc) != sa) return false;
    const double sum = oa + ob + oc;
    assert(sum != 0);
    const double inv = 1.0 / sum;
    *a = oa * inv; *b = ob * inv; *c = oc * inv;
    return true;
}

}  // namespace

double trilinearInterpolateField(vmath::vec3 p, double dx, const float *g, int w, int h, int d) {
    const double invdx = 1.0 / dx;
    const int gi = (int)std::floor(p.x * invdx), gj = (int)std::floor(p.y * invdx), gk = (int)std::floor(p.z * invdx);
    const float gx = (float)(gi * dx), gy = (float)(gj * dx), gz = (float)(gk * dx);
    const double ix = (p.x - gx) * invdx, iy = (p.y - gy) * invdx, iz = (p.z - gz) * invdx;
    double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (inRange(gi, gj, gk, w, h, d)) c[0] = g[flat(gi, gj, gk, w, h)];
    if (inRange(gi + 1, gj, gk, w, h, d)) c[1] = g[flat(gi + 1, gj, gk, w, h)];
    if (inRange(gi, gj + 1, gk, w, h, d)) c[2] = g[flat(gi, gj + 1, gk, w, h)];
    if (inRange(gi, gj, gk + 1, w, h, d)) c[3] = g[flat(gi, gj, gk + 1, w, h)];
    if (inRange(gi + 1, gj, gk + 1, w, h, d)) c[4] = g[flat(gi + 1, gj, gk + 1, w, h)];
    if (inRange(gi, gj + 1, gk + 1, w, h, d)) c[5] = g[flat(gi, gj + 1, gk + 1, w, h)];
    if (inRange(gi + 1, gj + 1, gk, w, h, d)) c[6] = g[flat(gi + 1, gj + 1, gk, w, h)];
    if (inRange(gi + 1, gj + 1, gk + 1, w, h, d)) c[7] = g[flat(gi + 1, gj + 1, gk + 1, w, h)];
    return c[0] * (1 - ix) * (1 - iy) * (1 - iz) + c[1] * ix * (1 - iy) * (1 - iz) + c[2] * (1 - ix) * iy * (1 - iz) +
           c[3] * (1 - ix) * (1 - iy) * iz + c[4] * ix * (1 - iy) * iz + c[5] * (1 - ix) * iy * iz +
           c[6] * ix * iy * (1 - iz) + c[7] * ix * iy * iz;
}

void trilinearInterpolateFieldGradient(vmath::vec3 p, double dx, const float *g, int w, int h, int d, vmath::vec3 *grad) {
    const double invdx = 1.0 / dx;
    const int gi = (int)std::floor(p.x * invdx), gj = (int)std::floor(p.y * invdx), gk = (int)std::floor(p.z * invdx);
    const float gx = (float)(gi * dx), gy = (float)(gj * dx), gz = (float)(gk * dx);
    const double ix = (p.x - gx) * invdx, iy = (p.y - gy) * invdx, iz = (p.z - gz) * invdx;
    float v[2][2][2] = {{{0, 0}, {0, 0}}, {{0, 0}, {0, 0}}};  // v[di][dj][dk]
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++)
            for (int c = 0; c < 2; c++)
                if (inRange(gi + a, gj + b, gk + c, w, h, d)) v[a][b][c] = g[flat(gi + a, gj + b, gk + c, w, h)];
    auto bil = [](double v00, double v10, double v01, double v11, double s, double t) {
        const double l1 = (1 - s) * v00 + s * v10, l2 = (1 - s) * v01 + s * v11;
        return (1 - t) * l1 + t * l2;
    };
    const float ddx00 = v[1][0][0] - v[0][0][0], ddx10 = v[1][1][0] - v[0][1][0], ddx01 = v[1][0][1] - v[0][0][1], ddx11 = v[1][1][1] - v[0][1][1];
    grad->x = (float)bil(ddx00, ddx10, ddx01, ddx11, iy, iz);
    const float ddy00 = v[0][1][0] - v[0][0][0], ddy10 = v[1][1][0] - v[1][0][0], ddy01 = v[0][1][1] - v[0][0][1], ddy11 = v[1][1][1] - v[1][0][1];
    grad->y = (float)bil(ddy00, ddy10, ddy01, ddy11, ix, iz);
    const float ddz00 = v[0][0][1] - v[0][0][0], ddz10 = v[1][0][1] - v[1][0][0], ddz01 = v[0][1][1] - v[0][1][0], ddz11 = v[1][1][1] - v[1][1][0];
    grad->z = (float)bil(ddz00, ddz10, ddz01, ddz11, ix, iy);
}

MeshLevelSet::MeshLevelSet(int isize, int jsize, int ksize, double dx)
    : _isize(isize), _jsize(jsize), _ksize(ksize), _dx(dx), _phi(isize + 1, jsize + 1, ksize + 1, 0.0f),
      _closest(isize + 1, jsize + 1, ksize + 1, -1) {}

float MeshLevelSet::getDistanceAtCellCenter(int i, int j, int k) const {
    const float *p = _phi.getRawArray();
    const int w = _phi.width, h = _phi.height;
    return 0.125f * (p[flat(i, j, k, w, h)] + p[flat(i + 1, j, k, w, h)] + p[flat(i, j + 1, k, w, h)] +
                     p[flat(i + 1, j + 1, k, w, h)] + p[flat(i, j, k + 1, w, h)] + p[flat(i + 1, j, k + 1, w, h)] +
                     p[flat(i, j + 1, k + 1, w, h)] + p[flat(i + 1, j + 1, k + 1, w, h)]);
}

float MeshLevelSet::trilinearInterpolate(vmath::vec3 pos) const {
    return (float)trilinearInterpolateField(pos, _dx, _phi.getRawArray(), _phi.width, _phi.height, _phi.depth);
}

vmath::vec3 MeshLevelSet::trilinearInterpolateGradient(vmath::vec3 pos) const {
    vmath::vec3 g;
    trilinearInterpolateFieldGradient(pos, _dx, _phi.getRawArray(), _phi.width, _phi.height, _phi.depth, &g);
    return g;
}

void MeshLevelSet::calculateSignedDistanceField(TriangleMesh &m, int bandwidth) {
    _mesh = m;
    std::vector<int> counts(_phi.size(), 0);
    _exactBand(bandwidth, counts);
    _propagate();
    _signs(counts);
}

void MeshLevelSet::calculateUnion(MeshLevelSet &other) {
    int oi, oj, ok;
    other.getGridDimensions(&oi, &oj, &ok);
    if (oi != _isize || oj != _jsize || ok != _ksize) throw std::invalid_argument("MeshLevelSet::calculateUnion: grid mismatch");
    const int offset = (int)_mesh.vertices.size();
    const TriangleMesh *om = other.getTriangleMesh();
    _mesh.vertices.insert(_mesh.vertices.end(), om->vertices.begin(), om->vertices.end());
    _mesh.triangles.reserve(_mesh.triangles.size() + om->triangles.size());
    for (const Triangle &t : om->triangles) _mesh.triangles.push_back(Triangle(t.tri[0] + offset, t.tri[1] + offset, t.tri[2] + offset));
    float *p = _phi.getRawArray();
    int *c = _closest.getRawArray();
    const float *q = other._phi.getRawArray();
    const int *oc = other._closest.getRawArray();
    const size_t n = _phi.size();
    for (size_t t = 0; t < n; t++)
        if (q[t] < p[t]) {
            p[t] = q[t];
            c[t] = oc[t] + offset;
        }
}

void MeshLevelSet::negate() {
    float *p = _phi.getRawArray();
    const size_t n = _phi.size();
    for (size_t t = 0; t < n; t++) p[t] = -p[t];
}

// reference meshlevelset.cpp:196-268
void MeshLevelSet::_exactBand(int band, std::vector<int> &counts) {
    const int w = _phi.width, h = _phi.height, d = _phi.depth;
    _phi.fill((float)((w + h + d) * _dx));
    _closest.fill(-1);
    float *phi = _phi.getRawArray();
    int *closest = _closest.getRawArray();
    const double invdx = 1.0 / _dx;
    for (size_t tidx = 0; tidx < _mesh.triangles.size(); tidx++) {
        const Triangle &t = _mesh.triangles[tidx];
        const vmath::vec3 p = _mesh.vertices[t.tri[0]], q = _mesh.vertices[t.tri[1]], r = _mesh.vertices[t.tri[2]];
        const double fip = (double)p.x * invdx, fjp = (double)p.y * invdx, fkp = (double)p.z * invdx;
        const double fiq = (double)q.x * invdx, fjq = (double)q.y * invdx, fkq = (double)q.z * invdx;
        const double fir = (double)r.x * invdx, fjr = (double)r.y * invdx, fkr = (double)r.z * invdx;
        const double imin = std::fmin(fip, std::fmin(fiq, fir)), imax = std::fmax(fip, std::fmax(fiq, fir));
        const double jmin = std::fmin(fjp, std::fmin(fjq, fjr)), jmax = std::fmax(fjp, std::fmax(fjq, fjr));
        const double kmin = std::fmin(fkp, std::fmin(fkq, fkr)), kmax = std::fmax(fkp, std::fmax(fkq, fkr));
        int i0 = clampi(int(imin) - band, 0, w - 1), i1 = clampi(int(imax) + band + 1, 0, w - 1);
        int j0 = clampi(int(jmin) - band, 0, h - 1), j1 = clampi(int(jmax) + band + 1, 0, h - 1);
        int k0 = clampi(int(kmin) - band, 0, d - 1), k1 = clampi(int(kmax) + band + 1, 0, d - 1);
        for (int k = k0; k <= k1; k++)
            for (int j = j0; j <= j1; j++)
                for (int i = i0; i <= i1; i++) {
                    const vmath::vec3 gpos((float)(i * _dx), (float)(j * _dx), (float)(k * _dx));
                    const float dist = pointTriangleDistance(gpos, p, q, r);
                    const size_t c = flat(i, j, k, w, h);
                    if (dist < phi[c]) {
                        phi[c] = dist;
                        closest[c] = (int)tidx;
                    }
                }
        // intersection counts along i for the parity sign
        j0 = clampi((int)std::ceil(jmin), 0, h - 1);
        k0 = clampi((int)std::ceil(kmin), 0, d - 1);
        j1 = clampi((int)std::floor(jmax), 0, h - 1);
        k1 = clampi((int)std::floor(kmax), 0, d - 1);
        for (int k = k0; k <= k1; k++)
            for (int j = j0; j <= j1; j++) {
                double a, b, c;
                if (barycentric(j, k, fjp, fkp, fjq, fkq, fjr, fkr, &a, &b, &c)) {
                    const double fi = a * fip + b * fiq + c * fir;
                    const int interval = int(std::ceil(fi));
                    if (interval < 0) counts[flat(0, j, k, w, h)] += 1;
                    else if (interval < w) counts[flat(interval, j, k, w, h)] += 1;
                }
            }
    }
}

// reference meshlevelset.cpp:270-329: breadth-first order from the band, then one pass in that order
void MeshLevelSet::_propagate() {
    const int w = _phi.width, h = _phi.height, d = _phi.depth;
    const size_t n = _phi.size();
    float *phi = _phi.getRawArray();
    int *closest = _closest.getRawArray();
    std::vector<unsigned> queue;
    queue.reserve(n);
    std::vector<unsigned char> seen(n, 0);
    for (size_t c = 0; c < n; c++)
        if (closest[c] != -1) {
            seen[c] = 1;
            queue.push_back((unsigned)c);
        }
    const size_t unknownStart = queue.size();
    const long sy = w, sz = (long)w * h;
    auto coords = [&](unsigned c, int &i, int &j, int &k) {
        i = (int)(c % (unsigned)w);
        const unsigned r = c / (unsigned)w;
        j = (int)(r % (unsigned)h);
        k = (int)(r / (unsigned)h);
    };
    for (size_t head = 0; head < queue.size(); head++) {
        const unsigned c = queue[head];
        int i, j, k;
        coords(c, i, j, k);
        const bool ok[6] = {i > 0, i < w - 1, j > 0, j < h - 1, k > 0, k < d - 1};
        const long off[6] = {-1, 1, -sy, sy, -sz, sz};
        for (int q = 0; q < 6; q++) {
            if (!ok[q]) continue;
            const size_t nb = (size_t)((long)c + off[q]);
            if (!seen[nb]) {
                seen[nb] = 1;
                queue.push_back((unsigned)nb);
            }
        }
    }
    for (size_t head = unknownStart; head < queue.size(); head++) {
        const unsigned c = queue[head];
        int i, j, k;
        coords(c, i, j, k);
        const vmath::vec3 gpos((float)(i * _dx), (float)(j * _dx), (float)(k * _dx));
        const bool ok[6] = {i > 0, i < w - 1, j > 0, j < h - 1, k > 0, k < d - 1};
        const long off[6] = {-1, 1, -sy, sy, -sz, sz};
        for (int q = 0; q < 6; q++) {
            if (!ok[q]) continue;
            const size_t nb = (size_t)((long)c + off[q]);
            const int tri = closest[nb];
            if (tri == -1) continue;
            const Triangle &t = _mesh.triangles[tri];
            const double dist = pointTriangleDistance(gpos, _mesh.vertices[t.tri[0]], _mesh.vertices[t.tri[1]], _mesh.vertices[t.tri[2]]);
            if (dist < phi[c]) {
                phi[c] = (float)dist;
                closest[c] = tri;
            }
        }
    }
}

// reference meshlevelset.cpp:331-347
void MeshLevelSet::_signs(const std::vector<int> &counts) {
    const int w = _phi.width, h = _phi.height, d = _phi.depth;
    float *phi = _phi.getRawArray();
    for (int k = 0; k < d; k++)
        for (int j = 0; j < h; j++) {
            int total = 0;
            for (int i = 0; i < w; i++) {
                const size_t c = flat(i, j, k, w, h);
                total += counts[c];
                if (total % 2 == 1) phi[c] = -phi[c];
            }
        }
}
