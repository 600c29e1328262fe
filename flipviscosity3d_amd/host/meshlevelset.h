// meshlevelset.h -- node-sampled signed distance field of a closed triangle mesh (host side, setup time).
//
// Same interface and same results as the reference's MeshLevelSet (reference meshlevelset.h:60-122,
// meshlevelset.cpp:138-347): exact distances in a band around the mesh, closest-triangle propagation over
// the rest of the grid, inside/outside by intersection parity along i.  The propagation is order dependent
// in the reference (a single breadth-first pass, :270-329), so the traversal order is kept; storage is flat
// arrays instead of bounds-checked Array3d accessors.
#pragma once
#include <vector>

#include "array3d.h"
#include "trianglemesh.h"

class MeshLevelSet {
public:
    MeshLevelSet() {}
    MeshLevelSet(int isize, int jsize, int ksize, double dx);

    float operator()(int i, int j, int k) const { return get(i, j, k); }
    float get(int i, int j, int k) const { return _phi.get(i, j, k); }
    int getClosestTriangleIndex(int i, int j, int k) const { return _closest.get(i, j, k); }
    float getDistanceAtCellCenter(int i, int j, int k) const;    // reference meshlevelset.cpp:66-76
    float trilinearInterpolate(vmath::vec3 pos) const;           // reference meshlevelset.cpp:82-84
    vmath::vec3 trilinearInterpolateGradient(vmath::vec3 pos) const;  // reference meshlevelset.cpp:86-90

    void getGridDimensions(int *i, int *j, int *k) const { *i = _isize; *j = _jsize; *k = _ksize; }
    TriangleMesh *getTriangleMesh() { return &_mesh; }

    void calculateSignedDistanceField(TriangleMesh &m, int bandwidth = 1);  // reference meshlevelset.cpp:138-150
    void calculateUnion(MeshLevelSet &levelset);                            // reference meshlevelset.cpp:152-184
    void negate();                                                          // reference meshlevelset.cpp:186-194

    float *getRawArray() { return _phi.getRawArray(); }   // (I+1)(J+1)(K+1) nodes, Array3d layout
    const float *getRawArray() const { return _phi.getRawArray(); }
    const int *getClosestRawArray() const { return _closest.getRawArray(); }

private:
    void _exactBand(int bandwidth, std::vector<int> &counts);
    void _propagate();
    void _signs(const std::vector<int> &counts);

    int _isize = 0, _jsize = 0, _ksize = 0;
    double _dx = 0.0;
    TriangleMesh _mesh;
    Array3d<float> _phi;
    Array3d<int> _closest;
};

// Interpolation::trilinearInterpolate / trilinearInterpolateGradient on a float grid
// (reference interpolation.cpp:68-108, 122-184): float position differences, fp64 weights,
// out-of-range corners read as 0.
double trilinearInterpolateField(vmath::vec3 p, double dx, const float *grid, int w, int h, int d);
void trilinearInterpolateFieldGradient(vmath::vec3 p, double dx, const float *grid, int w, int h, int d, vmath::vec3 *grad);
