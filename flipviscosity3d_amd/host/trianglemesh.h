// trianglemesh.h -- triangle soup + the file formats the reference's driver uses.
// Interface of the reference's TriangleMesh (reference trianglemesh.h:36-60) restricted to what the
// simulation and its frame loop need: public vertices/triangles, loadPLY, writeMeshToPLY/OBJ.
#pragma once
#include <string>
#include <vector>

#include "vmath.h"

struct Triangle {  // reference triangle.h:26-39
    int tri[3];
    Triangle() { tri[0] = tri[1] = tri[2] = 0; }
    Triangle(int a, int b, int c) { tri[0] = a; tri[1] = b; tri[2] = c; }
};

class TriangleMesh {
public:
    // binary little-endian PLY (reference trianglemesh.cpp:39-63, 426-615).  Unlike the reference this also
    // loads files shorter than its fixed 2048-byte header read (cube.ply, sheet.ply, cone.ply).
    bool loadPLY(const std::string &filename);
    void writeMeshToPLY(const std::string &filename) const;  // reference trianglemesh.cpp:190-343 (no colours)
    void writeMeshToOBJ(const std::string &filename) const;  // reference trianglemesh.cpp:381-418
    int numVertices() const { return (int)vertices.size(); }
    int numFaces() const { return (int)triangles.size(); }
    int numTriangles() const { return numFaces(); }
    void translate(vmath::vec3 t);

    std::vector<vmath::vec3> vertices;
    std::vector<vmath::vec3> vertexcolors;
    std::vector<vmath::vec3> normals;
    std::vector<Triangle> triangles;
};
