// vmath.h -- the subset of the reference's vmath (reference vmath.h:30-100) that the FluidSimulation API
// surface exposes: a POD 3-float vector.  FluidParticle is {vec3 position, velocity} = 6 packed floats,
// which is what the C-ABI moves (include/flipv.h, flipv_upload_particles).
#pragma once
#include <cmath>

namespace vmath {

struct vec3 {
    float x, y, z;
    vec3() : x(0.0f), y(0.0f), z(0.0f) {}
    vec3(float xx, float yy, float zz) : x(xx), y(yy), z(zz) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(const vec3 &a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(float s, const vec3 &v) { return vec3(v.x * s, v.y * s, v.z * s); }
inline vec3 operator*(const vec3 &v, float s) { return vec3(v.x * s, v.y * s, v.z * s); }
inline vec3 operator/(const vec3 &v, float s) {  // multiply by the reciprocal like the reference (vmath.cpp:96-99)
    const float inv = 1.0f / s;
    return vec3(v.x * inv, v.y * inv, v.z * inv);
}
inline vec3 &operator+=(vec3 &a, const vec3 &b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline vec3 &operator-=(vec3 &a, const vec3 &b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
inline float dot(const vec3 &a, const vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float lengthsq(const vec3 &v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
inline float length(const vec3 &v) { return std::sqrt(lengthsq(v)); }

}  // namespace vmath

static_assert(sizeof(vmath::vec3) == 12, "vec3 must be 3 packed floats");
