// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Minimal f32 3-vector with the arithmetic order nalgebra 0.31 uses for Vector3<f32>
// (the reference's vector type, Cargo.toml:27): dot = (x*x' + y*y') + z*z',
// norm_squared = dot(self,self), normalize = component / norm, cross = textbook.
#pragma once
#include <cmath>
#include <cstdint>

namespace orc {

struct V3 {
    float x = 0.f, y = 0.f, z = 0.f;
    V3() = default;
    V3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static V3 repeat(float v) { return V3(v, v, v); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(float s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
inline V3 cmul(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float norm_squared(V3 a) { return dot(a, a); }
inline float norm(V3 a) { return std::sqrt(norm_squared(a)); }
inline V3 normalize(V3 a) { return a / norm(a); }
inline V3 cross(V3 a, V3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

constexpr float PI = 3.14159265358979323846f;  // std::f32::consts::PI
constexpr float INF = __builtin_huge_valf();

}  // namespace orc
