// Minimal vector types for the host mirror (the reference uses glm, which is not part of this build).
#pragma once
#include <cmath>
#include <cstdint>

struct vec3 {
    float x{0}, y{0}, z{0};
    constexpr vec3() = default;
    constexpr explicit vec3(float s) : x(s), y(s), z(s) {}
    constexpr vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    template<typename V, typename = decltype(V{}.x)> constexpr explicit vec3(const V &v) : x(float(v.x)), y(float(v.y)), z(float(v.z)) {}
    float &operator[](int i) { return i == 0 ? x : i == 1 ? y : z; }
    float operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
    bool operator==(const vec3 &) const = default;
};
struct dvec3 {
    double x{0}, y{0}, z{0};
    constexpr dvec3() = default;
    constexpr explicit dvec3(double s) : x(s), y(s), z(s) {}
    constexpr dvec3(double a, double b, double c) : x(a), y(b), z(c) {}
    constexpr dvec3(const vec3 &v) : x(v.x), y(v.y), z(v.z) {}
    double &operator[](int i) { return i == 0 ? x : i == 1 ? y : z; }
    double operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
    bool operator==(const dvec3 &) const = default;
};
struct quat { // w, x, y, z as glm::quat's constructor order
    float w{1}, x{0}, y{0}, z{0};
    bool operator==(const quat &) const = default;
};
struct mat3 { // column-major: m[col][row]
    float m[3][3]{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    float *operator[](int c) { return m[c]; }
    const float *operator[](int c) const { return m[c]; }
};

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return a * s; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a / length(a); }
inline dvec3 operator+(dvec3 a, dvec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline dvec3 operator-(dvec3 a, dvec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline dvec3 operator*(dvec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline dvec3 operator*(dvec3 a, dvec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
