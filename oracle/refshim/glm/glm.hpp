/* Restatement of the part of glm (g-truc/glm, un-vendored submodule of the reference: RAST/.gitmodules:1-3, no pinned
 * commit) that the reference's kernels use: vec3, vec4, mat3, dot, length, max, transpose.  glm semantics: matrices are
 * COLUMN-major, mat3(a,b,c, d,e,f, g,h,i) fills column 0 = (a,b,c), column 1 = (d,e,f), column 2 = (g,h,i); m[i] is
 * column i; M * v and A * B are the ordinary products; mat3(s) is s * identity.  TEST INFRASTRUCTURE ONLY. */
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#define GLM_FN __host__ __device__ inline
namespace glm {
struct vec3 {
  float x, y, z;
  GLM_FN vec3() : x(0), y(0), z(0) {}
  GLM_FN vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  GLM_FN explicit vec3(float s) : x(s), y(s), z(s) {}
  GLM_FN float& operator[](int i) { return (&x)[i]; }
  GLM_FN const float& operator[](int i) const { return (&x)[i]; }
  GLM_FN vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
  GLM_FN vec3& operator+=(float s) { x += s; y += s; z += s; return *this; }
  GLM_FN vec3& operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
};
struct vec4 {
  float x, y, z, w;
  GLM_FN vec4() : x(0), y(0), z(0), w(0) {}
  GLM_FN vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
  GLM_FN float& operator[](int i) { return (&x)[i]; }
  GLM_FN const float& operator[](int i) const { return (&x)[i]; }
};
GLM_FN vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
GLM_FN vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
GLM_FN vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
GLM_FN vec3 operator*(float s, const vec3& a) { return vec3(s * a.x, s * a.y, s * a.z); }
GLM_FN vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
GLM_FN vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
GLM_FN vec3 operator/(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
GLM_FN float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GLM_FN float length(const vec3& a) { return sqrtf(dot(a, a)); }
GLM_FN vec3 max(const vec3& a, float s) { return vec3(fmaxf(a.x, s), fmaxf(a.y, s), fmaxf(a.z, s)); }
struct mat3 {
  vec3 c[3];  // columns
  GLM_FN mat3() { c[0] = vec3(1, 0, 0); c[1] = vec3(0, 1, 0); c[2] = vec3(0, 0, 1); }
  GLM_FN explicit mat3(float s) { c[0] = vec3(s, 0, 0); c[1] = vec3(0, s, 0); c[2] = vec3(0, 0, s); }
  GLM_FN mat3(float a, float b, float cc, float d, float e, float f, float g, float h, float i) {
    c[0] = vec3(a, b, cc); c[1] = vec3(d, e, f); c[2] = vec3(g, h, i);
  }
  GLM_FN vec3& operator[](int i) { return c[i]; }
  GLM_FN const vec3& operator[](int i) const { return c[i]; }
};
GLM_FN vec3 operator*(const mat3& m, const vec3& v) { return m[0] * v.x + m[1] * v.y + m[2] * v.z; }
GLM_FN mat3 operator*(const mat3& a, const mat3& b) {
  mat3 r(0.0f);
  r[0] = a * b[0]; r[1] = a * b[1]; r[2] = a * b[2];
  return r;
}
GLM_FN mat3 operator*(float s, const mat3& m) { mat3 r(0.0f); r[0] = s * m[0]; r[1] = s * m[1]; r[2] = s * m[2]; return r; }
GLM_FN mat3 operator*(const mat3& m, float s) { return s * m; }
GLM_FN mat3 transpose(const mat3& m) {
  return mat3(m[0].x, m[1].x, m[2].x, m[0].y, m[1].y, m[2].y, m[0].z, m[1].z, m[2].z);
}
}  // namespace glm
