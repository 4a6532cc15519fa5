// fe_math.h -- host-side transform math of the .pbrt front end, restating core/transform.rs in f32:
//   Matrix4x4::{mul :147-163, inverse :78-145, transpose}, Transform::{translate :255-270, scale :272-290,
//   rotate :329-355, look_at :357-393, perspective :399-411, Mul :653-660, swaps_handedness :638-645},
//   transform_point :413-432, transform_vector :496-504, transform_normal :529-541.
#pragma once
#include <cmath>
#include <cstring>

namespace fe {

struct Vec3 { float x = 0, y = 0, z = 0; };
inline Vec3 v3(float x, float y, float z) { Vec3 v; v.x = x; v.y = y; v.z = z; return v; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(Vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator/(Vec3 a, float s) { float inv = 1.0f / s; return v3(a.x * inv, a.y * inv, a.z * inv); }
inline float length(Vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline Vec3 normalize(Vec3 a) { return a / length(a); }
inline Vec3 cross(Vec3 a, Vec3 b) {  // vector.rs: evaluated in f64, as the reference does
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return v3((float)(ay * bz - az * by), (float)(az * bx - ax * bz), (float)(ax * by - ay * bx));
}

struct Mat4 {
    float m[4][4];
    Mat4() { std::memset(m, 0, sizeof m); for (int i = 0; i < 4; ++i) m[i][i] = 1.0f; }
};
inline Mat4 mat_mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
inline Mat4 mat_transpose(const Mat4 &a) { Mat4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i]; return r; }
inline Mat4 mat_inverse(const Mat4 &min) {  // Gauss-Jordan with full pivoting, transform.rs:78-145
    int indxc[4] = {0, 0, 0, 0}, indxr[4] = {0, 0, 0, 0}, ipiv[4] = {0, 0, 0, 0};
    Mat4 minv = min;
    for (int i = 0; i < 4; ++i) {
        int irow = 0, icol = 0; float big = 0.0f;
        for (int j = 0; j < 4; ++j)
            if (ipiv[j] != 1)
                for (int k = 0; k < 4; ++k)
                    if (ipiv[k] == 0) { float a = std::fabs(minv.m[j][k]); if (a >= big) { big = a; irow = j; icol = k; } }
        ipiv[icol] += 1;
        if (irow != icol) for (int k = 0; k < 4; ++k) { float t = minv.m[irow][k]; minv.m[irow][k] = minv.m[icol][k]; minv.m[icol][k] = t; }
        indxr[i] = irow; indxc[i] = icol;
        float pivinv = 1.0f / minv.m[icol][icol];
        minv.m[icol][icol] = 1.0f;
        for (int j = 0; j < 4; ++j) minv.m[icol][j] *= pivinv;
        for (int j = 0; j < 4; ++j)
            if (j != icol) { float save = minv.m[j][icol]; minv.m[j][icol] = 0.0f; for (int k = 0; k < 4; ++k) minv.m[j][k] -= minv.m[icol][k] * save; }
    }
    for (int i = 0; i < 4; ++i) {
        int j = 3 - i;
        if (indxr[j] != indxc[j]) for (int k = 0; k < 4; ++k) { float t = minv.m[k][indxr[j]]; minv.m[k][indxr[j]] = minv.m[k][indxc[j]]; minv.m[k][indxc[j]] = t; }
    }
    return minv;
}

struct Transform {
    Mat4 m, m_inv;
    Transform() {}
    Transform(const Mat4 &a, const Mat4 &ai) : m(a), m_inv(ai) {}
    explicit Transform(const Mat4 &a) : m(a), m_inv(mat_inverse(a)) {}
    Transform inverse() const { return Transform(m_inv, m); }
    Transform operator*(const Transform &o) const { return Transform(mat_mul(m, o.m), mat_mul(o.m_inv, m_inv)); }
    bool swaps_handedness() const {
        float det = m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1]) - m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0]) +
                    m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0]);
        return det < 0.0f;
    }
    Vec3 point(Vec3 p) const {
        float x = p.x, y = p.y, z = p.z;
        float xp = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z + m.m[0][3];
        float yp = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z + m.m[1][3];
        float zp = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z + m.m[2][3];
        float wp = m.m[3][0] * x + m.m[3][1] * y + m.m[3][2] * z + m.m[3][3];
        return wp == 1.0f ? v3(xp, yp, zp) : v3(xp, yp, zp) / wp;
    }
    Vec3 vector(Vec3 v) const {
        return v3(m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z, m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z,
                  m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z);
    }
    Vec3 normal(Vec3 n) const {
        return v3(m_inv.m[0][0] * n.x + m_inv.m[1][0] * n.y + m_inv.m[2][0] * n.z, m_inv.m[0][1] * n.x + m_inv.m[1][1] * n.y + m_inv.m[2][1] * n.z,
                  m_inv.m[0][2] * n.x + m_inv.m[1][2] * n.y + m_inv.m[2][2] * n.z);
    }
    void flat(float out[16], bool inv = false) const { const Mat4 &a = inv ? m_inv : m; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = a.m[i][j]; }

    static Transform translate(Vec3 d) {
        Mat4 a, ai; a.m[0][3] = d.x; a.m[1][3] = d.y; a.m[2][3] = d.z; ai.m[0][3] = -d.x; ai.m[1][3] = -d.y; ai.m[2][3] = -d.z;
        return Transform(a, ai);
    }
    static Transform scale(float x, float y, float z) {
        Mat4 a, ai; a.m[0][0] = x; a.m[1][1] = y; a.m[2][2] = z; ai.m[0][0] = 1.0f / x; ai.m[1][1] = 1.0f / y; ai.m[2][2] = 1.0f / z;
        return Transform(a, ai);
    }
    static Transform rotate(float theta, Vec3 axis) {  // transform.rs:329-355
        Vec3 a = normalize(axis);
        const float r = (3.14159265358979323846f / 180.0f) * theta;
        const float s = std::sin(r), c = std::cos(r);
        Mat4 mm;
        mm.m[0][0] = a.x * a.x + (1.0f - a.x * a.x) * c; mm.m[0][1] = a.x * a.y * (1.0f - c) - a.z * s; mm.m[0][2] = a.x * a.z * (1.0f - c) + a.y * s;
        mm.m[1][0] = a.x * a.y * (1.0f - c) + a.z * s; mm.m[1][1] = a.y * a.y + (1.0f - a.y * a.y) * c; mm.m[1][2] = a.y * a.z * (1.0f - c) - a.x * s;
        mm.m[2][0] = a.x * a.z * (1.0f - c) - a.y * s; mm.m[2][1] = a.y * a.z * (1.0f - c) + a.x * s; mm.m[2][2] = a.z * a.z + (1.0f - a.z * a.z) * c;
        return Transform(mm, mat_transpose(mm));
    }
    static Transform look_at(Vec3 pos, Vec3 look, Vec3 up) {  // transform.rs:357-393
        Mat4 c2w;
        c2w.m[0][3] = pos.x; c2w.m[1][3] = pos.y; c2w.m[2][3] = pos.z; c2w.m[3][3] = 1.0f;
        Vec3 dir = normalize(look - pos);
        Vec3 right = normalize(cross(normalize(up), dir));
        Vec3 new_up = cross(dir, right);
        c2w.m[0][0] = right.x; c2w.m[1][0] = right.y; c2w.m[2][0] = right.z; c2w.m[3][0] = 0.0f;
        c2w.m[0][1] = new_up.x; c2w.m[1][1] = new_up.y; c2w.m[2][1] = new_up.z; c2w.m[3][1] = 0.0f;
        c2w.m[0][2] = dir.x; c2w.m[1][2] = dir.y; c2w.m[2][2] = dir.z; c2w.m[3][2] = 0.0f;
        return Transform(mat_inverse(c2w), c2w);
    }
    static Transform perspective(float fov, float n, float f) {  // transform.rs:399-411
        Mat4 p;
        p.m[2][2] = f / (f - n); p.m[2][3] = -f * n / (f - n); p.m[3][2] = 1.0f; p.m[3][3] = 0.0f;
        const float inv_tan = 1.0f / std::tan((3.14159265358979323846f / 180.0f) * fov / 2.0f);
        return Transform::scale(inv_tan, inv_tan, 1.0f) * Transform(p);
    }
};

}  // namespace fe
