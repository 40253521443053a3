// =============================================================================
// raster_oracle.cpp -- CPU ORACLE for the 3D-Gaussian rasterizer hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
// product path (open-diffusiongs_amd/) never links, imports or falls back to it.
//
// It is a plain-C++ restatement (no CUDA, no glm, no torch) of the algorithm of
// the reference's diff-gaussian-rasterization submodule.  Every function cites
// the reference file:line it follows (paths relative to
// /root/reference/submodules/diff-gaussian-rasterization/).
//
// PARITY STATUS: PINNED to the reference's own code.  oracle/build_ref.py translates the reference's CUDA sources at
// build time (hipify-perl, into the git-ignored oracle/_ref/) and runs them on the MI355X; tests/test_raster_ref_gpu.py
// compares this restatement with them live, and tests/test_oracle_ref_golden.py (CPU) compares it with fixtures
// produced by that build (oracle/make_raster_ref_golden.py -> tests/golden/raster_ref_*.npz): radii, tiles_touched,
// num_rendered, ranges, per-tile sorted point_list, depths, means2D, conic/opacity, rgb and cov3D are BIT-identical
// (reference compiled with -ffp-contract=off); colour <= 1e-6 and n_contrib differs on <= 1 pixel of 65,536 (libm expf
// here vs ocml exp there); all 9 gradients <= 1e-5 relative.  The closed-form known-answer tests
// (tests/test_oracle_kat.py) and finite-difference gradient checks (tests/test_oracle_grad.py) stay as a second pin.
//
// Floating-point discipline (what "bit-exact vs the HIP path" means):
//   * compiled with -ffp-contract=off: a*b+c is two roundings unless written
//     as fmaf();  / and sqrtf are IEEE correctly rounded; same on the device.
//   * exp_mode 0: libm expf (the closest thing to CUDA's expf).
//     exp_mode 1: det_expf(), a fixed sequence of IEEE operations that the HIP
//     kernels reproduce bit for bit (rintf, fmaf chain, ldexpf).
//   * float->int conversion saturates (NaN -> 0) instead of being UB.
//   * glm's column-major mat3 product order (third_party/glm/glm/detail/
//     type_mat3x3.inl:486-519) is reproduced by struct M3 below.
// =============================================================================
#include <algorithm>
#include <omp.h>
#include <parallel/algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int TILE_X = 16;  // cuda_rasterizer/config.h:14-15
constexpr int TILE_Y = 16;
constexpr int TILE_PIX = TILE_X * TILE_Y;

// cuda_rasterizer/auxiliary.h:21-38
constexpr float kSH0 = 0.28209479177387814f;
constexpr float kSH1 = 0.4886025119029199f;
constexpr float kSH2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                           -1.0925484305920792f, 0.5462742152960396f};
constexpr float kSH3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                           0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                           -0.5900435899266435f};

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z
inline float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Column-major 3x3, c[col][row], with glm's product order (type_mat3x3.inl:486-519).
struct M3 {
    float c[3][3];
    static M3 cols(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7,
                   float a8) {
        M3 m;
        m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2;
        m.c[1][0] = a3; m.c[1][1] = a4; m.c[1][2] = a5;
        m.c[2][0] = a6; m.c[2][1] = a7; m.c[2][2] = a8;
        return m;
    }
};
inline M3 mul(const M3& A, const M3& B) {
    M3 R;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            R.c[j][i] = A.c[0][i] * B.c[j][0] + A.c[1][i] * B.c[j][1] + A.c[2][i] * B.c[j][2];
    return R;
}
inline M3 transpose(const M3& A) {
    M3 R;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) R.c[j][i] = A.c[i][j];
    return R;
}

inline int f2i_sat(float v) {  // C cast is UB out of range; the device saturates, so do we.
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

// Deterministic expf: one fixed sequence of IEEE-754 binary32 operations
// (Cody-Waite reduction by ln2 split hi/lo, degree-6 Cephes polynomial in an
// fmaf Horner chain, exact scaling by 2^n).  Max observed error < 1.5 ulp.
inline float det_expf(float x) {
    if (!(x == x)) return x;
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return INFINITY;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float y = fmaf(p, r2, r);
    y = y + 1.0f;
    return ldexpf(y, (int)n);
}

// auxiliary.h:41-44 -- evaluated in double, rounded to float on return.
inline float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// auxiliary.h:58-77, column-major flat 4x4 (m[4*col+row]).
inline V3 xform4x3(V3 p, const float* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
inline V4 xform4x4(V3 p, const float* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
            m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
// auxiliary.h:97-105
inline V3 xformVec4x3T(V3 p, const float* m) {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z};
}

// auxiliary.h:46-56
inline void tile_rect(float px, float py, int max_radius, int gx, int gy, int* x0, int* y0, int* x1,
                      int* y1) {
    const float r = (float)max_radius;
    *x0 = std::min(gx, std::max(0, f2i_sat((px - r) / (float)TILE_X)));
    *y0 = std::min(gy, std::max(0, f2i_sat((py - r) / (float)TILE_Y)));
    *x1 = std::min(gx, std::max(0, f2i_sat((px + r + (float)(TILE_X - 1)) / (float)TILE_X)));
    *y1 = std::min(gy, std::max(0, f2i_sat((py + r + (float)(TILE_Y - 1)) / (float)TILE_Y)));
}

// forward.cu:118-152 (quaternion (r,x,y,z) is NOT renormalised, :127).
inline void cov3d_from_scale_rot(V3 s, float mod, V4 q, float* out6) {
    M3 S = M3::cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.c[0][0] = mod * s.x; S.c[1][1] = mod * s.y; S.c[2][2] = mod * s.z;
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    const M3 R = M3::cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                          2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                          2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    const M3 Mm = mul(S, R);
    const M3 Sig = mul(transpose(Mm), Mm);
    out6[0] = Sig.c[0][0]; out6[1] = Sig.c[0][1]; out6[2] = Sig.c[0][2];
    out6[3] = Sig.c[1][1]; out6[4] = Sig.c[1][2]; out6[5] = Sig.c[2][2];
}

struct Cov2DInter {  // shared by forward.cu:74-113 and backward.cu:165-198
    V3 t;            // clamped camera-space mean
    float txtz, tytz, limx, limy;
    M3 J, Wm, Vrk, T, cov;
};
inline Cov2DInter cov2d_common(V3 mean, float fx, float fy, float tanx, float tany, const float* c6,
                               const float* vm) {
    Cov2DInter o;
    V3 t = xform4x3(mean, vm);
    o.limx = 1.3f * tanx; o.limy = 1.3f * tany;
    o.txtz = t.x / t.z; o.tytz = t.y / t.z;
    t.x = fminf(o.limx, fmaxf(-o.limx, o.txtz)) * t.z;
    t.y = fminf(o.limy, fmaxf(-o.limy, o.tytz)) * t.z;
    o.t = t;
    o.J = M3::cols(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z), 0.0f, fy / t.z,
                   -(fy * t.y) / (t.z * t.z), 0, 0, 0);
    o.Wm = M3::cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    o.T = mul(o.Wm, o.J);
    o.Vrk = M3::cols(c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]);
    o.cov = mul(mul(transpose(o.T), transpose(o.Vrk)), o.T);
    return o;
}

// forward.cu:20-71.  Returns clamped-to->=0 RGB; clamped[3] receives (value < 0).
inline V3 sh_to_rgb(int deg, int M, V3 pos, V3 cam, const float* sh_base, uint8_t* clamped) {
    V3 d = pos - cam;
    d = d / sqrtf(dot3(d, d));
    const V3* sh = reinterpret_cast<const V3*>(sh_base);
    (void)M;
    V3 res = kSH0 * sh[0];
    if (deg > 0) {
        const float x = d.x, y = d.y, z = d.z;
        res = res - kSH1 * y * sh[1] + kSH1 * z * sh[2] - kSH1 * x * sh[3];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            res = res + kSH2[0] * xy * sh[4] + kSH2[1] * yz * sh[5] +
                  kSH2[2] * (2.0f * zz - xx - yy) * sh[6] + kSH2[3] * xz * sh[7] +
                  kSH2[4] * (xx - yy) * sh[8];
            if (deg > 2) {
                res = res + kSH3[0] * y * (3.0f * xx - yy) * sh[9] + kSH3[1] * xy * z * sh[10] +
                      kSH3[2] * y * (4.0f * zz - xx - yy) * sh[11] +
                      kSH3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12] +
                      kSH3[4] * x * (4.0f * zz - xx - yy) * sh[13] + kSH3[5] * z * (xx - yy) * sh[14] +
                      kSH3[6] * x * (xx - 3.0f * yy) * sh[15];
            }
        }
    }
    res.x += 0.5f; res.y += 0.5f; res.z += 0.5f;
    clamped[0] = res.x < 0; clamped[1] = res.y < 0; clamped[2] = res.z < 0;
    return {fmaxf(res.x, 0.0f), fmaxf(res.y, 0.0f), fmaxf(res.z, 0.0f)};
}

struct Oracle {
    // configuration of the last forward
    int P = 0, D = 0, M = 0, W = 0, H = 0, gx = 0, gy = 0;
    int exp_mode = 0;
    float tanx = 0, tany = 0, fx = 0, fy = 0, scale_mod = 1.f;
    bool has_sh = false, has_cov_pre = false, has_col_pre = false;
    std::vector<float> means, shs, colors_pre, opac, scales, rots, cov_pre, vm, pm, cam, bg;
    // GeometryState (rasterizer_impl.h:29-45)
    std::vector<float> depths, means2D, cov3D, conic_opacity, rgb;
    std::vector<uint8_t> clamped;
    std::vector<int32_t> radii;
    std::vector<uint32_t> tiles_touched, point_offsets;
    // BinningState (rasterizer_impl.h:55-65)
    std::vector<uint64_t> keys;
    std::vector<uint32_t> point_list;
    // ImageState (rasterizer_impl.h:47-53)
    std::vector<uint32_t> ranges, n_contrib;
    std::vector<float> final_T, out_color;
    // gradients of the last backward
    std::vector<float> g_mean2D, g_conic, g_opacity, g_color, g_mean3D, g_cov3D, g_sh, g_scale, g_rot;
    int num_rendered = 0;

    float ex(float x) const { return exp_mode ? det_expf(x) : expf(x); }
};

template <class T>
void put(std::vector<T>& v, const T* p, size_t n) {
    if (p) v.assign(p, p + n); else v.clear();
}

// ----------------------------------------------------------------------------
// Forward.  rasterizer_impl.cu:198-336 (Rasterizer::forward) step by step.
// ----------------------------------------------------------------------------
int forward(Oracle& o) {
    const int P = o.P, W = o.W, H = o.H;
    o.gx = (W + TILE_X - 1) / TILE_X;
    o.gy = (H + TILE_Y - 1) / TILE_Y;
    const int T = o.gx * o.gy;
    // rasterizer_impl.cu:222-223
    o.fy = H / (2.0f * o.tany);
    o.fx = W / (2.0f * o.tanx);

    o.depths.assign(P, 0.f); o.means2D.assign(2 * (size_t)P, 0.f); o.cov3D.assign(6 * (size_t)P, 0.f);
    o.conic_opacity.assign(4 * (size_t)P, 0.f); o.rgb.assign(3 * (size_t)P, 0.f);
    o.clamped.assign(3 * (size_t)P, 0); o.radii.assign(P, 0); o.tiles_touched.assign(P, 0);
    o.point_offsets.assign(P, 0);
    o.ranges.assign(2 * (size_t)T, 0); o.n_contrib.assign((size_t)W * H, 0);
    o.final_T.assign((size_t)W * H, 0.f); o.out_color.assign(3 * (size_t)W * H, 0.f);
    o.keys.clear(); o.point_list.clear(); o.num_rendered = 0;
    if (P == 0) return 0;  // rasterize_points.cu:68-113: outputs stay zero

    const float* vm = o.vm.data();
    const float* pm = o.pm.data();
    // ---- preprocessCUDA, forward.cu:155-256 ----  (one Gaussian per iteration, no shared state: OpenMP over Gaussians)
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        const V3 p = {o.means[3 * i], o.means[3 * i + 1], o.means[3 * i + 2]};
        // in_frustum, auxiliary.h:139-164 (only the near-plane test is live)
        const V4 ph = xform4x4(p, pm);
        const float pw = 1.0f / (ph.w + 0.0000001f);
        const V3 pp = {ph.x * pw, ph.y * pw, ph.z * pw};
        const V3 pv = xform4x3(p, vm);
        if (pv.z <= 0.2f) continue;
        const float* c6;
        if (o.has_cov_pre) {
            c6 = &o.cov_pre[6 * (size_t)i];
        } else {
            cov3d_from_scale_rot({o.scales[3 * i], o.scales[3 * i + 1], o.scales[3 * i + 2]},
                                 o.scale_mod,
                                 {o.rots[4 * i], o.rots[4 * i + 1], o.rots[4 * i + 2], o.rots[4 * i + 3]},
                                 &o.cov3D[6 * (size_t)i]);
            c6 = &o.cov3D[6 * (size_t)i];
        }
        // computeCov2D, forward.cu:74-113
        const Cov2DInter ci = cov2d_common(p, o.fx, o.fy, o.tanx, o.tany, c6, vm);
        const float ca = ci.cov.c[0][0] + 0.3f, cb = ci.cov.c[0][1], cc = ci.cov.c[1][1] + 0.3f;
        // forward.cu:215-236
        const float det = ca * cc - cb * cb;
        if (det == 0.0f) continue;
        const float det_inv = 1.f / det;
        const float conx = cc * det_inv, cony = -cb * det_inv, conz = ca * det_inv;
        const float mid = 0.5f * (ca + cc);
        const float l1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float l2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        const float px = ndc2pix(pp.x, W), py = ndc2pix(pp.y, H);
        int x0, y0, x1, y1;
        tile_rect(px, py, f2i_sat(my_radius), o.gx, o.gy, &x0, &y0, &x1, &y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;
        // forward.cu:238-246
        if (!o.has_col_pre) {
            const V3 c = sh_to_rgb(o.D, o.M, p, {o.cam[0], o.cam[1], o.cam[2]},
                                   &o.shs[3 * (size_t)o.M * i], &o.clamped[3 * (size_t)i]);
            o.rgb[3 * i] = c.x; o.rgb[3 * i + 1] = c.y; o.rgb[3 * i + 2] = c.z;
        }
        // forward.cu:248-255
        o.depths[i] = pv.z;
        o.radii[i] = f2i_sat(my_radius);
        o.means2D[2 * i] = px; o.means2D[2 * i + 1] = py;
        o.conic_opacity[4 * i] = conx; o.conic_opacity[4 * i + 1] = cony;
        o.conic_opacity[4 * i + 2] = conz; o.conic_opacity[4 * i + 3] = o.opac[i];
        o.tiles_touched[i] = (uint32_t)((y1 - y0) * (x1 - x0));
    }
    // ---- InclusiveSum, rasterizer_impl.cu:277 ----
    uint32_t run = 0;
    for (int i = 0; i < P; ++i) { run += o.tiles_touched[i]; o.point_offsets[i] = run; }
    const int N = (int)run;  // rasterizer_impl.cu:281
    o.num_rendered = N;
    // ---- duplicateWithKeys, rasterizer_impl.cu:70-111 ----
    std::vector<uint64_t> keys_uns(N);
    std::vector<uint32_t> vals_uns(N);
    for (int i = 0; i < P; ++i) {
        if (o.radii[i] <= 0) continue;
        uint32_t off = (i == 0) ? 0 : o.point_offsets[i - 1];
        int x0, y0, x1, y1;
        tile_rect(o.means2D[2 * i], o.means2D[2 * i + 1], o.radii[i], o.gx, o.gy, &x0, &y0, &x1, &y1);
        uint32_t dbits;
        std::memcpy(&dbits, &o.depths[i], 4);
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                uint64_t key = (uint64_t)(y * o.gx + x);
                key <<= 32;
                key |= dbits;
                keys_uns[off] = key; vals_uns[off] = (uint32_t)i; ++off;
            }
    }
    // ---- cub::DeviceRadixSort::SortPairs over bits [0, 32+msb), rasterizer_impl.cu:300-308.
    // LSD radix sort is stable; tile ids are < 2^msb, so this equals a stable sort on the
    // full 64-bit key: order = (tile, depth bits, emission order == Gaussian index).
    std::vector<uint32_t> perm(N);
    for (int k = 0; k < N; ++k) perm[k] = (uint32_t)k;
    __gnu_parallel::stable_sort(perm.begin(), perm.end(),
                                [&](uint32_t a, uint32_t b) { return keys_uns[a] < keys_uns[b]; });
    o.keys.resize(N); o.point_list.resize(N);
    for (int k = 0; k < N; ++k) { o.keys[k] = keys_uns[perm[k]]; o.point_list[k] = vals_uns[perm[k]]; }
    // ---- identifyTileRanges, rasterizer_impl.cu:116-138 (+ memset :310) ----
    for (int k = 0; k < N; ++k) {
        const uint32_t cur = (uint32_t)(o.keys[k] >> 32);
        if (k == 0) o.ranges[2 * cur] = 0;
        else {
            const uint32_t prev = (uint32_t)(o.keys[k - 1] >> 32);
            if (cur != prev) { o.ranges[2 * prev + 1] = (uint32_t)k; o.ranges[2 * cur] = (uint32_t)k; }
        }
        if (k == N - 1) o.ranges[2 * cur + 1] = (uint32_t)N;
    }
    // ---- renderCUDA, forward.cu:261-374 ----
    const float* feat = o.has_col_pre ? o.colors_pre.data() : o.rgb.data();
    // tiles are independent ("OpenMP-over-tiles" CPU baseline); results do not depend on the thread count
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int ty = 0; ty < o.gy; ++ty)
        for (int tx = 0; tx < o.gx; ++tx) {
            const uint32_t r0 = o.ranges[2 * (ty * o.gx + tx)], r1 = o.ranges[2 * (ty * o.gx + tx) + 1];
            for (int ly = 0; ly < TILE_Y; ++ly)
                for (int lx = 0; lx < TILE_X; ++lx) {
                    const int pxi = tx * TILE_X + lx, pyi = ty * TILE_Y + ly;
                    if (!(pxi < W && pyi < H)) continue;
                    const float pfx = (float)pxi, pfy = (float)pyi;
                    float Tr = 1.0f, C[3] = {0, 0, 0};
                    uint32_t contributor = 0, last = 0;
                    for (uint32_t k = r0; k < r1; ++k) {
                        ++contributor;
                        const uint32_t g = o.point_list[k];
                        const float dx = o.means2D[2 * g] - pfx, dy = o.means2D[2 * g + 1] - pfy;
                        const float* co = &o.conic_opacity[4 * (size_t)g];
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float alpha = fminf(0.99f, co[3] * o.ex(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        const float test_T = Tr * (1 - alpha);
                        if (test_T < 0.0001f) break;  // done = true (forward.cu:346-350)
                        for (int ch = 0; ch < 3; ++ch) C[ch] += feat[3 * (size_t)g + ch] * alpha * Tr;
                        Tr = test_T;
                        last = contributor;
                    }
                    const size_t pid = (size_t)W * pyi + pxi;
                    o.final_T[pid] = Tr; o.n_contrib[pid] = last;
                    for (int ch = 0; ch < 3; ++ch)
                        o.out_color[(size_t)ch * H * W + pid] = C[ch] + Tr * o.bg[ch];
                }
        }
    return N;
}

// auxiliary.h:107-117
inline V3 dnormvdv3(V3 v, V3 dv) {
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float inv = 1.0f / sqrtf(sum2 * sum2 * sum2);
    V3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * inv;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * inv;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * inv;
    return r;
}

// backward.cu:20-139: SH backward.  Adds the direction term into g_mean[idx].
void sh_backward(const Oracle& o, int idx, const float* dL_dcolor, float* g_mean, float* g_sh) {
    const V3 pos = {o.means[3 * idx], o.means[3 * idx + 1], o.means[3 * idx + 2]};
    const V3 cam = {o.cam[0], o.cam[1], o.cam[2]};
    const V3 dir_orig = pos - cam;
    const V3 dir = dir_orig / sqrtf(dot3(dir_orig, dir_orig));
    const V3* sh = reinterpret_cast<const V3*>(&o.shs[3 * (size_t)o.M * idx]);
    V3 dRGB = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
    dRGB.x *= o.clamped[3 * idx + 0] ? 0 : 1;
    dRGB.y *= o.clamped[3 * idx + 1] ? 0 : 1;
    dRGB.z *= o.clamped[3 * idx + 2] ? 0 : 1;
    V3 dx = {0, 0, 0}, dy = {0, 0, 0}, dz = {0, 0, 0};
    const float x = dir.x, y = dir.y, z = dir.z;
    V3* out = reinterpret_cast<V3*>(&g_sh[3 * (size_t)o.M * idx]);
    out[0] = kSH0 * dRGB;
    if (o.D > 0) {
        out[1] = (-kSH1 * y) * dRGB;
        out[2] = (kSH1 * z) * dRGB;
        out[3] = (-kSH1 * x) * dRGB;
        dx = -kSH1 * sh[3]; dy = -kSH1 * sh[1]; dz = kSH1 * sh[2];
        if (o.D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            out[4] = (kSH2[0] * xy) * dRGB;
            out[5] = (kSH2[1] * yz) * dRGB;
            out[6] = (kSH2[2] * (2.f * zz - xx - yy)) * dRGB;
            out[7] = (kSH2[3] * xz) * dRGB;
            out[8] = (kSH2[4] * (xx - yy)) * dRGB;
            dx = dx + (kSH2[0] * y * sh[4] + kSH2[2] * 2.f * -x * sh[6] + kSH2[3] * z * sh[7] +
                       kSH2[4] * 2.f * x * sh[8]);
            dy = dy + (kSH2[0] * x * sh[4] + kSH2[1] * z * sh[5] + kSH2[2] * 2.f * -y * sh[6] +
                       kSH2[4] * 2.f * -y * sh[8]);
            dz = dz + (kSH2[1] * y * sh[5] + kSH2[2] * 2.f * 2.f * z * sh[6] + kSH2[3] * x * sh[7]);
            if (o.D > 2) {
                out[9] = (kSH3[0] * y * (3.f * xx - yy)) * dRGB;
                out[10] = (kSH3[1] * xy * z) * dRGB;
                out[11] = (kSH3[2] * y * (4.f * zz - xx - yy)) * dRGB;
                out[12] = (kSH3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dRGB;
                out[13] = (kSH3[4] * x * (4.f * zz - xx - yy)) * dRGB;
                out[14] = (kSH3[5] * z * (xx - yy)) * dRGB;
                out[15] = (kSH3[6] * x * (xx - 3.f * yy)) * dRGB;
                dx = dx + (kSH3[0] * sh[9] * 3.f * 2.f * xy + kSH3[1] * sh[10] * yz +
                           kSH3[2] * sh[11] * -2.f * xy + kSH3[3] * sh[12] * -3.f * 2.f * xz +
                           kSH3[4] * sh[13] * (-3.f * xx + 4.f * zz - yy) + kSH3[5] * sh[14] * 2.f * xz +
                           kSH3[6] * sh[15] * 3.f * (xx - yy));
                dy = dy + (kSH3[0] * sh[9] * 3.f * (xx - yy) + kSH3[1] * sh[10] * xz +
                           kSH3[2] * sh[11] * (-3.f * yy + 4.f * zz - xx) +
                           kSH3[3] * sh[12] * -3.f * 2.f * yz + kSH3[4] * sh[13] * -2.f * xy +
                           kSH3[5] * sh[14] * -2.f * yz + kSH3[6] * sh[15] * -3.f * 2.f * xy);
                dz = dz + (kSH3[1] * sh[10] * xy + kSH3[2] * sh[11] * 4.f * 2.f * yz +
                           kSH3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) +
                           kSH3[4] * sh[13] * 4.f * 2.f * xz + kSH3[5] * sh[14] * (xx - yy));
            }
        }
    }
    const V3 ddir = {dot3(dx, dRGB), dot3(dy, dRGB), dot3(dz, dRGB)};
    const V3 dm = dnormvdv3(dir_orig, ddir);
    g_mean[3 * idx] += dm.x; g_mean[3 * idx + 1] += dm.y; g_mean[3 * idx + 2] += dm.z;
}

// ----------------------------------------------------------------------------
// Backward.  rasterizer_impl.cu:340-434 (Rasterizer::backward).
// accum64 != 0 accumulates the per-Gaussian sums of the blend backward in double
// (the reference sums in float through atomics in an unspecified order; double gives
// the order-free value that float-atomic implementations scatter around).
// ----------------------------------------------------------------------------
void backward(Oracle& o, const float* dL_dpix, int accum64) {
    const int P = o.P, W = o.W, H = o.H;
    const size_t Ps = (size_t)P;
    o.g_mean2D.assign(3 * Ps, 0.f); o.g_conic.assign(4 * Ps, 0.f); o.g_opacity.assign(Ps, 0.f);
    o.g_color.assign(3 * Ps, 0.f); o.g_mean3D.assign(3 * Ps, 0.f); o.g_cov3D.assign(6 * Ps, 0.f);
    o.g_sh.assign(3 * Ps * (size_t)o.M, 0.f); o.g_scale.assign(3 * Ps, 0.f); o.g_rot.assign(4 * Ps, 0.f);
    if (P == 0) return;
    std::vector<double> acc;  // 9 per Gaussian: color3, mean2D.xy, conic.x/.y/.w, opacity
    if (accum64) acc.assign(9 * Ps, 0.0);
    const float* feat = o.has_col_pre ? o.colors_pre.data() : o.rgb.data();
    auto add = [&](size_t g, int slot, float* dst, float v) {
        if (accum64) acc[9 * g + slot] += (double)v; else *dst += v;
    };
    // ---- renderCUDA backward, backward.cu:399-557 ----
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
    for (int ty = 0; ty < o.gy; ++ty)
        for (int tx = 0; tx < o.gx; ++tx) {
            const uint32_t r0 = o.ranges[2 * (ty * o.gx + tx)], r1 = o.ranges[2 * (ty * o.gx + tx) + 1];
            const int todo = (int)(r1 - r0);
            struct Pix { bool in; float T, Tfin, acc_rec[3], dpix[3], last_alpha, last_col[3]; uint32_t contributor; int last_contrib; float fx, fy; };
            Pix px[TILE_PIX];
            for (int t = 0; t < TILE_PIX; ++t) {
                const int pxi = tx * TILE_X + (t % TILE_X), pyi = ty * TILE_Y + (t / TILE_X);
                Pix& q = px[t];
                q.in = pxi < W && pyi < H;
                const size_t pid = (size_t)W * pyi + pxi;
                q.Tfin = q.in ? o.final_T[pid] : 0; q.T = q.Tfin;
                q.contributor = (uint32_t)todo;
                q.last_contrib = q.in ? (int)o.n_contrib[pid] : 0;
                for (int ch = 0; ch < 3; ++ch) {
                    q.acc_rec[ch] = 0; q.last_col[ch] = 0;
                    q.dpix[ch] = q.in ? dL_dpix[(size_t)ch * H * W + pid] : 0.f;
                }
                q.last_alpha = 0; q.fx = (float)pxi; q.fy = (float)pyi;
            }
            for (int j = 0; j < todo; ++j) {  // back to front, all 256 "threads" per Gaussian
                const uint32_t g = o.point_list[r1 - 1 - j];
                const float* co = &o.conic_opacity[4 * (size_t)g];
                for (int t = 0; t < TILE_PIX; ++t) {
                    Pix& q = px[t];
                    if (!q.in) continue;
                    q.contributor--;
                    if ((int)q.contributor >= q.last_contrib) continue;
                    const float dx = o.means2D[2 * g] - q.fx, dy = o.means2D[2 * g + 1] - q.fy;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float G = o.ex(power);
                    const float alpha = fminf(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    q.T = q.T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * q.T;
                    float dL_dalpha = 0.0f;
                    for (int ch = 0; ch < 3; ++ch) {
                        const float c = feat[3 * (size_t)g + ch];
                        q.acc_rec[ch] = q.last_alpha * q.last_col[ch] + (1.f - q.last_alpha) * q.acc_rec[ch];
                        q.last_col[ch] = c;
                        const float dch = q.dpix[ch];
                        dL_dalpha += (c - q.acc_rec[ch]) * dch;
                        add(g, ch, &o.g_color[3 * (size_t)g + ch], dchannel_dcolor * dch);
                    }
                    dL_dalpha *= q.T;
                    q.last_alpha = alpha;
                    float bg_dot = 0;
                    for (int ch = 0; ch < 3; ++ch) bg_dot += o.bg[ch] * q.dpix[ch];
                    dL_dalpha += (-q.Tfin / (1.f - alpha)) * bg_dot;
                    const float dL_dG = co[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    add(g, 3, &o.g_mean2D[3 * (size_t)g + 0], dL_dG * dG_ddelx * ddelx_dx);
                    add(g, 4, &o.g_mean2D[3 * (size_t)g + 1], dL_dG * dG_ddely * ddely_dy);
                    add(g, 5, &o.g_conic[4 * (size_t)g + 0], -0.5f * gdx * dx * dL_dG);
                    add(g, 6, &o.g_conic[4 * (size_t)g + 1], -0.5f * gdx * dy * dL_dG);
                    add(g, 7, &o.g_conic[4 * (size_t)g + 3], -0.5f * gdy * dy * dL_dG);
                    add(g, 8, &o.g_opacity[g], G * dL_dalpha);
                }
            }
        }
    if (accum64)
        for (size_t g = 0; g < Ps; ++g) {
            for (int ch = 0; ch < 3; ++ch) o.g_color[3 * g + ch] = (float)acc[9 * g + ch];
            o.g_mean2D[3 * g + 0] = (float)acc[9 * g + 3]; o.g_mean2D[3 * g + 1] = (float)acc[9 * g + 4];
            o.g_conic[4 * g + 0] = (float)acc[9 * g + 5]; o.g_conic[4 * g + 1] = (float)acc[9 * g + 6];
            o.g_conic[4 * g + 3] = (float)acc[9 * g + 7]; o.g_opacity[g] = (float)acc[9 * g + 8];
        }

    const float* vm = o.vm.data();
    const float* proj = o.pm.data();
    for (int i = 0; i < P; ++i) {
        if (!(o.radii[i] > 0)) continue;
        const V3 mean = {o.means[3 * i], o.means[3 * i + 1], o.means[3 * i + 2]};
        // ---- computeCov2DCUDA, backward.cu:144-274 ----
        const float* c6 = o.has_cov_pre ? &o.cov_pre[6 * (size_t)i] : &o.cov3D[6 * (size_t)i];
        const float dcx = o.g_conic[4 * i], dcy = o.g_conic[4 * i + 1], dcz = o.g_conic[4 * i + 3];
        const Cov2DInter ci = cov2d_common(mean, o.fx, o.fy, o.tanx, o.tany, c6, vm);
        const float xgm = (ci.txtz < -ci.limx || ci.txtz > ci.limx) ? 0.f : 1.f;
        const float ygm = (ci.tytz < -ci.limy || ci.tytz > ci.limy) ? 0.f : 1.f;
        const M3& T = ci.T;
        const M3& V = ci.Vrk;
        const float a = ci.cov.c[0][0] + 0.3f, b = ci.cov.c[0][1], c = ci.cov.c[1][1] + 0.3f;
        const float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dcov = &o.g_cov3D[6 * (size_t)i];
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz);
            dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx);
            dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
            dcov[0] = (T.c[0][0] * T.c[0][0] * dL_da + T.c[0][0] * T.c[1][0] * dL_db + T.c[1][0] * T.c[1][0] * dL_dc);
            dcov[3] = (T.c[0][1] * T.c[0][1] * dL_da + T.c[0][1] * T.c[1][1] * dL_db + T.c[1][1] * T.c[1][1] * dL_dc);
            dcov[5] = (T.c[0][2] * T.c[0][2] * dL_da + T.c[0][2] * T.c[1][2] * dL_db + T.c[1][2] * T.c[1][2] * dL_dc);
            dcov[1] = 2 * T.c[0][0] * T.c[0][1] * dL_da + (T.c[0][0] * T.c[1][1] + T.c[0][1] * T.c[1][0]) * dL_db + 2 * T.c[1][0] * T.c[1][1] * dL_dc;
            dcov[2] = 2 * T.c[0][0] * T.c[0][2] * dL_da + (T.c[0][0] * T.c[1][2] + T.c[0][2] * T.c[1][0]) * dL_db + 2 * T.c[1][0] * T.c[1][2] * dL_dc;
            dcov[4] = 2 * T.c[0][2] * T.c[0][1] * dL_da + (T.c[0][1] * T.c[1][2] + T.c[0][2] * T.c[1][1]) * dL_db + 2 * T.c[1][1] * T.c[1][2] * dL_dc;
        } else {
            for (int k = 0; k < 6; ++k) dcov[k] = 0;
        }
        const float dT00 = 2 * (T.c[0][0] * V.c[0][0] + T.c[0][1] * V.c[0][1] + T.c[0][2] * V.c[0][2]) * dL_da +
                           (T.c[1][0] * V.c[0][0] + T.c[1][1] * V.c[0][1] + T.c[1][2] * V.c[0][2]) * dL_db;
        const float dT01 = 2 * (T.c[0][0] * V.c[1][0] + T.c[0][1] * V.c[1][1] + T.c[0][2] * V.c[1][2]) * dL_da +
                           (T.c[1][0] * V.c[1][0] + T.c[1][1] * V.c[1][1] + T.c[1][2] * V.c[1][2]) * dL_db;
        const float dT02 = 2 * (T.c[0][0] * V.c[2][0] + T.c[0][1] * V.c[2][1] + T.c[0][2] * V.c[2][2]) * dL_da +
                           (T.c[1][0] * V.c[2][0] + T.c[1][1] * V.c[2][1] + T.c[1][2] * V.c[2][2]) * dL_db;
        const float dT10 = 2 * (T.c[1][0] * V.c[0][0] + T.c[1][1] * V.c[0][1] + T.c[1][2] * V.c[0][2]) * dL_dc +
                           (T.c[0][0] * V.c[0][0] + T.c[0][1] * V.c[0][1] + T.c[0][2] * V.c[0][2]) * dL_db;
        const float dT11 = 2 * (T.c[1][0] * V.c[1][0] + T.c[1][1] * V.c[1][1] + T.c[1][2] * V.c[1][2]) * dL_dc +
                           (T.c[0][0] * V.c[1][0] + T.c[0][1] * V.c[1][1] + T.c[0][2] * V.c[1][2]) * dL_db;
        const float dT12 = 2 * (T.c[1][0] * V.c[2][0] + T.c[1][1] * V.c[2][1] + T.c[1][2] * V.c[2][2]) * dL_dc +
                           (T.c[0][0] * V.c[2][0] + T.c[0][1] * V.c[2][1] + T.c[0][2] * V.c[2][2]) * dL_db;
        const M3& Wm = ci.Wm;
        const float dJ00 = Wm.c[0][0] * dT00 + Wm.c[0][1] * dT01 + Wm.c[0][2] * dT02;
        const float dJ02 = Wm.c[2][0] * dT00 + Wm.c[2][1] * dT01 + Wm.c[2][2] * dT02;
        const float dJ11 = Wm.c[1][0] * dT10 + Wm.c[1][1] * dT11 + Wm.c[1][2] * dT12;
        const float dJ12 = Wm.c[2][0] * dT10 + Wm.c[2][1] * dT11 + Wm.c[2][2] * dT12;
        const float tz = 1.f / ci.t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        const float h_x = o.fx, h_y = o.fy;
        const float dtx = xgm * -h_x * tz2 * dJ02;
        const float dty = ygm * -h_y * tz2 * dJ12;
        const float dtz = -h_x * tz2 * dJ00 - h_y * tz2 * dJ11 + (2 * h_x * ci.t.x) * tz3 * dJ02 +
                          (2 * h_y * ci.t.y) * tz3 * dJ12;
        const V3 dmean_cov = xformVec4x3T({dtx, dty, dtz}, vm);
        o.g_mean3D[3 * i] = dmean_cov.x; o.g_mean3D[3 * i + 1] = dmean_cov.y; o.g_mean3D[3 * i + 2] = dmean_cov.z;

        // ---- preprocessCUDA backward, backward.cu:346-396 ----
        const V4 mh = xform4x4(mean, proj);
        const float m_w = 1.0f / (mh.w + 0.0000001f);
        const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        const float d2x = o.g_mean2D[3 * i], d2y = o.g_mean2D[3 * i + 1];
        V3 dm;
        dm.x = (proj[0] * m_w - proj[3] * mul1) * d2x + (proj[1] * m_w - proj[3] * mul2) * d2y;
        dm.y = (proj[4] * m_w - proj[7] * mul1) * d2x + (proj[5] * m_w - proj[7] * mul2) * d2y;
        dm.z = (proj[8] * m_w - proj[11] * mul1) * d2x + (proj[9] * m_w - proj[11] * mul2) * d2y;
        o.g_mean3D[3 * i] += dm.x; o.g_mean3D[3 * i + 1] += dm.y; o.g_mean3D[3 * i + 2] += dm.z;
        if (o.has_sh) sh_backward(o, i, o.g_color.data(), o.g_mean3D.data(), o.g_sh.data());
        if (!o.has_cov_pre) {
            // computeCov3D backward, backward.cu:278-341
            const V4 q = {o.rots[4 * i], o.rots[4 * i + 1], o.rots[4 * i + 2], o.rots[4 * i + 3]};
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            const M3 R = M3::cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                                  2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                                  2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
            M3 S = M3::cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
            const V3 s = {o.scale_mod * o.scales[3 * i], o.scale_mod * o.scales[3 * i + 1], o.scale_mod * o.scales[3 * i + 2]};
            S.c[0][0] = s.x; S.c[1][1] = s.y; S.c[2][2] = s.z;
            const M3 Mm = mul(S, R);
            const float* d6 = dcov;
            const M3 dSig = M3::cols(d6[0], 0.5f * d6[1], 0.5f * d6[2], 0.5f * d6[1], d6[3], 0.5f * d6[4],
                                     0.5f * d6[2], 0.5f * d6[4], d6[5]);
            // dL_dM = 2.0f * M * dL_dSigma : (scalar * M) first (type_mat3x3.inl:459-465), then product
            M3 M2;
            for (int cc2 = 0; cc2 < 3; ++cc2) for (int rr = 0; rr < 3; ++rr) M2.c[cc2][rr] = Mm.c[cc2][rr] * 2.0f;
            const M3 dM = mul(M2, dSig);
            const M3 Rt = transpose(R);
            M3 dMt = transpose(dM);
            float* gs = &o.g_scale[3 * (size_t)i];
            gs[0] = Rt.c[0][0] * dMt.c[0][0] + Rt.c[0][1] * dMt.c[0][1] + Rt.c[0][2] * dMt.c[0][2];
            gs[1] = Rt.c[1][0] * dMt.c[1][0] + Rt.c[1][1] * dMt.c[1][1] + Rt.c[1][2] * dMt.c[1][2];
            gs[2] = Rt.c[2][0] * dMt.c[2][0] + Rt.c[2][1] * dMt.c[2][1] + Rt.c[2][2] * dMt.c[2][2];
            for (int k = 0; k < 3; ++k) { dMt.c[0][k] *= s.x; dMt.c[1][k] *= s.y; dMt.c[2][k] *= s.z; }
            float* gq = &o.g_rot[4 * (size_t)i];
            gq[0] = 2 * z * (dMt.c[0][1] - dMt.c[1][0]) + 2 * y * (dMt.c[2][0] - dMt.c[0][2]) + 2 * x * (dMt.c[1][2] - dMt.c[2][1]);
            gq[1] = 2 * y * (dMt.c[1][0] + dMt.c[0][1]) + 2 * z * (dMt.c[2][0] + dMt.c[0][2]) + 2 * r * (dMt.c[1][2] - dMt.c[2][1]) - 4 * x * (dMt.c[2][2] + dMt.c[1][1]);
            gq[2] = 2 * x * (dMt.c[1][0] + dMt.c[0][1]) + 2 * r * (dMt.c[2][0] - dMt.c[0][2]) + 2 * z * (dMt.c[1][2] + dMt.c[2][1]) - 4 * y * (dMt.c[2][2] + dMt.c[0][0]);
            gq[3] = 2 * r * (dMt.c[0][1] - dMt.c[1][0]) + 2 * x * (dMt.c[2][0] + dMt.c[0][2]) + 2 * y * (dMt.c[1][2] + dMt.c[2][1]) - 4 * z * (dMt.c[1][1] + dMt.c[0][0]);
        }
    }
}

}  // namespace

// -----------------------------------------------------------------------------
// C interface used by tests / smoke / bench cpu_baseline through ctypes.
// -----------------------------------------------------------------------------
extern "C" {

void* dgs_oracle_create() { return new Oracle(); }
void dgs_oracle_destroy(void* h) { delete static_cast<Oracle*>(h); }
float dgs_oracle_det_expf(float x) { return det_expf(x); }
// threads used by the forward's preprocess / sort / per-tile blend loops (0: OpenMP default = all cores); returns the count in effect
int dgs_oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : omp_get_num_procs()); return omp_get_max_threads(); }

// Mirrors CudaRasterizer::Rasterizer::forward's argument list (rasterizer.h / rasterizer_impl.cu:198-221);
// null pointer == absent optional input.  Returns num_rendered, or <0 on error.
int dgs_oracle_forward(void* h, int P, int D, int M, const float* background, int W, int H,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, float scale_modifier,
                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                       const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                       int exp_mode) {
    Oracle& o = *static_cast<Oracle*>(h);
    o.P = P; o.D = D; o.M = M; o.W = W; o.H = H; o.tanx = tan_fovx; o.tany = tan_fovy;
    o.scale_mod = scale_modifier; o.exp_mode = exp_mode;
    o.has_sh = shs != nullptr; o.has_cov_pre = cov3D_precomp != nullptr; o.has_col_pre = colors_precomp != nullptr;
    if (P > 0 && !o.has_sh && !o.has_col_pre) return -1;
    if (P > 0 && !o.has_cov_pre && (!scales || !rotations)) return -2;
    const size_t Ps = (size_t)P;
    put(o.means, means3D, 3 * Ps); put(o.shs, shs, 3 * Ps * (size_t)M); put(o.colors_pre, colors_precomp, 3 * Ps);
    put(o.opac, opacities, Ps); put(o.scales, scales, 3 * Ps); put(o.rots, rotations, 4 * Ps);
    put(o.cov_pre, cov3D_precomp, 6 * Ps); put(o.vm, viewmatrix, (size_t)16); put(o.pm, projmatrix, (size_t)16);
    put(o.cam, cam_pos, (size_t)3); put(o.bg, background, (size_t)3);
    return forward(o);
}

// dL_dpix is [3,H,W].  Must follow a dgs_oracle_forward on the same handle.
int dgs_oracle_backward(void* h, const float* dL_dpix, int accum64) {
    Oracle& o = *static_cast<Oracle*>(h);
    backward(o, dL_dpix, accum64);
    return 0;
}

// checkFrustum / markVisible, rasterizer_impl.cu:54-66,141-153
int dgs_oracle_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                            uint8_t* present) {
    (void)projmatrix;
    for (int i = 0; i < P; ++i) {
        const V3 pv = xform4x3({means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]}, viewmatrix);
        present[i] = !(pv.z <= 0.2f);
    }
    return 0;
}

// Named read-only views into the state of the last forward/backward.  Returns element count
// (or -1 for an unknown name); *ptr receives the address; *elem_size the element size in bytes.
long dgs_oracle_get(void* h, const char* name, const void** ptr, int* elem_size) {
    Oracle& o = *static_cast<Oracle*>(h);
    const std::string n(name);
#define VIEW(nm, vec)                                                         \
    if (n == nm) {                                                            \
        *ptr = (vec).data(); *elem_size = (int)sizeof((vec)[0]); return (long)(vec).size(); \
    }
    VIEW("out_color", o.out_color) VIEW("radii", o.radii) VIEW("depths", o.depths)
    VIEW("means2D", o.means2D) VIEW("cov3D", o.cov3D) VIEW("conic_opacity", o.conic_opacity)
    VIEW("rgb", o.rgb) VIEW("clamped", o.clamped) VIEW("tiles_touched", o.tiles_touched)
    VIEW("point_offsets", o.point_offsets) VIEW("keys", o.keys) VIEW("point_list", o.point_list)
    VIEW("ranges", o.ranges) VIEW("n_contrib", o.n_contrib) VIEW("final_T", o.final_T)
    VIEW("dL_dmeans2D", o.g_mean2D) VIEW("dL_dconic", o.g_conic) VIEW("dL_dopacity", o.g_opacity)
    VIEW("dL_dcolors", o.g_color) VIEW("dL_dmeans3D", o.g_mean3D) VIEW("dL_dcov3D", o.g_cov3D)
    VIEW("dL_dsh", o.g_sh) VIEW("dL_dscales", o.g_scale) VIEW("dL_drotations", o.g_rot)
#undef VIEW
    return -1;
}

}  // extern "C"
