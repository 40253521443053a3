// camera.hip -- all (sample, view) cameras of a step in ONE launch (gfx950).
//
// Replaces the per-view `Camera` module of the reference (models/gsrenderer/gs_core.py:277-316): C2W.inverse(),
// tanfov = size / (2 f), the OpenCV projection matrix (znear .01, zfar 100) and full_proj = W2C^T . P^T -- about 15 tiny
// device kernels plus implicit .item() syncs per view there; here one thread per camera, no host round trip.
#include "dgs_device.h"
#include "dgs_raster.h"

namespace dgs {

// General 4x4 inverse, Gauss-Jordan with partial pivoting in fp32 (torch's inverse is LU with partial pivoting).
__device__ __forceinline__ void inv4(const float* m, float* out) {
    float a[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[r][c] = m[4 * r + c]; a[r][4 + c] = (r == c) ? 1.0f : 0.0f; }
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        float best = fabsf(a[col][col]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r > col && fabsf(a[r][col]) > best) { best = fabsf(a[r][col]); piv = r; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r == piv && piv != col)
#pragma unroll
                for (int c = 0; c < 8; ++c) { const float t = a[col][c]; a[col][c] = a[r][c]; a[r][c] = t; }
        const float d = 1.0f / a[col][col];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[col][c] *= d;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r != col) {
                const float f = a[r][col];
#pragma unroll
                for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c];
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[4 * r + c] = a[r][4 + c];
}

__global__ void cameras_kernel(int n, const float* c2w, const float* fxfycxcy, int H, int W, float znear, float zfar,
                               float* view, float* proj, float* campos, float* tanfov) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float w2c[16];
    inv4(c2w + 16 * i, w2c);
    const float fx = fxfycxcy[4 * i], fy = fxfycxcy[4 * i + 1], cx = fxfycxcy[4 * i + 2], cy = fxfycxcy[4 * i + 3];
    float P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) P[k] = 0.0f;
    P[0] = 2.0f * fx / (float)W;
    P[5] = 2.0f * fy / (float)H;
    P[2] = 2.0f * (cx / (float)W) - 1.0f;
    P[6] = 2.0f * (cy / (float)H) - 1.0f;
    P[10] = -(zfar + znear) / (zfar - znear);
    P[14] = 1.0f;
    P[11] = -(2.0f * zfar * znear) / (zfar - znear);
    // view = W2C^T ; proj = view . P^T  => proj[r][c] = sum_k w2c[k][r] * P[c][k]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            view[16 * i + 4 * r + c] = w2c[4 * c + r];
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += w2c[4 * k + r] * P[4 * c + k];
            proj[16 * i + 4 * r + c] = s;
        }
    campos[3 * i] = c2w[16 * i + 3]; campos[3 * i + 1] = c2w[16 * i + 7]; campos[3 * i + 2] = c2w[16 * i + 11];
    tanfov[2 * i] = (float)W / (2.0f * fx);
    tanfov[2 * i + 1] = (float)H / (2.0f * fy);
}

// Per-pixel rays of all (sample, view) cameras in one launch: TransformInput, diffusionGS/systems/utils.py:621-684,751-757
// (the step immediately before image_to_gaussians; the reference builds them with a meshgrid, a bmm and two norms per call).
// ray_d = normalize(R [ (x + .5 - cx) / fx, (y + .5 - cy) / fy, 1 ]),  ray_o = camera centre; outputs [n, 3, H, W] planar.
__global__ __launch_bounds__(256) void rays_kernel(int n, const float* c2w, const float* fxfycxcy, int H, int W, float* ray_o, float* ray_d) {
    const size_t HW = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * HW) return;
    const int cam = (int)(i / HW);
    const int px = (int)(i % W), py = (int)((i / W) % H);
    const float* m = c2w + 16 * cam;
    const float* k = fxfycxcy + 4 * cam;
    const float x = ((float)px + 0.5f - k[2]) / k[0];
    const float y = ((float)py + 0.5f - k[3]) / k[1];
    float dx = x * m[0] + y * m[1] + m[2];
    float dy = x * m[4] + y * m[5] + m[6];
    float dz = x * m[8] + y * m[9] + m[10];
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    dx /= nrm; dy /= nrm; dz /= nrm;
    const size_t o = (size_t)cam * 3 * HW + (size_t)py * W + px;
    ray_d[o] = dx; ray_d[o + HW] = dy; ray_d[o + 2 * HW] = dz;
    ray_o[o] = m[3]; ray_o[o + HW] = m[7]; ray_o[o + 2 * HW] = m[11];
}

}  // namespace dgs

extern "C" int dgs_rays_from_c2w(int32_t n, const float* c2w, const float* fxfycxcy, int32_t height, int32_t width, float* ray_o,
                                 float* ray_d, dgs_stream_t stream) {
    if (n < 0 || height <= 0 || width <= 0) return DGS_ERR_INVALID_ARGUMENT;
    if (n == 0) return DGS_OK;
    if (!c2w || !fxfycxcy || !ray_o || !ray_d) return DGS_ERR_INVALID_ARGUMENT;
    const size_t total = (size_t)n * height * width;
    hipLaunchKernelGGL(dgs::rays_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n, c2w,
                       fxfycxcy, height, width, ray_o, ray_d);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

extern "C" int dgs_cameras_from_c2w(int32_t n, const float* c2w, const float* fxfycxcy, int32_t height, int32_t width,
                                    float znear, float zfar, float* viewmatrix, float* projmatrix, float* campos,
                                    float* tanfov, dgs_stream_t stream) {
    if (n < 0 || height <= 0 || width <= 0) return DGS_ERR_INVALID_ARGUMENT;
    if (n == 0) return DGS_OK;
    if (!c2w || !fxfycxcy || !viewmatrix || !projmatrix || !campos || !tanfov) return DGS_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(dgs::cameras_kernel, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), n, c2w, fxfycxcy,
                       height, width, znear, zfar, viewmatrix, projmatrix, campos, tanfov);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
