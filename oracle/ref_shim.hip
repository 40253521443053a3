// =============================================================================
// ref_shim.hip -- host-pointer C ABI around the REFERENCE rasterizer itself.
//
// TEST INFRASTRUCTURE ONLY (same rule as raster_oracle.cpp): loaded by tests/ and
// oracle/make_raster_ref_golden.py, never by the product path.
//
// oracle/build_ref.py translates the reference's own CUDA sources
//   /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/
//       {auxiliary.h, config.h, forward.{h,cu}, backward.{h,cu}, rasterizer.h, rasterizer_impl.{h,cu}}
// with hipify-perl into the git-ignored oracle/_ref/src/ at build time and links them with
// this file into oracle/_ref/libdgs_ref_{strict,fast}.so.  Nothing of the reference is
// committed.  This file is the stand-in for the reference's torch binding
// (rasterize_points.cu:35-196): same zero-initialised outputs, same grow-by-callback
// scratch buffers, same argument order into CudaRasterizer::Rasterizer::forward/backward
// (rasterizer.h:24-84) -- but with host arrays at the boundary and the SAME named views
// as dgs_oracle_get() so that tests can compare the two field by field.
// =============================================================================
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "rasterizer.h"
#include "rasterizer_impl.h"

namespace {

struct DevBuf {
    char* p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    char* grow(size_t want) {  // rasterize_points.cu:27-33 (resizeFunctional)
        if (want > n) {
            if (p) (void)hipFree(p);
            (void)hipMalloc((void**)&p, want);
            n = want;
        }
        return p;
    }
    template <class T>
    T* up(const T* h, size_t cnt) {
        if (!h || cnt == 0) { return nullptr; }
        grow(cnt * sizeof(T));
        (void)hipMemcpy(p, h, cnt * sizeof(T), hipMemcpyHostToDevice);
        return reinterpret_cast<T*>(p);
    }
    template <class T>
    T* zeros(size_t cnt) {
        grow((cnt ? cnt : 1) * sizeof(T));
        (void)hipMemset(p, 0, (cnt ? cnt : 1) * sizeof(T));
        return reinterpret_cast<T*>(p);
    }
};

struct Ref {
    int P = 0, D = 0, M = 0, W = 0, H = 0, R = 0;
    float tanx = 0, tany = 0, scale_mod = 1;
    DevBuf bg, means, shs, colors, opac, scales, rots, cov, vm, pm, cam;   // inputs
    DevBuf geom, binning, img;                                              // the reference's three scratch buffers
    DevBuf out_color, radii;
    DevBuf dpix, g_mean2D, g_conic, g_opacity, g_color, g_mean3D, g_cov3D, g_sh, g_scale, g_rot;
    const float *d_bg = nullptr, *d_means = nullptr, *d_shs = nullptr, *d_colors = nullptr, *d_opac = nullptr,
                *d_scales = nullptr, *d_rots = nullptr, *d_cov = nullptr, *d_vm = nullptr, *d_pm = nullptr, *d_cam = nullptr;
    std::map<std::string, std::vector<char>> host;   // named host copies handed out by dgs_ref_get
    double fwd_ms = 0, bwd_ms = 0;
};

template <class T>
void stash(Ref& r, const char* name, const T* dev, size_t cnt) {
    std::vector<char>& v = r.host[name];
    v.resize(cnt * sizeof(T));
    if (cnt) (void)hipMemcpy(v.data(), dev, cnt * sizeof(T), hipMemcpyDeviceToHost);
}

}  // namespace

extern "C" {

void* dgs_ref_create() { return new Ref(); }
void dgs_ref_destroy(void* h) { delete static_cast<Ref*>(h); }

// Argument list of CudaRasterizer::Rasterizer::forward (rasterizer.h:31-52) with HOST pointers; null == absent.
// Returns num_rendered or <0.
int dgs_ref_forward(void* h, int P, int D, int M, const float* background, int W, int H, const float* means3D,
                    const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                    float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                    const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int repeat) {
    Ref& r = *static_cast<Ref*>(h);
    r.P = P; r.D = D; r.M = M; r.W = W; r.H = H; r.tanx = tan_fovx; r.tany = tan_fovy; r.scale_mod = scale_modifier;
    r.host.clear();
    const size_t Ps = (size_t)P;
    r.d_bg = r.bg.up(background, 3);
    r.d_means = r.means.up(means3D, 3 * Ps);
    r.d_shs = r.shs.up(shs, 3 * Ps * (size_t)M);
    r.d_colors = r.colors.up(colors_precomp, 3 * Ps);
    r.d_opac = r.opac.up(opacities, Ps);
    r.d_scales = r.scales.up(scales, 3 * Ps);
    r.d_rots = r.rots.up(rotations, 4 * Ps);
    r.d_cov = r.cov.up(cov3D_precomp, 6 * Ps);
    r.d_vm = r.vm.up(viewmatrix, 16);
    r.d_pm = r.pm.up(projmatrix, 16);
    r.d_cam = r.cam.up(cam_pos, 3);
    float* out_color = r.out_color.zeros<float>((size_t)3 * W * H);   // torch::full(.., 0.0), rasterize_points.cu:68
    int* radii = r.radii.zeros<int>(Ps);                               // rasterize_points.cu:69
    int rendered = 0;
    if (P != 0) {
        std::function<char*(size_t)> gf = [&r](size_t n) { return r.geom.grow(n); };
        std::function<char*(size_t)> bf = [&r](size_t n) { return r.binning.grow(n); };
        std::function<char*(size_t)> imf = [&r](size_t n) { return r.img.grow(n); };
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int it = 0; it < (repeat > 0 ? repeat : 1); ++it) {
            if (it == 1) (void)hipEventRecord(e0, 0);
            if (it > 0) { (void)hipMemsetAsync(out_color, 0, sizeof(float) * 3 * W * H, 0); (void)hipMemsetAsync(radii, 0, sizeof(int) * Ps, 0); }
            rendered = CudaRasterizer::Rasterizer::forward(gf, bf, imf, P, D, M, r.d_bg, W, H, r.d_means, r.d_shs, r.d_colors,
                                                           r.d_opac, r.d_scales, scale_modifier, r.d_rots, r.d_cov, r.d_vm,
                                                           r.d_pm, r.d_cam, tan_fovx, tan_fovy, false, out_color, radii, false);
        }
        if (repeat > 1) {
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            r.fwd_ms = ms / (repeat - 1);
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    r.R = rendered;
    stash(r, "out_color", out_color, (size_t)3 * W * H);
    stash(r, "radii", radii, Ps);
    if (P != 0) {
        char* c = r.geom.p;
        CudaRasterizer::GeometryState g = CudaRasterizer::GeometryState::fromChunk(c, Ps);   // rasterizer_impl.cu:155-170
        stash(r, "depths", g.depths, Ps);
        stash(r, "means2D", reinterpret_cast<float*>(g.means2D), 2 * Ps);
        stash(r, "cov3D", g.cov3D, 6 * Ps);
        stash(r, "conic_opacity", reinterpret_cast<float*>(g.conic_opacity), 4 * Ps);
        stash(r, "rgb", g.rgb, 3 * Ps);
        stash(r, "clamped", reinterpret_cast<uint8_t*>(g.clamped), 3 * Ps);
        stash(r, "tiles_touched", g.tiles_touched, Ps);
        stash(r, "point_offsets", g.point_offsets, Ps);
        char* ic = r.img.p;
        CudaRasterizer::ImageState im = CudaRasterizer::ImageState::fromChunk(ic, (size_t)W * H);  // :172-179
        const size_t T = (size_t)((W + 15) / 16) * ((H + 15) / 16);
        stash(r, "ranges", reinterpret_cast<uint32_t*>(im.ranges), 2 * T);
        stash(r, "n_contrib", im.n_contrib, (size_t)W * H);
        stash(r, "final_T", im.accum_alpha, (size_t)W * H);
        if (rendered > 0) {
            char* bc = r.binning.p;
            CudaRasterizer::BinningState b = CudaRasterizer::BinningState::fromChunk(bc, (size_t)rendered);  // :181-194
            stash(r, "keys", b.point_list_keys, (size_t)rendered);
            stash(r, "point_list", b.point_list, (size_t)rendered);
        } else {
            r.host["keys"]; r.host["point_list"];
        }
    }
    return rendered;
}

// RasterizeGaussiansBackwardCUDA (rasterize_points.cu:118-196): zero-initialised gradients, then Rasterizer::backward.
int dgs_ref_backward(void* h, const float* dL_dpix, int repeat) {
    Ref& r = *static_cast<Ref*>(h);
    const size_t Ps = (size_t)r.P, M = (size_t)r.M;
    const float* dpix = r.dpix.up(dL_dpix, (size_t)3 * r.W * r.H);
    float *gm2 = nullptr, *gcn = nullptr, *gop = nullptr, *gcol = nullptr, *gm3 = nullptr, *gcv = nullptr, *gsh = nullptr, *gsc = nullptr, *grt = nullptr;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < (repeat > 0 ? repeat : 1); ++it) {
        if (it == 1) (void)hipEventRecord(e0, 0);
        gm3 = r.g_mean3D.zeros<float>(3 * Ps); gm2 = r.g_mean2D.zeros<float>(3 * Ps); gcol = r.g_color.zeros<float>(3 * Ps);
        gcn = r.g_conic.zeros<float>(4 * Ps); gop = r.g_opacity.zeros<float>(Ps); gcv = r.g_cov3D.zeros<float>(6 * Ps);
        gsh = r.g_sh.zeros<float>(3 * Ps * M); gsc = r.g_scale.zeros<float>(3 * Ps); grt = r.g_rot.zeros<float>(4 * Ps);
        if (r.P != 0)
            CudaRasterizer::Rasterizer::backward(r.P, r.D, r.M, r.R, r.d_bg, r.W, r.H, r.d_means, r.d_shs, r.d_colors, r.d_scales,
                                                 r.scale_mod, r.d_rots, r.d_cov, r.d_vm, r.d_pm, r.d_cam, r.tanx, r.tany,
                                                 reinterpret_cast<int*>(r.radii.p), r.geom.p, r.binning.p, r.img.p, dpix, gm2, gcn,
                                                 gop, gcol, gm3, gcv, gsh, gsc, grt, false);
    }
    if (repeat > 1) {
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        r.bwd_ms = ms / (repeat - 1);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    stash(r, "dL_dmeans2D", gm2, 3 * Ps); stash(r, "dL_dconic", gcn, 4 * Ps); stash(r, "dL_dopacity", gop, Ps);
    stash(r, "dL_dcolors", gcol, 3 * Ps); stash(r, "dL_dmeans3D", gm3, 3 * Ps); stash(r, "dL_dcov3D", gcv, 6 * Ps);
    stash(r, "dL_dsh", gsh, 3 * Ps * M); stash(r, "dL_dscales", gsc, 3 * Ps); stash(r, "dL_drotations", grt, 4 * Ps);
    return 0;
}

// markVisible (rasterize_points.cu:198-216)
int dgs_ref_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present) {
    DevBuf m, v, p, o;
    float* dm = m.up(means3D, (size_t)3 * P);
    float* dv = v.up(viewmatrix, 16);
    float* dp = p.up(projmatrix, 16);
    bool* dout = reinterpret_cast<bool*>(o.zeros<uint8_t>((size_t)P));
    if (P != 0) CudaRasterizer::Rasterizer::markVisible(P, dm, dv, dp, dout);
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    if (P) (void)hipMemcpy(present, dout, (size_t)P, hipMemcpyDeviceToHost);
    return 0;
}

// Same contract as dgs_oracle_get: element count (or -1), *ptr = host copy, *elem_size = bytes per element.
long dgs_ref_get(void* h, const char* name, const void** ptr, int* elem_size) {
    Ref& r = *static_cast<Ref*>(h);
    auto it = r.host.find(name);
    if (it == r.host.end()) return -1;
    const std::string n(name);
    int es = 4;
    if (n == "keys") es = 8;
    if (n == "clamped") es = 1;
    *ptr = it->second.data();
    *elem_size = es;
    return (long)(it->second.size() / (size_t)es);
}

// Mean GPU time of the repeated calls (repeat > 1), milliseconds: what the reference's own kernels cost on this GPU.
double dgs_ref_time_ms(void* h, int which) { return which == 0 ? static_cast<Ref*>(h)->fwd_ms : static_cast<Ref*>(h)->bwd_ms; }

}  // extern "C"
