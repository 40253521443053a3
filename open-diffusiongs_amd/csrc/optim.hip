// optim.hip -- AdamW step + refresh of the engine's weight copies in one launch (include/dgs_optim.h).
//
// HBM-bound elementwise work: per parameter 16 bytes read (p, g, m, v) and 12 + 2 (+ 2) written.  A tile is 4,096 consecutive
// elements of a flat tensor, or a 64 x 64 block of a matrix that also keeps a transposed bf16 copy (the block goes through LDS
// once so that the transposed rows leave as 16-byte stores too).  256 threads, 16 bytes per lane and access, no atomics, nothing
// depends on the order of the tiles.
#include "dgs_device.h"
#include "dgs_optim.h"
#include "dit_common.h"

namespace dgs {

struct AdamWHyper { float decay, one_minus_b1, b2, one_minus_b2, inv_bc2_sqrt, eps, step_size; const float* grad_sumsq; float max_grad_norm; };

// ---- gradient norm for the global-norm clip (Lightning `gradient_clip_val`, torch.nn.utils.clip_grad_norm_) ----
// One partial per kSumsqChunk consecutive elements, WRITTEN (not accumulated) by the workgroup that owns the chunk: lanes sum their
// elements in a fixed order, the wave and the workgroup combine in a fixed order -- the same bits every run, whatever the launch
// order of the buckets' calls.  dgs_sumsq_finish adds the partials in index order (fp64) into one fp32 word the AdamW launch reads.
constexpr int kSumsqChunk = 65536;

__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ x, long long n, float* __restrict__ partials) {
    __shared__ float s_w[4];
    const long long base = (long long)blockIdx.x * kSumsqChunk;
    const int tid = threadIdx.x;
    float acc = 0.0f;
    if (base + kSumsqChunk <= n) {
        // a full chunk: 64 float4 per thread, eight loads in flight (one load per round trip ran this pass at 0.8 TB/s); the adds keep
        // the order j = 0 .. 63, x, y, z, w -- the same bits as the tail form below
        const float4* src = reinterpret_cast<const float4*>(x + base) + tid;
#pragma unroll 1
        for (int j0 = 0; j0 < kSumsqChunk / 1024; j0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(j0 + u) * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc += v[u].x * v[u].x; acc += v[u].y * v[u].y; acc += v[u].z * v[u].z; acc += v[u].w * v[u].w; }
        }
    } else {
        for (int j = 0; j < kSumsqChunk / 1024; ++j) {
            const long long i = base + (long long)j * 1024 + tid * 4;
            if (i + 3 < n) {
                const float4 v = *reinterpret_cast<const float4*>(x + i);
                acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
            } else {
                for (int e = 0; e < 4; ++e)
                    if (i + e < n) acc += x[i + e] * x[i + e];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) s_w[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ partials, int count, float* __restrict__ total) {
    __shared__ double s_t[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) acc += (double)partials[i];
    s_t[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_t[threadIdx.x] += s_t[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = (float)s_t[0];
}

// the clip coefficient of torch.nn.utils.clip_grad_norm_: max_norm / (total_norm + 1e-6), clamped to 1
__device__ __forceinline__ float clip_coef(const AdamWHyper& h) {
    if (h.grad_sumsq == nullptr) return 1.0f;
    const float c = h.max_grad_norm / (sqrtf(h.grad_sumsq[0]) + 1.0e-6f);
    return c < 1.0f ? c : 1.0f;
}

__device__ __forceinline__ float adamw_one(float& p, float g, float& m, float& v, const AdamWHyper& h) {
    p = p * h.decay;
    m = m + (g - m) * h.one_minus_b1;
    v = v * h.b2 + (h.one_minus_b2 * g) * g;
    const float denom = sqrtf(v) * h.inv_bc2_sqrt + h.eps;
    p = p - h.step_size * (m / denom);
    return p;
}

__global__ __launch_bounds__(256) void adamw_refresh_kernel(const DgsAdamWTensor* __restrict__ tab, int n_tensors, AdamWHyper h) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64][64 + 8];     // bf16 block for the transposed copy (rows padded: 144 B)
    const int tid = threadIdx.x, bid = blockIdx.x;
    // the tensor this tile belongs to: the last entry whose first_tile <= bid (uniform: scalar loads)
    int lo = 0, hi = n_tensors - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first_tile <= bid) lo = mid; else hi = mid - 1;
    }
    const DgsAdamWTensor t = tab[lo];
    const int local = bid - t.first_tile;
    // A non-finite gradient norm (an overflowed loss, a NaN from upstream): the whole update is skipped on every rank alike (the norm
    // is the all-reduced one) -- parameters, both moments and the engine's copies stay as they are -- which is what the reference's
    // 16-mixed training does with such a step (torch.cuda.amp.GradScaler.step skips the optimizer when it found inf / NaN).
    if (h.grad_sumsq != nullptr && (__float_as_uint(h.grad_sumsq[0]) & 0x7f800000u) == 0x7f800000u) return;   // exponent all ones: inf / NaN
    const float gc = clip_coef(h);                                 // 1 without a clip: g * 1.0f is g
    if (t.copy_t == nullptr) {
        // ---- flat tile: elements [local * 4096, +4096) ----
        const long long n = t.rows * t.cols, base = (long long)local * 4096;
        const bool vec = (n & 3) == 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long i = base + (long long)j * 1024 + tid * 4;
            if (i >= n) break;
            if (vec) {
                float4 p = *reinterpret_cast<const float4*>(t.p + i);
                const float4 g = *reinterpret_cast<const float4*>(t.g + i);
                float4 m = *reinterpret_cast<const float4*>(t.m + i), v = *reinterpret_cast<const float4*>(t.v + i);
                adamw_one(p.x, g.x * gc, m.x, v.x, h); adamw_one(p.y, g.y * gc, m.y, v.y, h);
                adamw_one(p.z, g.z * gc, m.z, v.z, h); adamw_one(p.w, g.w * gc, m.w, v.w, h);
                *reinterpret_cast<float4*>(t.p + i) = p;
                *reinterpret_cast<float4*>(t.m + i) = m;
                *reinterpret_cast<float4*>(t.v + i) = v;
                if (t.copy_kind == DGS_OPTIM_COPY_BF16)
                    *reinterpret_cast<uint2*>(static_cast<bf16_t*>(t.copy) + i) = make_uint2(pack_bf2(p.x, p.y), pack_bf2(p.z, p.w));
                else if (t.copy_kind == DGS_OPTIM_COPY_F32) *reinterpret_cast<float4*>(static_cast<float*>(t.copy) + i) = p;
            } else {
                for (int e = 0; e < 4 && i + e < n; ++e) {
                    float p = t.p[i + e], m = t.m[i + e], v = t.v[i + e];
                    adamw_one(p, t.g[i + e] * gc, m, v, h);
                    t.p[i + e] = p; t.m[i + e] = m; t.v[i + e] = v;
                    if (t.copy_kind == DGS_OPTIM_COPY_BF16) static_cast<bf16_t*>(t.copy)[i + e] = (bf16_t)(pack_bf2(p, 0.0f) & 0xffffu);
                    else if (t.copy_kind == DGS_OPTIM_COPY_F32) static_cast<float*>(t.copy)[i + e] = p;
                }
            }
        }
        return;
    }
    // ---- 64 x 64 block of a matrix with a transposed copy ----
    const int tiles_c = (int)(t.cols / 64);
    const int r0 = (local / tiles_c) * 64, c0 = (local % tiles_c) * 64;
    const int c4 = (tid & 15) * 4, rr = tid >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = rr + 16 * j;
        const long long i = (long long)(r0 + r) * t.cols + c0 + c4;
        float4 p = *reinterpret_cast<const float4*>(t.p + i);
        const float4 g = *reinterpret_cast<const float4*>(t.g + i);
        float4 m = *reinterpret_cast<const float4*>(t.m + i), v = *reinterpret_cast<const float4*>(t.v + i);
        adamw_one(p.x, g.x * gc, m.x, v.x, h); adamw_one(p.y, g.y * gc, m.y, v.y, h);
        adamw_one(p.z, g.z * gc, m.z, v.z, h); adamw_one(p.w, g.w * gc, m.w, v.w, h);
        *reinterpret_cast<float4*>(t.p + i) = p;
        *reinterpret_cast<float4*>(t.m + i) = m;
        *reinterpret_cast<float4*>(t.v + i) = v;
        const uint2 b = make_uint2(pack_bf2(p.x, p.y), pack_bf2(p.z, p.w));
        if (t.copy_kind == DGS_OPTIM_COPY_BF16) *reinterpret_cast<uint2*>(static_cast<bf16_t*>(t.copy) + i) = b;
        else if (t.copy_kind == DGS_OPTIM_COPY_F32) *reinterpret_cast<float4*>(static_cast<float*>(t.copy) + i) = p;
        *reinterpret_cast<uint2*>(&tile[r][c4]) = b;
    }
    __syncthreads();
    // transposed: thread -> column c of the block, 16 consecutive rows: 32 contiguous bytes of copy_t[c0 + c][r0 + ...]
    const int c = tid >> 2, rq = (tid & 3) * 16;
    unsigned short u[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) u[e] = tile[rq + e][c];
    bf16_t* dst = static_cast<bf16_t*>(t.copy_t) + (long long)(c0 + c) * t.rows + r0 + rq;
    auto pk = [&](int e) { return (uint32_t)u[e] | ((uint32_t)u[e + 1] << 16); };
    *reinterpret_cast<uint4*>(dst) = make_uint4(pk(0), pk(2), pk(4), pk(6));
    *reinterpret_cast<uint4*>(dst + 8) = make_uint4(pk(8), pk(10), pk(12), pk(14));
}

}  // namespace dgs

extern "C" int32_t dgs_adamw_plan(DgsAdamWTensor* tab, int32_t n) {
    if (!tab || n <= 0) return -1;
    long long tiles = 0;
    for (int i = 0; i < n; ++i) {
        DgsAdamWTensor& t = tab[i];
        if (!t.p || !t.g || !t.m || !t.v || t.rows <= 0 || t.cols <= 0) return -1;
        // the kernel moves 16 bytes per lane (8 for a bf16 copy): a tensor that is a view at an odd element offset has no such alignment
        const auto misaligned = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) != 0; };
        if (misaligned(t.p, 16) || misaligned(t.g, 16) || misaligned(t.m, 16) || misaligned(t.v, 16)) return -1;
        if (t.copy && misaligned(t.copy, t.copy_kind == DGS_OPTIM_COPY_BF16 ? 8 : 16)) return -1;
        if (t.copy_t && misaligned(t.copy_t, 16)) return -1;
        if (t.copy_kind < DGS_OPTIM_COPY_NONE || t.copy_kind > DGS_OPTIM_COPY_F32 || (t.copy_kind != DGS_OPTIM_COPY_NONE && !t.copy)) return -1;
        t.first_tile = (int32_t)tiles;
        if (t.copy_t) {
            if (t.rows % 64 || t.cols % 64) return -1;
            tiles += (t.rows / 64) * (t.cols / 64);
        } else {
            tiles += (t.rows * t.cols + 4095) / 4096;
        }
        if (tiles > 0x7fffffffLL) return -1;
    }
    return (int32_t)tiles;
}

extern "C" int dgs_adamw_step(const DgsAdamWArgs* a, dgs_stream_t stream) {
    if (!a || !a->tensors || a->n_tensors <= 0 || a->n_tiles <= 0 || !(a->bias_correction1 > 0.0f) || !(a->bias_correction2_sqrt > 0.0f))
        return DGS_ERR_INVALID_ARGUMENT;
    dgs::AdamWHyper h;
    h.decay = 1.0f - a->lr * a->weight_decay;
    h.one_minus_b1 = 1.0f - a->beta1;
    h.b2 = a->beta2;
    h.one_minus_b2 = 1.0f - a->beta2;
    h.inv_bc2_sqrt = 1.0f / a->bias_correction2_sqrt;
    h.eps = a->eps;
    h.step_size = a->lr / a->bias_correction1;
    h.grad_sumsq = a->max_grad_norm > 0.0f ? a->grad_sumsq : nullptr;
    h.max_grad_norm = a->max_grad_norm;
    if (a->max_grad_norm > 0.0f && !a->grad_sumsq) return DGS_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(dgs::adamw_refresh_kernel, dim3(a->n_tiles), dim3(256), 0, static_cast<hipStream_t>(stream), a->tensors, a->n_tensors, h);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

extern "C" int32_t dgs_sumsq_count(int64_t n) { return n <= 0 ? 0 : (int32_t)((n + dgs::kSumsqChunk - 1) / dgs::kSumsqChunk); }

extern "C" int dgs_sumsq_partials(const float* x, int64_t n, float* partials, dgs_stream_t stream) {
    if (!x || !partials || n <= 0 || (reinterpret_cast<uintptr_t>(x) & 15)) return DGS_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(dgs::sumsq_partials_kernel, dim3(dgs_sumsq_count(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long long)n, partials);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

extern "C" int dgs_sumsq_finish(const float* partials, int32_t count, float* total, dgs_stream_t stream) {
    if (!partials || !total || count <= 0) return DGS_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(dgs::sumsq_finish_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), partials, count, total);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
