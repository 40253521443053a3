// dit_attention.hip -- flash-style fused attention forward for the DiT blocks (gfx950, wave64, bf16 MFMA).
//
// Replaces F.scaled_dot_product_attention(q, k, v) as called by timm==0.9.16 Attention.forward (non-causal, no mask,
// scale = head_dim^-1/2, head_dim = 64), used by DiTBlock (utils_transformer.py:254-256, 286).  The S = L x L score
// matrix is never materialised: per 128-query workgroup (4 waves x 32 queries) the kernel walks 64-key tiles with an
// online softmax (fp32 running max / sum, fp32 accumulators).
//
// Formulation (cdna_hip_programming.md appendix B, "swapped QK^T"): every MFMA is computed transposed,
//     S^T[key, q] = K . Q^T            A = K fragment (LDS),  B = Q fragment (registers, loaded once)
//     O^T[d,  q]  = V^T . P^T          A = V^T fragment (LDS), B = P^T fragment (registers)
// so the query index is the lane (column of the 32x32 accumulator): softmax statistics and the O rescale are per-lane
// scalars, the row max needs ONE cross-lane exchange (lane ^ 32), and the P^T accumulator registers of a lane are --
// after bf16 packing -- directly the B fragment of the second MFMA (the k-slot <-> key mapping of an MFMA is free as long
// as both operands agree, so the V^T fragment is simply read with the accumulator's row pattern).
// V arrives already transposed ([B, heads*64, lpad]) from the QKV GEMM epilogue, so no transposing LDS access is needed.
//
// LDS: K tile [64 keys][128 B] with 16-byte chunk swizzle c ^ ((key >> 1) & 7) (conflict-free ds_read_b128), V^T tile
// [64 d][128 B] with 8-byte slot swizzle s ^ ((d >> 1) & 15) (conflict-free ds_read_b64); two stages, register-staged
// prefetch of the next tile issued before the MFMAs of the current one (T14), one barrier per tile.
#include "dit_common.h"
#include "dgs_dit.h"

namespace dgs {

constexpr int QB = 128;          // queries per workgroup
constexpr int KB = 64;           // keys per tile
constexpr int KV_TILE_BYTES = KB * 64 * 2;   // 8 KiB

struct AttnParams {
    int B, heads, L, lpad, ld_qk;
    const bf16_t* qk;
    const bf16_t* vt;
    bf16_t* out;
    float scale_log2e;
};

__global__ __launch_bounds__(256) void attention_fwd_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 2 * KV_TILE_BYTES];   // [stage][K, V^T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qblk = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int l31 = lane & 31, half = lane >> 5;
    const size_t row0 = (size_t)b * p.lpad;
    const bf16_t* Qg = p.qk + row0 * p.ld_qk + head * 64;
    const bf16_t* Kg = Qg + p.heads * 64;
    const bf16_t* Vg = p.vt + ((size_t)b * p.heads + head) * 64 * p.lpad;

    // Q fragments (B operand): query = lane & 31, d-chunk = 2 ks + half
    const int q = qblk * QB + wave * 32 + l31;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Qg + (size_t)q * p.ld_qk + (2 * ks + half) * 8);

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.0f;
    float m_run = -1.0e30f, l_run = 0.0f;

    const int ntiles = (p.L + KB - 1) / KB;
    // staging assignment: 512 16-byte chunks per tile, two per thread
    uint4 kreg[2], vreg[2];
    auto issue_loads = [&](int t) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j, r = i >> 3, c = i & 7;
            kreg[j] = *reinterpret_cast<const uint4*>(Kg + (size_t)(t * KB + r) * p.ld_qk + c * 8);
            vreg[j] = *reinterpret_cast<const uint4*>(Vg + (size_t)r * p.lpad + t * KB + c * 8);
        }
    };
    auto write_lds = [&](int stage) {
        char* kb = lds + stage * 2 * KV_TILE_BYTES;
        char* vb = kb + KV_TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j, r = i >> 3, c = i & 7;
            *reinterpret_cast<uint4*>(kb + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = kreg[j];
            const int sw = (r >> 1) & 15;
            *reinterpret_cast<uint2*>(vb + r * 128 + (((2 * c) ^ sw) << 3)) = make_uint2(vreg[j].x, vreg[j].y);
            *reinterpret_cast<uint2*>(vb + r * 128 + (((2 * c + 1) ^ sw) << 3)) = make_uint2(vreg[j].z, vreg[j].w);
        }
    };
    issue_loads(0);
    write_lds(0);
    __syncthreads();

    const int kswz = (l31 >> 1) & 7, vswz = (l31 >> 1) & 15;
    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) issue_loads(t + 1);
        const char* kb = lds + (t & 1) * 2 * KV_TILE_BYTES;
        const char* vb = kb + KV_TILE_BYTES;

        // ---- S^T = K . Q^T : two 32-key blocks ----
        f32x16 s[2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kbk][r] = 0.0f;
            const char* krow = kb + (kbk * 32 + l31) * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + (((2 * ks + half) ^ kswz) << 4));
                s[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kbk], 0, 0, 0);
            }
        }
        // ---- mask keys >= L (only the last tile can cross) ----
        if ((t + 1) * KB > p.L) {
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KB + kbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (key >= p.L) s[kbk][r] = -__builtin_inff();
                }
        }
        // ---- online softmax (per-lane query) ----
        float mx = s[0][0];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kbk][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f((m_run - m_new) * p.scale_log2e);
        const float mb = m_new * p.scale_log2e;
        float psum = 0.0f;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = exp2f(s[kbk][r] * p.scale_log2e - mb);
                s[kbk][r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;

        // ---- O^T += V^T . P^T : k-step ks covers keys 16 ks .. 16 ks + 15 of the tile ----
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kbk = ks >> 1, r0 = 8 * (ks & 1);
            union { bf16x8 v; uint32_t u[4]; } pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) pf.u[j] = pack_bf2(s[kbk][r0 + 2 * j], s[kbk][r0 + 2 * j + 1]);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const char* vrow = vb + (db * 32 + l31) * 128;
                union { bf16x8 v; uint2 h[2]; } vf;
                vf.h[0] = *reinterpret_cast<const uint2*>(vrow + (((4 * ks + half) ^ vswz) << 3));
                vf.h[1] = *reinterpret_cast<const uint2*>(vrow + (((4 * ks + 2 + half) ^ vswz) << 3));
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, oacc[db], 0, 0, 0);
            }
        }
        if (more) write_lds((t + 1) & 1);
        __syncthreads();
    }

    // ---- finish: O[q, d] = O^T / l ; lane owns query q, d = db*32 + 8 (r >> 2) + 4 half + (r & 3) ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    bf16_t* orow = p.out + (row0 + q) * (size_t)(p.heads * 64) + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 v;
            v.x = pack_bf2(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv);
            v.y = pack_bf2(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(orow + db * 32 + 8 * g + 4 * half) = v;
        }
}

}  // namespace dgs

using namespace dgs;

extern "C" int dgs_dit_attention(const DgsDitAttentionArgs* a, dgs_stream_t stream) {
    if (!a || a->B <= 0 || a->heads <= 0 || a->L <= 0 || a->lpad < a->L || a->lpad % QB || !a->qk || !a->vt || !a->out)
        return DGS_ERR_INVALID_ARGUMENT;
    AttnParams p;
    p.B = a->B; p.heads = a->heads; p.L = a->L; p.lpad = a->lpad; p.ld_qk = 2 * a->heads * 64;
    p.qk = a->qk; p.vt = a->vt; p.out = a->out;
    p.scale_log2e = a->scale * 1.44269504088896341f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(attention_fwd_kernel, dim3(a->lpad / QB, a->heads, a->B), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
