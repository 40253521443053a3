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
// LDS: K tile [64 keys][128 B] and V^T tile [64 d][128 B], both with the 16-byte slot swizzle s ^ ((row >> 1) & 7)
// (conflict-free ds_read_b128); inside a V^T row every 16-key group is stored as [k0-3, k8-11 | k4-7, k12-15] so the 8
// keys a lane contributes to one MFMA k-step are one 16-byte read.  Two stages, register-staged prefetch of the next
// tile issued before the MFMAs of the current one (T14), one barrier per tile.
#include "dit_common.h"
#include "dgs_dit.h"

namespace dgs {

constexpr int QB = 128;          // queries per workgroup
constexpr int KB = 64;           // keys per tile
constexpr int KV_TILE_BYTES = KB * 64 * 2;   // 8 KiB
constexpr float RESCALE_THR = 6.0f;          // deferred-max threshold in exp2 units: P <= 64

struct MaskOn { static constexpr bool value = true; };
struct MaskOff { static constexpr bool value = false; };

struct AttnParams {
    int B, heads, L, lpad, ld_qk;
    const bf16_t* qk;
    const bf16_t* vt;
    bf16_t* out;
    float scale_log2e;
};

__global__ __launch_bounds__(256, 2) void attention_fwd_kernel(AttnParams p) {
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
    // staging assignment: 512 16-byte chunks per tile, two per thread (chunk i = tid, tid + 256: row i >> 3, column i & 7).
    // Named scalars (not arrays) so the prefetch registers never go through scratch.
    const int sr = tid >> 3, sc = tid & 7;                 // rows sr and sr + 32
    const bf16_t* kptr = Kg + (size_t)sr * p.ld_qk + sc * 8;
    const bf16_t* vptr = Vg + (size_t)sr * p.lpad + sc * 8;
    const size_t kstep = (size_t)32 * p.ld_qk, vstep = (size_t)32 * p.lpad;
    const int ksw = (sr >> 1) & 7;                         // (sr + 32) >> 1 has the same low 3 bits
    const int koff = sr * 128 + ((sc ^ ksw) << 4);
    // V^T row d: each 16-key group is stored as [k0-3, k8-11 | k4-7, k12-15] so that the 8 keys one lane needs for an
    // MFMA k-step (accumulator rows 4h..4h+3 and 8+4h..8+4h+3) are ONE 16-byte slot: slot 2g + h of the row.
    // The chunk this thread loaded (keys 8 sc .. 8 sc + 7) is group g = sc >> 1, keys 8 (sc & 1) + {0-3 | 4-7}.
    const int vslot = (sc >> 1) * 2;
    const int voff0 = sr * 128 + (((vslot) ^ ksw) << 4) + 8 * (sc & 1);       // keys +0..3  -> slot 2g   , half sc & 1
    const int voff1 = sr * 128 + (((vslot + 1) ^ ksw) << 4) + 8 * (sc & 1);   // keys +4..7  -> slot 2g+1 , half sc & 1
    uint4 kreg0, kreg1, vreg0, vreg1;
#define ATTN_ISSUE_LOADS(t)                                                                      \
    do {                                                                                         \
        kreg0 = *reinterpret_cast<const uint4*>(kptr + (size_t)(t) * KB * p.ld_qk);              \
        kreg1 = *reinterpret_cast<const uint4*>(kptr + (size_t)(t) * KB * p.ld_qk + kstep);      \
        vreg0 = *reinterpret_cast<const uint4*>(vptr + (size_t)(t) * KB);                        \
        vreg1 = *reinterpret_cast<const uint4*>(vptr + (size_t)(t) * KB + vstep);                \
    } while (0)
#define ATTN_WRITE_LDS(stage)                                                                    \
    do {                                                                                         \
        char* kb_ = lds + (stage) * 2 * KV_TILE_BYTES;                                           \
        char* vb_ = kb_ + KV_TILE_BYTES;                                                         \
        *reinterpret_cast<uint4*>(kb_ + koff) = kreg0;                                           \
        *reinterpret_cast<uint4*>(kb_ + koff + 32 * 128) = kreg1;                                \
        *reinterpret_cast<uint2*>(vb_ + voff0) = make_uint2(vreg0.x, vreg0.y);                   \
        *reinterpret_cast<uint2*>(vb_ + voff1) = make_uint2(vreg0.z, vreg0.w);                   \
        *reinterpret_cast<uint2*>(vb_ + voff0 + 32 * 128) = make_uint2(vreg1.x, vreg1.y);        \
        *reinterpret_cast<uint2*>(vb_ + voff1 + 32 * 128) = make_uint2(vreg1.z, vreg1.w);        \
    } while (0)
    ATTN_ISSUE_LOADS(0);
    ATTN_WRITE_LDS(0);
    // Retire the Q-fragment loads HERE: left pending, hipcc's in-order vmcnt bookkeeping makes every tile's first MFMAs
    // wait for that tile's just-issued K/V prefetch (vmcnt(3..0) inside the QK^T chain), serialising HBM latency.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();

    const int kswz = (l31 >> 1) & 7;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // One 64-key tile.  MASK is instantiated only for the last tile (keys >= L masked to -inf).
    // a wave whose 32 queries are all padding rows only helps with staging and barriers
    const bool wave_live = qblk * QB + wave * 32 < p.L;
    auto tile = [&](int t, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        if (!wave_live) return;
        const char* kb = lds + (t & 1) * 2 * KV_TILE_BYTES;
        const char* vb = kb + KV_TILE_BYTES;
        // ---- S^T = K . Q^T : two 32-key blocks ----
        f32x16 s[2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            const char* krow = kb + (kbk * 32 + l31) * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + (((2 * ks + half) ^ kswz) << 4));
                s[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? zero16 : s[kbk], 0, 0, 0);
            }
        }
        if (MASK) {
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KB + kbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (key >= p.L) s[kbk][r] = -__builtin_inff();
                }
        }
        // ---- online softmax, per-lane query.  Deferred rescale (T13): the running max only moves when some query of
        //      the wave outgrew it by more than RESCALE_THR (in exp2 units), so P stays <= 2^THR and the O / l rescale
        //      is skipped on almost every tile.
        float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
        mx = xor32_max(mx);
        if (__any((mx - m_run) * p.scale_log2e > RESCALE_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = fast_exp2((m_run - m_new) * p.scale_log2e);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
        const float mb = m_run * p.scale_log2e;
        float psum = 0.0f;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = fast_exp2(__builtin_fmaf(s[kbk][r], p.scale_log2e, -mb));
                s[kbk][r] = e;
                psum += e;
            }
        l_run += psum;
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
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vrow + (((2 * ks + half) ^ kswz) << 4));
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, oacc[db], 0, 0, 0);
            }
        }
    };
    // Full tiles in the loop, the (at most one) partially valid tile peeled behind it: no mask code and no
    // data-dependent branch on the hot path.
    const bool last_masked = ntiles * KB > p.L;
    const int nfull = last_masked ? ntiles - 1 : ntiles;
    for (int t = 0; t < nfull; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) ATTN_ISSUE_LOADS(t + 1);
        tile(t, MaskOff{});
        if (more) ATTN_WRITE_LDS((t + 1) & 1);
        __syncthreads();
    }
    if (last_masked) tile(ntiles - 1, MaskOn{});

    // ---- finish: O[q, d] = O^T / l ; lane owns query q, d = db*32 + 8 (r >> 2) + 4 half + (r & 3) ----
    if (!wave_live) return;
    const float l_tot = xor32_sum(l_run);
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
