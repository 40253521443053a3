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

constexpr int QB = 256;          // queries per workgroup: 8 waves x 32
constexpr int NW = QB / 32;
constexpr int KB = 64;           // keys per tile
constexpr int KV_TILE_BYTES = KB * 64 * 2;   // 8 KiB
constexpr float RESCALE_THR = 6.0f;          // deferred-max threshold in exp2 units: P <= 64

struct AttnParams {
    int B, heads, L, lpad, ld_qk, nqb_full, nqb;
    const bf16_t* qk;
    const bf16_t* vt;
    bf16_t* out;
    float scale_log2e;
};

// S^T block pair of one 64-key tile: A = K fragments from LDS, B = Q fragments (registers).
__device__ __forceinline__ void qk_tile(const char* kb, const bf16x8 (&qf)[4], int l31, int half, int kswz, f32x16& s0, f32x16& s1) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const char* krow = kb + l31 * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int off = ((2 * ks + half) ^ kswz) << 4;
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(krow + off);
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(krow + 32 * 128 + off);
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ks], ks == 0 ? zero16 : s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ks], ks == 0 ? zero16 : s1, 0, 0, 0);
    }
}

__device__ __forceinline__ void mask_tile(f32x16& s0, f32x16& s1, int key0, int half, int L) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (key >= L) s0[r] = -__builtin_inff();
        if (key + 32 >= L) s1[r] = -__builtin_inff();
    }
}

// Workgroup = 256 queries of one (sample, head): 8 waves x 32 queries, two waves per SIMD.  The grid is 1-D and ordered so
// that (a) head = id % heads, i.e. all query blocks of a head run on the same XCD (dispatch places block b on XCD b % 8)
// and re-read its K / V^T (1 MiB) from that XCD's L2, and (b) the mostly-padding last query block of every (sample,
// head) comes LAST, so that at batch 1 the 256 full blocks are exactly one workgroup per CU.
// Software pipeline (T15): while the VALU works through the softmax of tile t, the matrix pipe already runs
// S(t+1) = K(t+1) Q^T; K and V^T therefore live in two 2-deep rings that are one tile out of phase.
__global__ __launch_bounds__(512, 2) void attention_fwd_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char lds[4 * KV_TILE_BYTES];   // K ring [2] | V^T ring [2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // ---- block -> (sample, head, query block) ----
    int id = blockIdx.x, b, head, qblk;
    const int nfull_blocks = p.B * p.heads * p.nqb_full;
    if (id < nfull_blocks) {
        head = id % p.heads; id /= p.heads;
        qblk = id % p.nqb_full; b = id / p.nqb_full;
    } else {                       // ragged last query block (only exists when nqb > nqb_full)
        id -= nfull_blocks;
        head = id % p.heads; b = id / p.heads;
        qblk = p.nqb_full;
    }
    const size_t row0 = (size_t)b * p.lpad;
    const bf16_t* Qg = p.qk + row0 * p.ld_qk + head * 64;
    const bf16_t* Kg = Qg + p.heads * 64;
    const bf16_t* Vg = p.vt + ((size_t)b * p.heads + head) * 64 * p.lpad;

    // Q fragments (B operand): query = lane & 31, d-chunk = 2 ks + half.  Rows >= lpad do not exist: clamp (never stored).
    const int q = qblk * QB + wave * 32 + l31;
    const int qld = q < p.lpad ? q : p.lpad - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Qg + (size_t)qld * p.ld_qk + (2 * ks + half) * 8);
    // a wave whose 32 queries are all padding rows only helps with staging and barriers
    const bool wave_live = qblk * QB + wave * 32 < p.L;

    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.0f;
    float m_run = -1.0e30f, l_run = 0.0f;

    const int ntiles = (p.L + KB - 1) / KB;
    const int mask_from = p.L / KB;                       // first tile that contains keys >= L (== ntiles if none)
    // staging: 512 16-byte chunks per tile, one K chunk and one V^T chunk per thread (row sr, column sc)
    const int sr = tid >> 3, sc = tid & 7;
    const bf16_t* kptr = Kg + (size_t)sr * p.ld_qk + sc * 8;
    const bf16_t* vptr = Vg + (size_t)sr * p.lpad + sc * 8;
    const int ksw = (sr >> 1) & 7;
    const int koff = sr * 128 + ((sc ^ ksw) << 4);
    // V^T row d: each 16-key group is stored as [k0-3, k8-11 | k4-7, k12-15] so that the 8 keys one lane needs for an
    // MFMA k-step (accumulator rows 4h..4h+3 and 8+4h..8+4h+3) are ONE 16-byte slot: slot 2g + h of the row.
    const int vslot = (sc >> 1) * 2;
    const int voff0 = sr * 128 + (((vslot) ^ ksw) << 4) + 8 * (sc & 1);
    const int voff1 = sr * 128 + (((vslot + 1) ^ ksw) << 4) + 8 * (sc & 1);
    char* const kring = lds;
    char* const vring = lds + 2 * KV_TILE_BYTES;
    uint4 kreg, vreg;
    const int kswz = (l31 >> 1) & 7;

    // ---- prologue: K(0), V(0) -> LDS; K(1) -> LDS; S_cur = QK(0) ----
    kreg = *reinterpret_cast<const uint4*>(kptr);
    vreg = *reinterpret_cast<const uint4*>(vptr);
    *reinterpret_cast<uint4*>(kring + koff) = kreg;
    *reinterpret_cast<uint2*>(vring + voff0) = make_uint2(vreg.x, vreg.y);
    *reinterpret_cast<uint2*>(vring + voff1) = make_uint2(vreg.z, vreg.w);
    if (ntiles > 1) {
        kreg = *reinterpret_cast<const uint4*>(kptr + (size_t)KB * p.ld_qk);
        *reinterpret_cast<uint4*>(kring + KV_TILE_BYTES + koff) = kreg;
    }
    // Retire every outstanding global load (incl. the Q fragments) HERE: left pending, hipcc's in-order vmcnt
    // bookkeeping makes each tile's first MFMAs wait for that tile's just-issued prefetch.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    f32x16 s0, s1;
    if (wave_live) {
        qk_tile(kring, qf, l31, half, kswz, s0, s1);
        if (mask_from == 0) mask_tile(s0, s1, 0, half, p.L);
    }
    __syncthreads();   // K(0) is overwritten by K(2) at the end of iteration 0: every wave must be done reading it

    for (int t = 0; t < ntiles; ++t) {
        // prefetch K(t+2) and V(t+1) into registers
        const bool pk = t + 2 < ntiles, pv = t + 1 < ntiles;
        if (pk) kreg = *reinterpret_cast<const uint4*>(kptr + (size_t)(t + 2) * KB * p.ld_qk);
        if (pv) vreg = *reinterpret_cast<const uint4*>(vptr + (size_t)(t + 1) * KB);
        f32x16 n0, n1;
        if (wave_live) {
            // ---- matrix pipe: S(t+1) ----
            if (pv) {
                qk_tile(kring + ((t + 1) & 1) * KV_TILE_BYTES, qf, l31, half, kswz, n0, n1);
                if (t + 1 >= mask_from) mask_tile(n0, n1, (t + 1) * KB, half, p.L);
            }
            // ---- VALU: online softmax of S(t), per-lane query.  Deferred rescale (T13): the running max only moves when
            //      some query of the wave outgrew it by more than RESCALE_THR (exp2 units), so P <= 2^THR and the O / l
            //      rescale is skipped on almost every tile.
            float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
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
            for (int r = 0; r < 16; ++r) {
                s0[r] = fast_exp2(__builtin_fmaf(s0[r], p.scale_log2e, -mb));
                s1[r] = fast_exp2(__builtin_fmaf(s1[r], p.scale_log2e, -mb));
                psum += s0[r] + s1[r];
            }
            l_run += psum;
            // ---- matrix pipe: O^T += V^T(t) . P^T ; k-step ks covers keys 16 ks .. 16 ks + 15 of the tile ----
            const char* vb = vring + (t & 1) * KV_TILE_BYTES + l31 * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int r0 = 8 * (ks & 1);
                union { bf16x8 v; uint32_t u[4]; } pf;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    pf.u[j] = (ks < 2) ? pack_bf2(s0[r0 + 2 * j], s0[r0 + 2 * j + 1]) : pack_bf2(s1[r0 + 2 * j], s1[r0 + 2 * j + 1]);
                const int off = ((2 * ks + half) ^ kswz) << 4;
                const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(vb + off);
                const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(vb + 32 * 128 + off);
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf.v, oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf.v, oacc[1], 0, 0, 0);
            }
            s0 = n0; s1 = n1;
        }
        // ---- publish K(t+2) (overwrites K(t), last read one iteration ago) and V(t+1) (overwrites V(t-1)) ----
        if (pk) *reinterpret_cast<uint4*>(kring + (t & 1) * KV_TILE_BYTES + koff) = kreg;
        if (pv) {
            char* vdst = vring + ((t + 1) & 1) * KV_TILE_BYTES;
            *reinterpret_cast<uint2*>(vdst + voff0) = make_uint2(vreg.x, vreg.y);
            *reinterpret_cast<uint2*>(vdst + voff1) = make_uint2(vreg.z, vreg.w);
        }
        __syncthreads();
    }

    // ---- finish: O[q, d] = O^T / l ; lane owns query q, d = db*32 + 8 (r >> 2) + 4 half + (r & 3) ----
    if (!wave_live || q >= p.lpad) return;
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
    if (!a || a->B <= 0 || a->heads <= 0 || a->L <= 0 || a->lpad < a->L || a->lpad % 128 || !a->qk || !a->vt || !a->out)
        return DGS_ERR_INVALID_ARGUMENT;
    AttnParams p;
    p.B = a->B; p.heads = a->heads; p.L = a->L; p.lpad = a->lpad; p.ld_qk = 2 * a->heads * 64;
    p.nqb_full = a->L / QB;                    // query blocks made only of valid rows
    p.nqb = (a->L + QB - 1) / QB;
    p.qk = a->qk; p.vt = a->vt; p.out = a->out;
    p.scale_log2e = a->scale * 1.44269504088896341f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(attention_fwd_kernel, dim3(a->B * a->heads * p.nqb), dim3(512), 0, st, p);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
