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
    int B, heads, L, lpad, ld_qk, k_offset, nqb, extra_unit;
    long long vt_batch_stride;
    float* lse2;
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

__device__ __forceinline__ float row_max(const f32x16& s0, const f32x16& s1) {
    float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
    return xor32_max(mx);
}

template <int V> struct Mode { static constexpr int value = V; };

__device__ __forceinline__ void mask_tile(f32x16& s0, f32x16& s1, int key0, int half, int L) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (key >= L) s0[r] = -__builtin_inff();
        if (key + 32 >= L) s1[r] = -__builtin_inff();
    }
}

// Workgroup = 256 queries of one (sample, head): 8 waves x 32 queries, two waves per SIMD, one workgroup per CU.  The
// kernel is latency-bound per wave (a wave needs ~the same time for its 65 tiles whether or not the CU is shared), so the
// work decomposition must come out in ONE round of the 256 CUs: L = 4098 = 16 * 256 + 2, and a 17th query block per head
// for the two learned-token queries would cost a whole second round.  Instead the launch uses 9-wave workgroups and the
// 9th wave of a head's last block takes the odd 32-query unit (`extra_unit`); everywhere else it exits at once.
// The grid is 1-D with head = id % heads: all query blocks of a head run on one XCD (dispatch places block b on XCD
// b % 8) and re-read that head's K / V^T (1 MiB) from its L2.
// Software pipeline (T15): while the VALU works through the softmax of tile t, the matrix pipe already runs
// S(t+1) = K(t+1) Q^T; K and V^T therefore live in two 2-deep rings that are one tile out of phase.
__global__ __launch_bounds__(576) void attention_fwd_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char lds[4 * KV_TILE_BYTES];   // K ring [2] | V^T ring [2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // ---- block -> (sample, head, query block) ----
    int id = blockIdx.x;
    const int head = id % p.heads; id /= p.heads;
    const int qblk = id % p.nqb, b = id / p.nqb;
    // 32-query unit of this wave; wave 8 exists only to take the odd unit behind the last full block
    int unit = qblk * NW + wave;
    if (wave == NW) {
        if (!(p.extra_unit && qblk == p.nqb - 1)) return;   // leaves before the first barrier
        unit = p.nqb * NW;
    }
    const size_t row0 = (size_t)b * p.lpad;
    const bf16_t* Qg = p.qk + row0 * p.ld_qk + head * 64;
    const bf16_t* Kg = Qg + p.k_offset;
    const bf16_t* Vg = p.vt + (size_t)b * p.vt_batch_stride + (size_t)head * 64 * p.lpad;

    // Q fragments (B operand): query = lane & 31, d-chunk = 2 ks + half.  Rows >= lpad do not exist: clamp (never stored).
    const int q = unit * 32 + l31;
    const int qld = q < p.lpad ? q : p.lpad - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Qg + (size_t)qld * p.ld_qk + (2 * ks + half) * 8);
    // a wave whose 32 queries are all padding rows only helps with staging and barriers
    const bool wave_live = unit * 32 < p.L;

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
    const bool stager = wave < NW;        // the 512 threads of waves 0..7 move the tiles
    if (stager) {
        kreg = *reinterpret_cast<const uint4*>(kptr);
        vreg = *reinterpret_cast<const uint4*>(vptr);
        *reinterpret_cast<uint4*>(kring + koff) = kreg;
        *reinterpret_cast<uint2*>(vring + voff0) = make_uint2(vreg.x, vreg.y);
        *reinterpret_cast<uint2*>(vring + voff1) = make_uint2(vreg.z, vreg.w);
        if (ntiles > 1) {
            kreg = *reinterpret_cast<const uint4*>(kptr + (size_t)KB * p.ld_qk);
            *reinterpret_cast<uint4*>(kring + KV_TILE_BYTES + koff) = kreg;
        }
    }
    // Retire every outstanding global load (incl. the Q fragments) HERE: left pending, hipcc's in-order vmcnt
    // bookkeeping makes each tile's first MFMAs wait for that tile's just-issued prefetch.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    // ---- first tile: S(0), its row max, and the initial running max ----
    f32x16 s0, s1;
    float mx = -1.0e30f;
    if (wave_live) {
        qk_tile(kring, qf, l31, half, kswz, s0, s1);
        if (mask_from == 0) mask_tile(s0, s1, 0, half, p.L);
        mx = row_max(s0, s1);
        m_run = mx;                          // first tile always "rescales" (O and l are still zero)
    }
    __syncthreads();   // K(0) is overwritten by K(2) at the end of iteration 0: every wave must be done reading it

    // One iteration = one 64-key tile t, in two MFMA||VALU phases that live in ONE basic block each:
    //   phase A   matrix pipe: S(t+1) = K(t+1) Q^T        VALU: P(t) = exp2(S(t) c - m c), row sums, bf16 packing
    //   phase B   matrix pipe: O^T += V^T(t) P^T(t)       VALU: row max of S(t+1)
    // then the (rare, wave-uniform) deferred rescale for tile t+1 -- after ALL of P(t) V(t) has been issued (T13 hazard) --
    // and the staging writes + barrier.  MODE 0: steady state, 1: S(t+1) is the ragged last tile (masked), 2: last tile
    // (no S(t+1)).  Three straight-line copies instead of in-loop branches keep every phase a single scheduling region.
    auto iteration = [&](int t, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const bool pk = stager && t + 2 < ntiles, pvs = stager && MODE != 2;
        if (pk) kreg = *reinterpret_cast<const uint4*>(kptr + (size_t)(t + 2) * KB * p.ld_qk);
        if (pvs) vreg = *reinterpret_cast<const uint4*>(vptr + (size_t)(t + 1) * KB);
        if (wave_live) {
            f32x16 n0, n1;
            const float mb = m_run * p.scale_log2e;
            // ---------------- phase A ----------------
            if (MODE != 2) qk_tile(kring + ((t + 1) & 1) * KV_TILE_BYTES, qf, l31, half, kswz, n0, n1);
            float psum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = fast_exp2(__builtin_fmaf(s0[r], p.scale_log2e, -mb));
                s1[r] = fast_exp2(__builtin_fmaf(s1[r], p.scale_log2e, -mb));
                psum += s0[r] + s1[r];
            }
            l_run += psum;
            union { bf16x8 v; uint32_t u[4]; } pf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 8 * (ks & 1) + 2 * j;
                    pf[ks].u[j] = (ks < 2) ? pack_bf2(s0[r], s0[r + 1]) : pack_bf2(s1[r], s1[r + 1]);
                }
#ifndef HIPEMU
            if (MODE != 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {          // 8 x { 1 LDS read, 1 MFMA, 14 VALU }
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);
                }
            }
#endif
            if (MODE == 1) mask_tile(n0, n1, (t + 1) * KB, half, p.L);
            // ---------------- phase B ----------------
            const char* vb = vring + (t & 1) * KV_TILE_BYTES + l31 * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = ((2 * ks + half) ^ kswz) << 4;
                const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(vb + off);
                const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(vb + 32 * 128 + off);
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf[ks].v, oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf[ks].v, oacc[1], 0, 0, 0);
            }
            if (MODE != 2) {
                mx = row_max(n0, n1);
#ifndef HIPEMU
#pragma unroll
                for (int i = 0; i < 8; ++i) {          // 8 x { 1 LDS read, 1 MFMA, 3 VALU }
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 1);
                }
#endif
                // deferred rescale (T13) for tile t+1: the running max only moves when some query of the wave outgrew it by
                // more than RESCALE_THR (exp2 units), so P <= 2^THR and the O / l rescale is skipped on almost every tile
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
                s0 = n0; s1 = n1;
            }
        }
        // ---- publish K(t+2) (overwrites K(t), last read one iteration ago) and V(t+1) (overwrites V(t-1)) ----
        if (pk) *reinterpret_cast<uint4*>(kring + (t & 1) * KV_TILE_BYTES + koff) = kreg;
        if (pvs) {
            char* vdst = vring + ((t + 1) & 1) * KV_TILE_BYTES;
            *reinterpret_cast<uint2*>(vdst + voff0) = make_uint2(vreg.x, vreg.y);
            *reinterpret_cast<uint2*>(vdst + voff1) = make_uint2(vreg.z, vreg.w);
        }
        if (MODE != 2) __syncthreads();
    };
    const bool ragged = mask_from < ntiles && ntiles > 1;       // the last tile holds keys >= L
    const int steady_end = ragged ? ntiles - 2 : ntiles - 1;    // iterations [0, steady_end) use MODE 0
    for (int t = 0; t < steady_end; ++t) iteration(t, Mode<0>{});
    if (ragged) iteration(ntiles - 2, Mode<1>{});
    iteration(ntiles - 1, Mode<2>{});

    // ---- finish: O[q, d] = O^T / l ; lane owns query q, d = db*32 + 8 (r >> 2) + 4 half + (r & 3) ----
    if (!wave_live || q >= p.lpad) return;
    const float l_tot = xor32_sum(l_run);
    const float inv = 1.0f / l_tot;
    if (p.lse2 && half == 0) p.lse2[((size_t)b * p.heads + head) * p.lpad + q] = m_run * p.scale_log2e + log2f(l_tot);
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
    p.B = a->B; p.heads = a->heads; p.L = a->L; p.lpad = a->lpad;
    p.ld_qk = a->ld_qk > 0 ? a->ld_qk : 2 * a->heads * 64;
    p.k_offset = a->k_offset > 0 ? a->k_offset : a->heads * 64;
    p.vt_batch_stride = a->vt_batch_stride > 0 ? a->vt_batch_stride : (long long)a->heads * 64 * a->lpad;
    p.lse2 = a->lse2;
    const int units = (a->L + 31) / 32;        // 32-query wave units
    p.extra_unit = (units % NW == 1 && units > 1) ? 1 : 0;
    p.nqb = p.extra_unit ? units / NW : (units + NW - 1) / NW;
    p.qk = a->qk; p.vt = a->vt; p.out = a->out;
    p.scale_log2e = a->scale * 1.44269504088896341f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(attention_fwd_kernel, dim3(a->B * a->heads * p.nqb), dim3(576), 0, st, p);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
