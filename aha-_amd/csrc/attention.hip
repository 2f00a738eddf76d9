// Attention for both halves of the path, one kernel template:
//   LM  : T new tokens x G query heads per KV head (GQA) against a stream's KV cache that is
//         addressed through the ring mapping of StreamStep (no per-step cache copy; the
//         reference re-allocates and copies the whole cache every step, test/sink_cache.py:134-162).
//         Visibility rule: key j visible to new token t iff j <= causal_off + t.
//   ViT : dense non-causal attention over the Np patch tokens of one frame.
//
// Structure (gfx950, wave64):  a workgroup = 4 waves owns one (kv head, 64-row group, key split);
// each wave owns ONE 16-row query tile.  Per 64-key block the workgroup stages K (row-major,
// 16-B chunks XOR-swizzled => conflict-free ds_read_b128) and V (row-major, rows padded by 32 B) in LDS, then
// each wave computes  S^T = K * Q^T  with v_mfma_f32_16x16x32_bf16 (A = K rows, B = Q rows kept
// in registers), so a lane holds scores of ONE query row (lane&15) for 4 consecutive keys per
// key tile: the softmax row reduction is in-register + two xor-shuffles (16, 32), and the
// exponentiated tile, packed to bf16, IS the B operand of  O^T += V^T * P^T  with no LDS round
// trip (the k-slot permutation is absorbed by fetching the V^T fragment as two transposed 4-key reads,
// ds_read_b64_tr_b16).
// Online softmax in fp32; P rounded to bf16 before PV (flash/sdpa semantics).  With more than
// one key split the kernel writes (m, l, unnormalised O) partials and attn_combine merges them.
#include "aha_kernels.h"
#include <type_traits>

// LM mode: where a vector of <= 8 channels starting at channel `col` (a multiple of its width) of new token t of stream b goes - row-major
// [B*T][ldo], or k-blocked [ldo/32][okb rows][32] for the mid-M o_proj GEMM, whose LDS-DMA then pulls contiguous 1-KiB panels (gemm_wl.hip)
static __device__ __forceinline__ bf16* lm_out_ptr(const AttnArgs& a, const int b, const int t, const int col) {
    if (a.okb) return a.out + ((long)(col >> 5) * a.okb + (long)b * a.T + t) * 32 + (col & 31);
    return a.out + b * a.o_bs + (long)t * a.ldo + col;
}


// acc *= a as four single v_mul_f32.  Left as a vector multiply, hipcc makes v_pk_mul_f32 of it, and next to MFMAs a packed f32 instruction holds
// the SIMD's issue port ~4x as long as the two scalar ones it replaces (MI355X_MICROARCH.md, per-instruction constants) - the dense
// attention loops are issue-bound (profiles/r05_pmc_attn_dense.txt).  The empty asm makes each product opaque, so nothing re-packs them.
static __device__ __forceinline__ void scale_acc(f32x4& acc, const float a) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = acc[e] * a;
        asm("" : "+v"(t));
        acc[e] = t;
    }
}

template <int D> struct AttnCfg {
    static constexpr int CPR = D / 8;          // 16-B chunks per K row
    static constexpr int VST = D + 16;         // V row stride in elements: 2*D + 32 bytes, so the 8 rows a 32-lane half
                                               // touches in one transposed read start 8 banks apart (conflict-free)
    static constexpr int DT = D / 16;          // d-tiles of the output
    static constexpr int KSQ = D / 32;         // k-steps of the QK^T product
    // LDS slot of 16-B chunk ch of key row `row`: the 16 rows one ds_read_b128 quarter-wave reads must land on 16 different
    // 16-byte bank groups.  Power-of-two rows (128 / 256 B) fold the row into the chunk; 192-byte rows (D = 96) already step
    // 64 B per row mod 256, so only rows 4 apart collide and the XOR rotates within a group of four chunks.
    static __device__ __forceinline__ int kslot(int row, int ch) {
        if constexpr ((CPR & (CPR - 1)) == 0) return ch ^ (row & (CPR - 1));
        else return ch ^ ((row >> 2) & 3);
    }
};

template <int D, bool LM>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a, const StepDesc* __restrict__ sdp) {
    using C = AttnCfg<D>;
    __shared__ __attribute__((aligned(16))) bf16 Ks[64 * D];
    __shared__ __attribute__((aligned(16))) bf16 Vs[64 * C::VST];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q4 = lane >> 4, r16 = lane & 15;
    const int b = blockIdx.z;
    const int R = a.G * a.T, RT = ceil_div(R, 16), RG = ceil_div(RT, 4);
    const int hk = blockIdx.y / RG, rg = blockIdx.y % RG;
    const int split = blockIdx.x;

    int Lk, off;
    const bf16 *kb, *vb;
    int ldk;
    StreamStep ss;
    if constexpr (LM) {
        ss = sdp->s[b];                                     // device-resident step descriptor (uploaded once per step)
        Lk = ss.len_after;
        off = ss.causal_off;
        const long lo = ((long)a.layer * a.Hkv + hk) * ss.cap * a.hd;
        kb = ss.k_base + lo;
        vb = ss.v_base + lo;
        ldk = a.hd;
    } else {
        Lk = a.Lk;
        off = 1 << 30;
        kb = a.k + b * a.kv_bs + hk * a.hd;
        vb = a.v + b * a.kv_bs + hk * a.hd;
        ldk = a.ldk;
    }
    const int j0 = split * a.split_len;
    const int j1 = min(Lk, j0 + a.split_len);
    if (j0 >= j1) return;                                   // uniform per block

    const int rt = rg * 4 + wave;
    const bool wave_on = rt < RT;
    int r = rt * 16 + r16;
    const bool row_ok = wave_on && r < R;
    if (r > R - 1) r = R - 1;
    const int g = r / a.T, t = r % a.T;
    const int head = hk * a.G + g;

    // Q fragments (B operand): Q[row][ks*32 + 8*q4 .. +7]
    bf16x8 qf[C::KSQ];
    if (LM && a.q_partial) {
        // Frozen-static steps: build Q straight from the QKV GEMM's split-K slabs - reduce + bias ->
        // bf16, RoPE (modeling_qwen2.py:107-131) with the pair (d, d + D/2) held by the same lane in
        // qf[ks] / qf[ks + KSQ/2] - so no qkv_finish launch and no q_rot round trip.  Same rounding
        // points as qkv_finish_kernel (bit-identical Q).
        const int m = b * a.T + t;
        const int pos = min(ss.pos_base + t, a.n_pos - 1);
        float xq[C::KSQ][8];
#pragma unroll
        for (int ks = 0; ks < C::KSQ; ++ks) {
            const int col = head * D + ks * 32 + q4 * 8;
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            const float* p = a.q_partial + (long)m * a.q_ldp + col;
            for (int s0 = 0; s0 < a.q_S; s0 += 8) {
                f32x4 t0[8], t1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (s0 + j < a.q_S) {
                        t0[j] = *reinterpret_cast<const f32x4*>(p + (s0 + j) * a.q_slab_stride);
                        t1[j] = *reinterpret_cast<const f32x4*>(p + (s0 + j) * a.q_slab_stride + 4);
                    }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (s0 + j < a.q_S) { a0 += t0[j]; a1 += t1[j]; }
            }
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(a.q_bias + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xq[ks][e] = rbf(a0[e] + bf2f(bv[e]));
                xq[ks][4 + e] = rbf(a1[e] + bf2f(bv[4 + e]));
            }
        }
        constexpr int HK = C::KSQ / 2;
#pragma unroll
        for (int ks = 0; ks < HK; ++ks) {
            const int d = ks * 32 + q4 * 8;
            const bf16x8 c1 = *reinterpret_cast<const bf16x8*>(a.rope_cos + (long)pos * D + d);
            const bf16x8 s1 = *reinterpret_cast<const bf16x8*>(a.rope_sin + (long)pos * D + d);
            const bf16x8 c2 = *reinterpret_cast<const bf16x8*>(a.rope_cos + (long)pos * D + d + D / 2);
            const bf16x8 s2 = *reinterpret_cast<const bf16x8*>(a.rope_sin + (long)pos * D + d + D / 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x1 = xq[ks][e], x2 = xq[ks + HK][e];
                qf[ks][e] = f2bf(rbf(x1 * bf2f(c1[e])) + rbf(-x2 * bf2f(s1[e])));
                qf[ks + HK][e] = f2bf(rbf(x2 * bf2f(c2[e])) + rbf(x1 * bf2f(s2[e])));
            }
        }
    } else {
        // channels >= a.hd (zero padding of a head dim that is not a template size, e.g. so400m's 72)
        // are loaded from a clamped address and zeroed by a select: no guarded loads
        const bf16* qp = a.q + b * a.q_bs + (long)t * a.ldq + head * a.hd;
        const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < C::KSQ; ++ks) {
            const int d0 = ks * 32 + q4 * 8;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(qp + min(d0, a.hd - 8));
            qf[ks] = d0 < a.hd ? v : z8;
        }
    }

    f32x4 o[C::DT];
#pragma unroll
    for (int i = 0; i < C::DT; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;                         // m_run lives in the base-2 domain: c2 * max(score)
    const float c2 = a.scale * 1.4426950408889634f;              // scale * log2(e)

    // K/V staging is issue-early / write-late with TWO key blocks in flight: register sets A and B hold blocks
    // jb+64 and jb+128 while block jb computes, so a workgroup's chain of key blocks (9 for one ViT frame, where
    // only 144 workgroups exist and nothing else hides the latency) no longer exposes one load latency per block.
    // Loads are unconditional with clamped indices (finite data; masked below / zeroed at the store), the loop is
    // unrolled by two so each set is statically indexed (counted vmcnt), and a trailing block that lies wholly past
    // j1 is an exact no-op of the online softmax (all scores -inf: alpha = 1, p = 0).
    constexpr int NCH = (64 * C::CPR) / 256;
    bf16x8 kA[NCH], vA[NCH], kB[NCH], vB[NCH];
    auto gload = [&](bf16x8 (&kreg)[NCH], bf16x8 (&vreg)[NCH], int jb) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C::CPR, ch = idx % C::CPR;
            int j = jb + row; if (j > j1 - 1) j = j1 - 1;
            int slot = j;
            if constexpr (LM) slot = phys_slot(ss, j);
            const int dch = min(ch * 8, a.hd - 8);
            kreg[i] = *reinterpret_cast<const bf16x8*>(kb + (long)slot * ldk + dch);
            vreg[i] = *reinterpret_cast<const bf16x8*>(vb + (long)slot * ldk + dch);
        }
    };
    auto block = [&](bf16x8 (&kreg)[NCH], bf16x8 (&vreg)[NCH], int jb) {
        __syncthreads();                                    // previous block's LDS reads done
        // ---- stage K (swizzled rows) and V (transposed) for keys jb .. jb+63 from the prefetched registers
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C::CPR, ch = idx % C::CPR;
            const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
            const bool pad = ch * 8 >= a.hd;
            const bf16x8 kv = pad ? z8 : kreg[i], vv = pad ? z8 : vreg[i];
            *reinterpret_cast<bf16x8*>(&Ks[row * D + ((ch ^ (row & (C::CPR - 1))) << 3)]) = kv;
            *reinterpret_cast<bf16x8*>(&Vs[row * C::VST + ch * 8]) = vv;
        }
        __syncthreads();
        gload(kreg, vreg, jb + 128);                        // refill this set two blocks ahead
        if (!wave_on) return;

        // ---- S^T tiles: acc[kt][e] = score(key jb + kt*16 + 4*q4 + e, row r)
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = kt * 16 + r16;
#pragma unroll
            for (int ks = 0; ks < C::KSQ; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(
                    &Ks[row * D + (((ks * 4 + q4) ^ (row & (C::CPR - 1))) << 3)]);
                s[kt] = mfma16(kf, qf[ks], s[kt]);
            }
        }
        // ---- mask, block max, probabilities: the base-2 online softmax of attn_lm_kernel, operation for operation, so that a
        // row gets the same bits from either kernel (a step's kernel is chosen by its shape; batched, solo and last-token-only
        // evaluations of a row must agree exactly)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = jb + kt * 16 + 4 * q4 + e;
                if (!((j < j1) && (j <= off + t))) s[kt][e] = -INFINITY;
            }
        float bmax = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
        for (int kt = 1; kt < 4; ++kt) bmax = fmaxf(bmax, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
        bmax = fmaxf(bmax, __shfl_xor(bmax, 16, 64));
        bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
        const float m_new = fmaxf(m_run, bmax * c2);             // -inf * c2 = -inf
        float alpha = 1.f, psum = 0.f;
        bf16x8 pb[2];
        if (m_new == -INFINITY) {                            // nothing visible yet for this row
            pb[0] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            pb[1] = pb[0];
        } else {
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // m_run = -inf -> 0
            const float nm = -m_new;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][e], c2, nm));   // masked: 2^-inf = 0
                    psum += p;
                    pb[kt >> 1][(kt & 1) * 4 + e] = f2bf(p);
                }
        }
        m_run = m_new;
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < C::DT; ++i) o[i] *= alpha;
        // ---- O^T += V^T * P^T ; k-slot (q4, e): e<4 -> key (2kp)*16+4q4+e, e>=4 -> key (2kp+1)*16+4q4+e-4.
        // V sits row-major in LDS; ds_read_b64_tr_b16 hands lane r16 of each 16-lane group column dt*16+r16 of the
        // group's 4-key block (lane 4q+p supplies the address of block row q, columns 4p..4p+3), i.e. exactly the
        // V^T fragment, with no transposing store.  All 64 lanes are active here (wave_on is per wave).
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        const int vrow = 4 * q4 + (r16 >> 2), vcol = 4 * (r16 & 3);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4)(&Vs[((2 * kp) * 16 + vrow) * C::VST + dt * 16 + vcol]));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4)(&Vs[((2 * kp + 1) * 16 + vrow) * C::VST + dt * 16 + vcol]));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                o[dt] = mfma16(vf, pb[kp], o[dt]);
            }
        }
    };
    gload(kA, vA, j0);
    __builtin_amdgcn_sched_barrier(0);                      // keep set A's loads older than set B's (counted vmcnt at the loop top)
    gload(kB, vB, j0 + 64);
    __builtin_amdgcn_sched_barrier(0);
    for (int jb = j0; jb < j1; jb += 128) {
        block(kA, vA, jb);
        block(kB, vB, jb + 64);
    }
    if (!wave_on) return;
    // row sum across the 4 lanes that share this query row
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (!row_ok) return;

    // o[dt][e] <-> d = dt*16 + 4*q4 + e of row r
    if (a.n_splits == 1) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;        // a row that sees no key (flash-attn-2 mask of a frozen static cache) gives 0
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            bf16x4 ov = {f2bf(o[dt][0] * inv), f2bf(o[dt][1] * inv), f2bf(o[dt][2] * inv), f2bf(o[dt][3] * inv)};
            if (dt * 16 + 4 * q4 < a.hd) *reinterpret_cast<bf16x4*>(lm_out_ptr(a, b, t, head * a.hd + dt * 16 + 4 * q4)) = ov;
        }
    } else {
        const int Rpad = RT * 16;
        const long prow = (((long)b * a.Hkv + hk) * a.n_splits + split) * Rpad + (rt * 16 + r16);
        float* po = a.part_o + prow * D + 4 * q4;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) *reinterpret_cast<f32x4*>(po + dt * 16) = o[dt];
        if (q4 == 0) {
            a.part_ml[prow * 2] = m_run;
            a.part_ml[prow * 2 + 1] = l_run;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Dense (vision tower) attention: the same S^T = K Q^T / online softmax / O^T += V^T P^T structure per query row as
// attn_fwd_kernel, without the LM's cache addressing, with the softmax in the base-2 domain (one FMA + one v_exp_f32
// per probability, mask only in the tail block), and with TPW 16-row query tiles per wave (a workgroup covers 64*TPW
// rows of one head; TPW > 1 stages 1/TPW of the K/V bytes and shares every V^T fragment read between TPW MFMAs).
// A row's instruction sequence does not depend on TPW, so all settings are bit-identical; see launch_attn for why
// TPW = 1 is the default.
// ---------------------------------------------------------------------------------------------
template <int D, int TPW>
static __device__ __forceinline__ void attn_dense_body(const AttnArgs& a, const int by, const int b, bf16* Ks, bf16* Vs) {
    using C = AttnCfg<D>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q4 = lane >> 4, r16 = lane & 15;
    const int R = a.G * a.T, RT = ceil_div(R, 16), RG = ceil_div(RT, 4 * TPW);
    const int hk = by / RG, rg = by % RG;
    const int Lk = a.Lk, ldk = a.ldk;
    const bf16* kb = a.k + b * a.kv_bs + hk * a.hd;
    const bf16* vb = a.v + b * a.kv_bs + hk * a.hd;
    const int j1 = Lk;
    if (j1 <= 0) return;

    const int rt0 = (rg * 4 + wave) * TPW;
    const bool wave_on = rt0 < RT;
    bool row_ok[TPW];
    int trow[TPW], thead[TPW];
    bf16x8 qf[TPW][C::KSQ];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        int r = (rt0 + tt) * 16 + r16;
        row_ok[tt] = (rt0 + tt) < RT && r < R;
        if (r > R - 1) r = R - 1;
        trow[tt] = r % a.T;
        thead[tt] = hk * a.G + r / a.T;
        // channels >= a.hd (zero padding of a head dim that is not a template size) come from a clamped address
        const bf16* qp = a.q + b * a.q_bs + (long)trow[tt] * a.ldq + thead[tt] * a.hd;
        const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < C::KSQ; ++ks) {
            const int d0 = ks * 32 + q4 * 8;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(qp + min(d0, a.hd - 8));
            qf[tt][ks] = d0 < a.hd ? v : z8;
        }
    }

    f32x4 o[TPW][C::DT];
    float m_run[TPW], l_run[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        m_run[tt] = -INFINITY;
        l_run[tt] = 0.f;
#pragma unroll
        for (int i = 0; i < C::DT; ++i) o[tt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // K/V staging: issue-early / write-late, two key blocks in flight (see attn_fwd_kernel)
    constexpr int NCH = (64 * C::CPR) / 256;
    bf16x8 kA[NCH], vA[NCH], kB[NCH], vB[NCH];
    auto gload = [&](bf16x8 (&kreg)[NCH], bf16x8 (&vreg)[NCH], int jb) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C::CPR, ch = idx % C::CPR;
            int j = jb + row; if (j > j1 - 1) j = j1 - 1;
            const int dch = min(ch * 8, a.hd - 8);
            kreg[i] = *reinterpret_cast<const bf16x8*>(kb + (long)j * ldk + dch);
            vreg[i] = *reinterpret_cast<const bf16x8*>(vb + (long)j * ldk + dch);
        }
    };
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    const int vrow = 4 * q4 + (r16 >> 2), vcol = 4 * (r16 & 3);
    const float c2 = a.scale * 1.4426950408889634f;                 // scale * log2(e)
    auto block = [&](bf16x8 (&kreg)[NCH], bf16x8 (&vreg)[NCH], int jb) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C::CPR, ch = idx % C::CPR;
            const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
            const bool pad = ch * 8 >= a.hd;
            const bf16x8 kv = pad ? z8 : kreg[i], vv = pad ? z8 : vreg[i];
            *reinterpret_cast<bf16x8*>(&Ks[row * D + (C::kslot(row, ch) << 3)]) = kv;
            *reinterpret_cast<bf16x8*>(&Vs[row * C::VST + ch * 8]) = vv;
        }
        __syncthreads();
        gload(kreg, vreg, jb + 128);
        if (!wave_on) return;

        // ---- per tile: online softmax in the base-2 domain.  softmax(scale*s) = 2^(c*s - c*max) / sum with
        // c = scale*log2(e) > 0, so the running max is kept as c*max(s) and every probability costs one FMA and one
        // v_exp_f32; keys past the end exist only in the last block, so only that block pays for the mask.
        // One tile at a time (its 16 score registers die before the next tile starts; the K fragments are re-read from
        // LDS per tile - LDS has the bandwidth, the register file does not: keeping all tiles' scores live cost 240
        // AGPR moves per block).
        bf16x8 pb[TPW][2];
        const bool tail = jb + 64 > j1;                              // uniform
        const int lim = j1 - jb - 4 * q4;                            // key kt*16 + e of this lane's quad exists iff kt*16 + e < lim
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            f32x4 s[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const int row = kt * 16 + r16;
#pragma unroll
                for (int ks = 0; ks < C::KSQ; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[row * D + (C::kslot(row, ks * 4 + q4) << 3)]);
                    s[kt] = mfma16(kf, qf[tt][ks], s[kt]);
                }
            }
            if (tail) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kt * 16 + e >= lim) s[kt][e] = -INFINITY;
            }
            float bmax = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt) bmax = fmaxf(bmax, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
            bmax = fmaxf(bmax, __shfl_xor(bmax, 16, 64));
            bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
            const float m_new = fmaxf(m_run[tt], bmax * c2);         // -inf * c2 = -inf
            // a row that has seen no key yet (m_new = -inf: only while every key so far was masked) takes alpha = 1 and the offset 0, so
            // that its probabilities are 2^(-inf) = 0 without a branch (the branch cost eleven register presets per tile in every block)
            const bool unseen = m_new == -INFINITY;
            // (the difference is taken on a sanitised operand: -inf - -inf of an unseen row would be a NaN on the unselected arm, which this file's
            // -fno-honor-nans lets the compiler treat as poison - ADVICE r5)
            const float alpha = unseen ? 1.f : __builtin_amdgcn_exp2f((unseen ? 0.f : m_run[tt]) - m_new);   // m_run = -inf, m_new finite -> 0
            const float nm = unseen ? 0.f : -m_new;
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][e], c2, nm));   // masked: 2^-inf = 0
                    psum += p;
                    pb[tt][kt >> 1][(kt & 1) * 4 + e] = f2bf(p);
                }
            m_run[tt] = m_new;
            l_run[tt] = l_run[tt] * alpha + psum;
#pragma unroll
            for (int i = 0; i < C::DT; ++i) scale_acc(o[tt][i], alpha);
            if (TPW > 1) __builtin_amdgcn_sched_barrier(0);          // keep the tiles sequential (hipcc would re-merge them)
        }
        // ---- O^T += V^T P^T: one transposed V fragment feeds TPW MFMAs
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4)(&Vs[((2 * kp) * 16 + vrow) * C::VST + dt * 16 + vcol]));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4)(&Vs[((2 * kp + 1) * 16 + vrow) * C::VST + dt * 16 + vcol]));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) o[tt][dt] = mfma16(vf, pb[tt][kp], o[tt][dt]);
            }
        }
    };
    gload(kA, vA, 0);
    __builtin_amdgcn_sched_barrier(0);
    gload(kB, vB, 64);
    __builtin_amdgcn_sched_barrier(0);
    for (int jb = 0; jb < j1; jb += 128) {
        block(kA, vA, jb);
        block(kB, vB, jb + 64);
    }
    if (!wave_on) return;
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        float l = l_run[tt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (!row_ok[tt]) continue;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        bf16* op = a.out + b * a.o_bs + (long)trow[tt] * a.ldo + thead[tt] * a.hd + 4 * q4;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            bf16x4 ov = {f2bf(o[tt][dt][0] * inv), f2bf(o[tt][dt][1] * inv), f2bf(o[tt][dt][2] * inv), f2bf(o[tt][dt][3] * inv)};
            if (dt * 16 + 4 * q4 < a.hd) *reinterpret_cast<bf16x4*>(op + dt * 16) = ov;
        }
    }
}

template <int D, int TPW>
__global__ __launch_bounds__(256, (TPW == 1 && D <= 64) ? 4 : 1) void attn_dense_kernel(AttnArgs a) {      // D = 64: four workgroups per CU (<= 128 VGPRs)
    using C = AttnCfg<D>;
    __shared__ __attribute__((aligned(16))) bf16 Ks[64 * D];
    __shared__ __attribute__((aligned(16))) bf16 Vs[64 * C::VST];
    attn_dense_body<D, TPW>(a, blockIdx.y, blockIdx.z, Ks, Vs);
}
// Butterfly steps over lanes l ^ 16 and l ^ 32 on the vector ALU (v_permlane16_swap / v_permlane32_swap, gfx950) instead of
// ds_bpermute round trips through the LDS: swap(x, x) leaves [r0 r0 r2 r2] / [r1 r1 r3 r3] (16-lane rows) resp. [lo lo] / [hi hi]
// (32-lane halves) in the two results, so one max / add of the pair is the xor-16 / xor-32 reduction step in every lane.  max is
// exact and the adds are commutative: same bits as the __shfl_xor form.
static __device__ __forceinline__ float xor16_max(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
static __device__ __forceinline__ float xor32_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
static __device__ __forceinline__ float xor16_sum(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
static __device__ __forceinline__ float xor32_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---------------------------------------------------------------------------------------------
// Dense attention with the WHOLE head LDS-resident (round 3): at 576 keys x 64 channels the K and V of one (frame, head) are
// 2 x 72 KB - they fit the CU's 160 KB.  One workgroup of NW waves owns one (frame, head): K and V go into LDS ONCE by LDS-DMA
// (global_load_lds_dwordx4, swizzles on the source address), and after one barrier every wave runs its TPW query tiles over
// all key blocks with no synchronisation at all - attn_dense_kernel restages every 64-key block through registers for each
// 64-row group (9 x per head) behind two barriers per block and measured 108 us per 32-frame layer, three times its VALU
// time.
// LDS images, 128-byte rows (8 chunks of 16 B), lane-linear DMA destinations:
//   K: chunk c of key r at slot c ^ (r & 7)             -> conflict-free ds_read_b128 A fragments (as attn_dense_kernel)
//   V: chunk c of key r at slot c ^ (2 * ((r >> 1) & 3)) -> conflict-free ds_read_b64_tr_b16: a 32-lane half reads 8 rows x 32 B;
//      rows are 32 banks apart, so rows r and r+2 would collide - the XOR moves each row pair to its own 32-byte column.
// A row's arithmetic is instruction for instruction that of attn_dense_kernel (same 64-key blocks, same online softmax in the
// base-2 domain, same MFMA order): bit-identical results, so the choice of kernel (which depends on the geometry only, never
// on the batch) cannot change an embedding.
// ---------------------------------------------------------------------------------------------
template <int NW, int TPW>
__global__ __launch_bounds__(64 * NW) void attn_head64_kernel(AttnArgs a, int lk_pad) {
    constexpr int D = 64;
    using C = AttnCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char dsm_raw[];
    bf16* Ks = reinterpret_cast<bf16*>(dsm_raw);
    bf16* Vs = Ks + (long)lk_pad * D;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, r16 = lane & 15;
    const int b = blockIdx.y, hk = blockIdx.x, rg = blockIdx.z;      // rg: row group of NW * TPW query tiles (latency path: several workgroups per head)
    const int R = a.G * a.T, RT = ceil_div(R, 16);
    const int Lk = a.Lk, ldk = a.ldk;
    const bf16* kb = a.k + b * a.kv_bs + hk * D;
    const bf16* vb = a.v + b * a.kv_bs + hk * D;
    const int j1 = Lk;

    // ---- K then V into LDS: piece = 8 key rows (1 KiB); every wave issues the same number of pieces per operand (surplus
    // ones repeat the last piece: same bytes to the same place), so the counted wait below is the same for all waves
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int np = lk_pad >> 3, pw = ceil_div(np, NW);
    {
        const int lr = lane >> 3, sl = lane & 7;
        for (int i = 0; i < pw; ++i) {
            const int pc = min(wave + i * NW, np - 1), row = pc * 8 + lr;
            const bf16* src = kb + (long)min(row, j1 - 1) * ldk + ((sl ^ (row & 7)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ks + pc * 512), 16, 0, 0);
        }
        for (int i = 0; i < pw; ++i) {
            const int pc = min(wave + i * NW, np - 1), row = pc * 8 + lr;
            const bf16* src = vb + (long)min(row, j1 - 1) * ldk + ((sl ^ (2 * ((row >> 1) & 3))) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Vs + pc * 512), 16, 0, 0);
        }
    }

    const int rt0 = (rg * NW + wave) * TPW;
    const bool wave_on = rt0 < RT;
    bool row_ok[TPW];
    int trow[TPW], thead[TPW];
    bf16x8 qf[TPW][C::KSQ];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        int r = (rt0 + tt) * 16 + r16;
        row_ok[tt] = (rt0 + tt) < RT && r < R;
        if (r > R - 1) r = R - 1;
        trow[tt] = r % a.T;
        thead[tt] = hk * a.G + r / a.T;
        const bf16* qp = a.q + b * a.q_bs + (long)trow[tt] * a.ldq + thead[tt] * D;
#pragma unroll
        for (int ks = 0; ks < C::KSQ; ++ks) qf[tt][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32 + q4 * 8);
    }
    f32x4 o[TPW][C::DT];
    float m_run[TPW], l_run[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        m_run[tt] = -INFINITY;
        l_run[tt] = 0.f;
#pragma unroll
        for (int i = 0; i < C::DT; ++i) o[tt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    const int vrow = 4 * q4 + (r16 >> 2), vcol = 4 * (r16 & 3);
    const float c2 = a.scale * 1.4426950408889634f;                 // scale * log2(e)

    // Everything this wave issued has landed (the Q fragments are ordinary loads, the youngest entries of its queue: hipcc
    // waits vmcnt(0) at their first use anyway while LDS-DMA is in flight), then every wave's pieces have
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // One 64-key block for the wave's TPW tiles.  TAIL (only the last block of a key count that is not a multiple of 64) masks the keys
    // past the end and handles rows that have seen no key yet; the full-block form has neither: every score is finite, so m_new is, and
    // the 16 key-index compares, the masked-row branch and its register presets (~60 vector instructions per tile that the one-loop
    // form executed or set up in EVERY block) are gone.  Same operations in the same order on every value: same bits as before and as
    // attn_dense_kernel.
    auto block = [&](const int jb, auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
        bf16x8 pb[TPW][2];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            f32x4 s[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const int row = jb + kt * 16 + r16;
#pragma unroll
                for (int ks = 0; ks < C::KSQ; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[row * D + (((ks * 4 + q4) ^ (r16 & 7)) << 3)]);   // row & 7 == r16 & 7
                    s[kt] = mfma16(kf, qf[tt][ks], s[kt]);
                }
            }
            if constexpr (TAIL) {
                const int lim = j1 - jb - 4 * q4;                    // key kt*16 + e of this lane's quad exists iff kt*16 + e < lim
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kt * 16 + e >= lim) s[kt][e] = -INFINITY;
            }
            float bmax = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt) bmax = fmaxf(bmax, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
            bmax = xor16_max(bmax);                                  // lanes l, l^16, l^32, l^48 hold one query row's keys
            bmax = xor32_max(bmax);
            const float m_new = fmaxf(m_run[tt], bmax * c2);         // -inf * c2 = -inf
            // a row that has seen no key yet (m_new = -inf: only while every key so far was masked) takes alpha = 1 and the offset 0, so
            // that its probabilities are 2^(-inf) = 0 without a branch (the branch cost eleven register presets per tile in every block)
            const bool unseen = TAIL && m_new == -INFINITY;
            // (the difference is taken on a sanitised operand: -inf - -inf of an unseen row would be a NaN on the unselected arm, which this file's
            // -fno-honor-nans lets the compiler treat as poison - ADVICE r5)
            const float alpha = unseen ? 1.f : __builtin_amdgcn_exp2f((unseen ? 0.f : m_run[tt]) - m_new);   // m_run = -inf, m_new finite -> 0
            const float nm = unseen ? 0.f : -m_new;
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][e], c2, nm));   // masked: 2^-inf = 0
                    psum += p;
                    pb[tt][kt >> 1][(kt & 1) * 4 + e] = f2bf(p);
                }
            m_run[tt] = m_new;
            l_run[tt] = l_run[tt] * alpha + psum;
#pragma unroll
            for (int i = 0; i < C::DT; ++i) scale_acc(o[tt][i], alpha);
            if (TPW > 1) __builtin_amdgcn_sched_barrier(0);          // keep the tiles sequential (hipcc would re-merge them)
        }
        // ---- O^T += V^T P^T: one transposed V fragment feeds TPW MFMAs.  (r >> 1) & 3 of a V row r = jb + 32 kp (+16) + vrow is that of vrow.
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const int r0 = jb + (2 * kp) * 16 + vrow, r1 = r0 + 16;
                const int vsl = (((2 * dt + (vcol >> 3)) ^ (2 * ((vrow >> 1) & 3))) << 3) + (vcol & 7);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&Vs[r0 * D + vsl]));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&Vs[r1 * D + vsl]));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) o[tt][dt] = mfma16(vf, pb[tt][kp], o[tt][dt]);
            }
        }
    };
    {
        int jb = 0;
        for (; jb + 64 <= j1; jb += 64) block(jb, std::false_type{});
        if (jb < j1) block(jb, std::true_type{});
    }
    if (!wave_on) return;
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        float l = l_run[tt];
        l = xor16_sum(l);
        l = xor32_sum(l);
        if (!row_ok[tt]) continue;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        bf16* op = a.out + b * a.o_bs + (long)trow[tt] * a.ldo + thead[tt] * D + 4 * q4;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            bf16x4 ov = {f2bf(o[tt][dt][0] * inv), f2bf(o[tt][dt][1] * inv), f2bf(o[tt][dt][2] * inv), f2bf(o[tt][dt][3] * inv)};
            *reinterpret_cast<bf16x4*>(op + dt * 16) = ov;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LM attention for frame-sized steps (more than 64 query rows per KV head: T = 36 tokens x 7 query heads = 252 rows).
// attn_fwd_kernel gives every 64-row group its own workgroup, so each (stream, KV head, key split) is staged from HBM/L2 into
// LDS four times, through registers, with one block of keys in flight per workgroup: at L_kv = 2,048 it measured 11.5 us
// (1 stream) / 24.7 us (8 streams) per layer for 4.2 / 33.5 MB of K/V - latency- and re-staging-bound (round 2 trace).  Here one
// 8-wave workgroup owns ALL row tiles of a (stream, KV head, key split) - two 16-row tiles per wave, up to 256 rows - so
// K/V are staged once, by LDS-DMA (global_load_lds_dwordx4 straight from the ring slots: the per-lane SOURCE address carries
// the ring mapping and the swizzles), three 64-key stages deep behind a counted vmcnt and one raw s_barrier per block, and
// every K fragment / transposed V fragment read from LDS feeds two MFMAs.  LDS images (an LDS-DMA writes lane-linear, 4 key
// rows of 256 B per wave-instruction):
//   K: chunk c of key row r at slot  r*16 + (c ^ (r & 15))         -> conflict-free ds_read_b128 A fragments (as attn_fwd_kernel)
//   V: chunk c of key row r at slot  r*16 + (c ^ (2 * (r & 7)))    -> conflict-free ds_read_b64_tr_b16: a 32-lane half reads
//      8 rows x 4 eight-byte pieces of two adjacent chunks; rows are 64 banks apart, so the XOR of the chunk PAIR index with
//      (r & 7) puts the 8 rows on 8 different bank octets.
// Same visibility rule, online softmax and partial format as attn_fwd_kernel (attn_combine merges key splits).
// ---------------------------------------------------------------------------------------------
#ifndef AHA_ATTN_ABLATE
#define AHA_ATTN_ABLATE 0      // diagnostic builds only (tools/diag/attn_lm_ablate.sh): 1 no compute, 2 no softmax, 4 no result store, 8 no DMA
#endif
template <int D, int NW>
__global__ __launch_bounds__(64 * NW) void attn_lm_kernel(AttnArgs a, const StepDesc* __restrict__ sdp) {
    constexpr int ABL = AHA_ATTN_ABLATE;
    static_assert(D == 128, "built for head_dim 128 (16 chunks per key row)");
    static_assert(NW == 8 || NW == 4 || NW == 2, "waves per workgroup (8 is the shipped one; 4 and 2 were measured slower than attn_fwd_kernel)");
    constexpr int STG = 64 * D;                  // elements of one operand of one stage (64 keys)
    constexpr int PPW = 16 / NW;                 // K (and V) pieces per wave and stage (1 KiB = 4 key rows each; 16 per operand)
    constexpr int STAGES = 3, P = 2 * PPW;
    constexpr int DT = D / 16, KSQ = D / 32;
    extern __shared__ __attribute__((aligned(16))) char alm_smem[];
    bf16* lds = reinterpret_cast<bf16*>(alm_smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, r16 = lane & 15;
    const int b = blockIdx.z;
    const int R = a.G * a.T, RT = ceil_div(R, 16), RGW = ceil_div(RT, 2 * NW);      // a workgroup covers 2*NW row tiles
    const int hk = blockIdx.y / RGW, rg = blockIdx.y % RGW;
    const int split = blockIdx.x;
    const StreamStep ss = sdp->s[b];
    const int Lk = ss.len_after, off = ss.causal_off;
    const long lo = ((long)a.layer * a.Hkv + hk) * ss.cap * D;
    const bf16* kb = ss.k_base + lo;
    const bf16* vb = ss.v_base + lo;
    const int j0 = split * a.split_len;
    const int j1 = min(Lk, j0 + a.split_len);
    if (j0 >= j1) return;                                   // uniform per block
    const int nblk = ceil_div(j1 - j0, 64);

    // ---- this wave's two query tiles
    int trow[2], thead[2];
    bool row_ok[2];
    bf16x8 qf[2][KSQ];
    const int rt0 = (rg * NW + wave) * 2;
    const bool wave_on = rt0 < RT;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        int r = (rt0 + tt) * 16 + r16;
        row_ok[tt] = (rt0 + tt) < RT && r < R;
        if (r > R - 1) r = R - 1;
        const int g = r / a.T;
        trow[tt] = r % a.T;
        thead[tt] = hk * a.G + g;
        const bf16* qp = a.q + b * a.q_bs + (long)trow[tt] * a.ldq + thead[tt] * D;
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) qf[tt][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32 + q4 * 8);
    }

    // ---- LDS-DMA of one 64-key block: piece (wave*2 + pk) = key rows 4*(wave*2+pk) .. +3 of the block, K and V
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int prow = lane >> 4, pch = lane & 15;
    auto dma = [&](int blk, int stage) {
        if constexpr (ABL & 8) return;
        const int jb = j0 + min(blk, nblk - 1) * 64;        // past the end: refill a dead stage (keeps the vmcnt counts fixed)
        bf16* sk = lds + stage * (2 * STG);
#pragma unroll
        for (int pk = 0; pk < PPW; ++pk) {
            const int row = (wave * PPW + pk) * 4 + prow;
            const int j = min(jb + row, j1 - 1);            // keys past the end re-load the last key: finite data, masked below
            const long so = (long)phys_slot(ss, j) * D;
            __builtin_amdgcn_global_load_lds((gptr_t)(kb + so + ((pch ^ (row & 15)) << 3)), (lptr_t)(sk + (wave * PPW + pk) * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(vb + so + ((pch ^ (2 * (row & 7))) << 3)), (lptr_t)(sk + STG + (wave * PPW + pk) * 512), 16, 0, 0);
        }
    };

    f32x4 o[2][DT];
    float m_run[2], l_run[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        m_run[tt] = -INFINITY;
        l_run[tt] = 0.f;
#pragma unroll
        for (int i = 0; i < DT; ++i) o[tt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    const int vrow = 4 * q4 + (r16 >> 2), vcol = 4 * (r16 & 3);
    const float c2 = a.scale * 1.4426950408889634f;              // scale * log2(e)

    auto compute = [&](int stage, int jb) {
        const bf16* Ks = lds + stage * (2 * STG);
        const bf16* Vs = Ks + STG;
        // ---- S^T tiles of both query tiles: every K fragment feeds two MFMAs
        f32x4 s[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int row = kt * 16 + r16;
            bf16x8 kf[KSQ];
#pragma unroll
            for (int ks = 0; ks < KSQ; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(&Ks[row * D + (((ks * 4 + q4) ^ (row & 15)) << 3)]);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                s[tt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KSQ; ++ks) s[tt][kt] = mfma16(kf[ks], qf[tt][ks], s[tt][kt]);
            }
        }
        // ---- online softmax in the base-2 domain (softmax(scale*s) = 2^(c2*s - c2*max), c2 = scale*log2(e) > 0): one FMA and one
        // v_exp_f32 per probability.  The visibility mask costs ~5 VALU per score, so it is applied only in blocks that touch
        // the causal edge or the end of the split (wave-uniform test); the accumulator rescale is skipped while no row's
        // running maximum moved.  This kernel was VALU-bound on exactly these three (round-2 trace: 6.4k cycles per block).
        bf16x8 pb[2][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const bool all_vis = __all((jb + 63 <= off + trow[tt]) && (jb + 64 <= j1));
            if (!all_vis) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = jb + kt * 16 + 4 * q4 + e;
                        if (!((j < j1) && (j <= off + trow[tt]))) s[tt][kt][e] = -INFINITY;
                    }
            }
            float bmax = fmaxf(fmaxf(s[tt][0][0], s[tt][0][1]), fmaxf(s[tt][0][2], s[tt][0][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt) bmax = fmaxf(bmax, fmaxf(fmaxf(s[tt][kt][0], s[tt][kt][1]), fmaxf(s[tt][kt][2], s[tt][kt][3])));
            bmax = fmaxf(bmax, __shfl_xor(bmax, 16, 64));
            bmax = fmaxf(bmax, __shfl_xor(bmax, 32, 64));
            const float m_new = fmaxf(m_run[tt], bmax * c2);        // -inf * c2 = -inf
            float alpha = 1.f, psum = 0.f;
            if constexpr (ABL & 2) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { psum += s[tt][kt][e]; pb[tt][kt >> 1][(kt & 1) * 4 + e] = f2bf(s[tt][kt][e]); }
                l_run[tt] += psum;
                continue;
            }
            if (m_new == -INFINITY) {                        // nothing visible yet for this row
                pb[tt][0] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                pb[tt][1] = pb[tt][0];
            } else {
                alpha = __builtin_amdgcn_exp2f(m_run[tt] - m_new);   // m_run = -inf -> 0
                const float nm = -m_new;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[tt][kt][e], c2, nm));   // masked: 2^-inf = 0
                        psum += p;
                        pb[tt][kt >> 1][(kt & 1) * 4 + e] = f2bf(p);
                    }
            }
            m_run[tt] = m_new;
            l_run[tt] = l_run[tt] * alpha + psum;
            if (__any(alpha != 1.f)) {
#pragma unroll
                for (int i = 0; i < DT; ++i) o[tt][i] *= alpha;
            }
        }
        // ---- O^T += V^T P^T: one transposed V fragment feeds both tiles (k-slot permutation as in attn_fwd_kernel)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const int rlo = (2 * kp) * 16 + vrow, rhi = rlo + 16, col = dt * 16 + vcol;
                const s16x4 vlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4)(&Vs[rlo * D + ((((col >> 3) ^ (2 * (rlo & 7)))) << 3) + (col & 7)]));
                const s16x4 vhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4)(&Vs[rhi * D + ((((col >> 3) ^ (2 * (rhi & 7)))) << 3) + (col & 7)]));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vlo, vhi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) o[tt][dt] = mfma16(vf, pb[tt][kp], o[tt][dt]);
            }
        }
    };

#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st) dma(st, st);
    int st_cur = 0, st_new = STAGES - 1;
    for (int blk = 0; blk < nblk; ++blk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * P) : "memory");       // this wave's pieces of block blk have landed
        __builtin_amdgcn_s_barrier();                        // everyone's have; everyone is done reading the stage refilled next
        dma(blk + STAGES - 1, st_new);
        if (wave_on && !(ABL & 1)) compute(st_cur, j0 + blk * 64);
        st_cur = st_cur == STAGES - 1 ? 0 : st_cur + 1;
        st_new = st_new == STAGES - 1 ? 0 : st_new + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // nothing may still target LDS when the workgroup retires
    if (!wave_on) return;
    if constexpr (ABL & 4) { if (l_run[0] + l_run[1] != 12345.678f) return; }   // keeps the arithmetic alive, never stores

#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        float l = l_run[tt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (!row_ok[tt]) continue;
        if (a.n_splits == 1) {
            const float inv = l > 0.f ? 1.0f / l : 0.f;               // a row that sees no key gives 0
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 ov = {f2bf(o[tt][dt][0] * inv), f2bf(o[tt][dt][1] * inv), f2bf(o[tt][dt][2] * inv), f2bf(o[tt][dt][3] * inv)};
                *reinterpret_cast<bf16x4*>(lm_out_ptr(a, b, trow[tt], thead[tt] * D + dt * 16 + 4 * q4)) = ov;
            }
        } else {
            const int Rpad = RT * 16;
            const long prw = (((long)b * a.Hkv + hk) * a.n_splits + split) * Rpad + ((rt0 + tt) * 16 + r16);
            float* po = a.part_o + prw * D + 4 * q4;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(po + dt * 16) = o[tt][dt];
            if (q4 == 0) {
                a.part_ml[prw * 2] = m_run[tt];
                a.part_ml[prw * 2 + 1] = l;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Merging the key splits.  One arithmetic for all three kernels below (a row gets the same bits from any of them):
//   per CHUNK of 16 splits (in split order):  Mc = max m_s;  w_s = 2^(m_s - Mc);  Lc = fma(w_s, l_s, Lc);  accc = fma(w_s, p_s, accc)
//   chunks merged in order into (M, L, acc):  Mn = max(M, Mc);  a = 2^(M - Mn), b = 2^(Mc - Mn) (0 for -inf);  L = fma(b, Lc, a L);  acc = fma(b, accc, a acc)
// Up to 16 splits (every cache below 4,096 keys) that is one chunk and exactly the arithmetic of rounds 1-3 (a = 0, b = 1).  Longer caches
// (round 4: up to 64 splits so that a 21.6k-key cache spreads over the chip) have 2-4 chunks; the chunk sums are independent of each
// other, so attn_combine16c_kernel gives each chunk its own threads: one memory round trip for 64 splits instead of four dependent ones
// (the sequential form took 18.7 us per layer on a 600-frame growing stream, as long as the attention kernel itself).
// ---------------------------------------------------------------------------------------------
struct CombChunk8 { float M, L, acc[8]; };
struct CombChunk1 { float M, L, acc; };

template <int D>
static __device__ __forceinline__ CombChunk8 comb_chunk8(const AttnArgs& a, long base, int Rpad, int r, int d0, int s0, int ns) {
    float m_[16], l_[16];
    f32x4 p0[16], p1[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
        if (s0 + s < ns) {                                         // every load of the chunk is issued before the first is used
            const long prow = (base + s0 + s) * Rpad + r;
            m_[s] = a.part_ml[prow * 2];
            l_[s] = a.part_ml[prow * 2 + 1];
            p0[s] = *reinterpret_cast<const f32x4*>(a.part_o + prow * D + d0);
            p1[s] = *reinterpret_cast<const f32x4*>(a.part_o + prow * D + d0 + 4);
        }
    CombChunk8 c;
    c.M = -INFINITY; c.L = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) c.acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s)
        if (s0 + s < ns) c.M = fmaxf(c.M, m_[s]);
#pragma unroll
    for (int s = 0; s < 16; ++s)
        if (s0 + s < ns) {
            const float w = c.M == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_[s] - c.M);    // running maxima live in the base-2 domain
            c.L = __builtin_fmaf(w, l_[s], c.L);
#pragma unroll
            for (int e = 0; e < 4; ++e) { c.acc[e] = __builtin_fmaf(w, p0[s][e], c.acc[e]); c.acc[4 + e] = __builtin_fmaf(w, p1[s][e], c.acc[4 + e]); }
        }
    return c;
}
static __device__ __forceinline__ void comb_merge(float& M, float& L, float* acc, int n, float Mc, float Lc, const float* accc) {
    const float Mn = fmaxf(M, Mc);
    const float fa = M == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(M - Mn), fb = Mc == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(Mc - Mn);
    L = __builtin_fmaf(fb, Lc, fa * L);
    for (int e = 0; e < n; ++e) acc[e] = __builtin_fmaf(fb, accc[e], fa * acc[e]);
    M = Mn;
}

// 16 query rows per 256-thread block: a thread owns 8 output channels of one row (two 16-B partial loads per split, one 16-B store), so
// the 8,064 tiny blocks of attn_combine_kernel at 8 streams become 512.  Chunks one after the other: the form for up to 16 splits.
template <int D>
__global__ __launch_bounds__(256) void attn_combine16_kernel(AttnArgs a, const StepDesc* __restrict__ sdp) {
    static_assert(D == 128, "16 threads x 8 channels per row");
    const int R = a.G * a.T, RT = ceil_div(R, 16), Rpad = RT * 16;
    const int r = blockIdx.x * 16 + (threadIdx.x >> 4), d0 = (threadIdx.x & 15) * 8;
    const int hk = blockIdx.y, b = blockIdx.z;
    if (r >= R) return;
    const int Lk = sdp->s[b].len_after;
    const int ns = min(a.n_splits, ceil_div(Lk, a.split_len));
    const long base = ((long)b * a.Hkv + hk) * a.n_splits;
    float M = -INFINITY, L = 0.f;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < ns; s0 += 16) {
        const CombChunk8 c = comb_chunk8<D>(a, base, Rpad, r, d0, s0, ns);
        comb_merge(M, L, acc, 8, c.M, c.L, c.acc);
    }
    const int g = r / a.T, t = r % a.T;
    bf16x8 ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) ov[e] = f2bf(L > 0.f ? acc[e] / L : 0.f);
    *reinterpret_cast<bf16x8*>(lm_out_ptr(a, b, t, (hk * a.G + g) * D + d0)) = ov;
}

// More than 16 splits: 4 query rows per 256-thread block, the (up to four) chunks of a row on different threads - every partial load of
// the row is in flight at once - then merged in chunk order through LDS.
template <int D>
__global__ __launch_bounds__(256) void attn_combine16c_kernel(AttnArgs a, const StepDesc* __restrict__ sdp) {
    static_assert(D == 128 && AHA_MAX_KEY_SPLITS <= 64, "16 threads x 8 channels per row, four chunks of 16 splits");
    __shared__ float sm[4][4][16][10];                              // [row][chunk][channel group][M, L, acc 0..7]
    const int R = a.G * a.T, RT = ceil_div(R, 16), Rpad = RT * 16;
    const int rl = threadIdx.x >> 6, ch = (threadIdx.x >> 4) & 3, cg = threadIdx.x & 15;
    const int r = blockIdx.x * 4 + rl, d0 = cg * 8;
    const int hk = blockIdx.y, b = blockIdx.z;
    const int Lk = sdp->s[b].len_after;
    const int ns = min(a.n_splits, ceil_div(Lk, a.split_len));
    const long base = ((long)b * a.Hkv + hk) * a.n_splits;
    const int rc = min(r, R - 1);
    if (ch * 16 < ns) {
        const CombChunk8 c = comb_chunk8<D>(a, base, Rpad, rc, d0, ch * 16, ns);
        float* o = sm[rl][ch][cg];
        o[0] = c.M; o[1] = c.L;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[2 + e] = c.acc[e];
    }
    __syncthreads();
    if (ch != 0 || r >= R) return;
    float M = -INFINITY, L = 0.f;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c * 16 < ns; ++c) {
        const float* o = sm[rl][c][cg];
        comb_merge(M, L, acc, 8, o[0], o[1], o + 2);
    }
    const int g = r / a.T, t = r % a.T;
    bf16x8 ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) ov[e] = f2bf(L > 0.f ? acc[e] / L : 0.f);
    *reinterpret_cast<bf16x8*>(lm_out_ptr(a, b, t, (hk * a.G + g) * D + d0)) = ov;
}

// Merge key splits: one block of D threads per (b, kv head, row) - the general form (any head dim, attn_fwd_kernel's partner).
template <int D>
__global__ void attn_combine_kernel(AttnArgs a, const StepDesc* __restrict__ sdp) {
    const int R = a.G * a.T, RT = ceil_div(R, 16), Rpad = RT * 16;
    const int r = blockIdx.x, hk = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
    const int Lk = sdp->s[b].len_after;
    const int ns = min(a.n_splits, ceil_div(Lk, a.split_len));
    const long base = ((long)b * a.Hkv + hk) * a.n_splits;
    float M = -INFINITY, L = 0.f, acc = 0.f;
    for (int s0 = 0; s0 < ns; s0 += 16) {
        float m_[16], l_[16], p_[16];
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s0 + s < ns) {
                const long prow = (base + s0 + s) * Rpad + r;
                m_[s] = a.part_ml[prow * 2];
                l_[s] = a.part_ml[prow * 2 + 1];
                p_[s] = a.part_o[prow * D + d];
            }
        float Mc = -INFINITY, Lc = 0.f, accc = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s0 + s < ns) Mc = fmaxf(Mc, m_[s]);
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s0 + s < ns) {
                const float w = Mc == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_[s] - Mc);
                Lc = __builtin_fmaf(w, l_[s], Lc);
                accc = __builtin_fmaf(w, p_[s], accc);
            }
        comb_merge(M, L, &acc, 1, Mc, Lc, &accc);
    }
    const int g = r / a.T, t = r % a.T;
    if (d < a.hd) *lm_out_ptr(a, b, t, (hk * a.G + g) * a.hd + d) = f2bf(L > 0.f ? acc / L : 0.f);
}

static int g_dense_tpw = 0;      // tuning "attn_tpw": query tiles per wave of the dense kernel (0 = auto, 1..3 forced)
extern "C" void aha_attention_set_dense_tpw(int v) { g_dense_tpw = v; }
static int g_attn_head = 1;      // tuning "attn_head": whole-head-in-LDS dense attention (0 off, 1 auto, 2 always when eligible)
extern "C" void aha_attention_set_head_kernel(int v) { g_attn_head = v; }
static int g_attn_d96 = 1;       // tuning "attn_d96": 96-wide dense template for head dims 65..96 (so400m's 72): 0 = pad to 128 as round 2
extern "C" void aha_attention_set_d96(int v) { g_attn_d96 = v; }
static int g_attn_lm = 1;        // tuning "attn_lm": attn_lm_kernel for frame-sized steps (> 64 rows per KV head, head_dim 128): 0 never, 1 auto, 2 always
extern "C" void aha_attention_set_lm_kernel(int v) { g_attn_lm = v; }

template <int D>
static hipError_t launch_attn(const AttnArgs& a, const StepDesc* sd_dev, int B, hipStream_t st) {
    const int R = a.G * a.T, RT = ceil_div(R, 16), RG = ceil_div(RT, 4);
    dim3 grid(a.n_splits, a.Hkv * RG, B);
    if (sd_dev) {
        if constexpr (D == 128) {
            // attn_lm_kernel (all 16 row tiles of a frame-sized step share one K/V staging) once its grid fills half the chip
            // (4+ streams at W = 2,048: measured 34.7 vs 49.8 us per layer at 8 streams); with fewer streams attn_fwd_kernel's
            // four narrower workgroups per (stream, KV head, split) win (20.6 vs 22.7 us at one stream).  Both kernels give a
            // row the same bits, so the choice - which depends on the batch - cannot change a score.
            // [r6] ... or once a cache is long enough for 23 key splits (~5,700 keys at one stream): measured attention + combine per layer,
            // automatic choice / attn_lm_kernel forced: 5,000 keys 27.8 / 28.6 us, 6,500 keys 30.8 / 29.0 (profiles/r05_ablation_attn_lm_and_split_k.txt
            // section 5; the grid-fill rule alone switched at 7,936 keys).
            const int wgs8 = a.n_splits * a.Hkv * ceil_div(RT, 16) * B;
            if (a.hd == D && RT > 4 && (g_attn_lm == 2 || (g_attn_lm == 1 && (wgs8 >= 128 || a.n_splits >= 23)))) {      // tuning "attn_lm": 0 never, 1 auto, 2 always
                constexpr int LDS = 3 * 2 * 64 * D * 2;
                static bool attr_set = false;
                if (!attr_set) {
                    hipError_t e = hipFuncSetAttribute((const void*)attn_lm_kernel<D, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                    if (e != hipSuccess) return e;
                    attr_set = true;
                }
                hipLaunchKernelGGL((attn_lm_kernel<D, 8>), dim3(a.n_splits, a.Hkv * ceil_div(RT, 16), B), dim3(512), LDS, st, a, sd_dev);
                if (a.n_splits > 16)
                    hipLaunchKernelGGL((attn_combine16c_kernel<D>), dim3(ceil_div(R, 4), a.Hkv, B), dim3(256), 0, st, a, sd_dev);
                else if (a.n_splits > 1)
                    hipLaunchKernelGGL((attn_combine16_kernel<D>), dim3(ceil_div(R, 16), a.Hkv, B), dim3(256), 0, st, a, sd_dev);
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL((attn_fwd_kernel<D, true>), grid, dim3(256), 0, st, a, sd_dev);
        if constexpr (D == 128) {
            // more than one chunk of 16 splits (a one-stream cache of 4k-8k keys): the chunk-parallel merge, one memory round trip instead of
            // two to four dependent ones (same arithmetic, same bits; 9.5 -> 7.9 us at 5,000 keys, 11.1 -> 8.2 at 6,500)
            if (a.n_splits > 16 && a.hd == D) {
                hipLaunchKernelGGL((attn_combine16c_kernel<D>), dim3(ceil_div(R, 4), a.Hkv, B), dim3(256), 0, st, a, sd_dev);
                return hipGetLastError();
            }
        }
        if (a.n_splits > 1)
            hipLaunchKernelGGL((attn_combine_kernel<D>), dim3(R, a.Hkv, B), dim3(D), 0, st, a, sd_dev);
    } else {
        // dense (vision tower): one query tile per wave while the grid is small (single-frame latency), more tiles per
        // wave (less K/V staging per flop) once it fills the chip several times over
        // Measured on 32 frames (4608 x 10 key blocks): 1 / 2 / 3 tiles per wave give the same tower time within noise
        // (20.2 / 20.6 / 20.2 ms), so K/V staging is not what bounds it; halving the softmax's FMA / add instruction count with
        // packed fp32 ops and skipping idle rescales did not move it either (profiles/r02_vit_batch.txt): the per-block chain
        // barrier -> fragment reads -> MFMA -> softmax -> MFMA runs with little overlap at ~108 us.  Default: one tile per wave, the
        // same code path for every batch size; 128-wide heads do not fit more tiles in 256 VGPRs anyway.
        if constexpr (D == 64) {
            // whole head LDS-resident (attn_head64_kernel): 64-wide heads whose K + V fit the CU's LDS.  Throughput path (>= 128 (frame, head)
            // pairs: every CU gets a workgroup): one 12-wave workgroup per pair, three query tiles per wave.  [r5] Latency path (one to
            // seven frames): 4-wave workgroups, each staging the head's K / V once (147 KB by LDS-DMA, ~2.5 us) and running its NW * TPW
            // query tiles over all key blocks with no barrier - TPW the smallest of 1..3 that keeps the launch within one workgroup per CU.
            // The restaging kernel it replaces there walks the 9 key blocks behind two barriers each: 12.2 us per layer at one frame.
            // Same per-row arithmetic in every form: the choice (geometry only) cannot change a bit.  Tuning "attn_head": 0 off, 1 auto, 2
            // the 12-wave form whenever eligible.
            const int lk_pad = round_up(a.Lk, 64), lds = 2 * lk_pad * 64 * 2;
            if (g_attn_head && a.hd == 64 && lds <= 160 * 1024 - 1024 && a.Lk >= 64) {
                static bool attr_set = false;
                if (!attr_set) {
                    hipError_t e = hipFuncSetAttribute((const void*)attn_head64_kernel<12, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_head64_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_head64_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_head64_kernel<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e != hipSuccess) return e;
                    attr_set = true;
                }
                if (RT <= 36 && (g_attn_head == 2 || a.Hkv * B >= 128)) {
                    hipLaunchKernelGGL((attn_head64_kernel<12, 3>), dim3(a.Hkv, B, 1), dim3(768), lds, st, a, lk_pad);
                    return hipGetLastError();
                }
                if (g_attn_head == 1 && a.Hkv * B < 128) {
                    int tpw = 1;
                    while (tpw < 3 && (long)a.Hkv * B * ceil_div(RT, 4 * tpw) > 256) ++tpw;
                    const dim3 grid(a.Hkv, B, ceil_div(RT, 4 * tpw));
                    if (tpw == 1) hipLaunchKernelGGL((attn_head64_kernel<4, 1>), grid, dim3(256), lds, st, a, lk_pad);
                    else if (tpw == 2) hipLaunchKernelGGL((attn_head64_kernel<4, 2>), grid, dim3(256), lds, st, a, lk_pad);
                    else hipLaunchKernelGGL((attn_head64_kernel<4, 3>), grid, dim3(256), lds, st, a, lk_pad);
                    return hipGetLastError();
                }
            }
        }
        int tpw = g_dense_tpw == 0 ? 1 : g_dense_tpw;
        if (D > 64) tpw = 1;
        if (tpw == 3) {
            dim3 g(1, a.Hkv * ceil_div(RT, 12), B);
            hipLaunchKernelGGL((attn_dense_kernel<D, D <= 64 ? 3 : 1>), g, dim3(256), 0, st, a);
        } else if (tpw == 2) {
            dim3 g(1, a.Hkv * ceil_div(RT, 8), B);
            hipLaunchKernelGGL((attn_dense_kernel<D, D <= 64 ? 2 : 1>), g, dim3(256), 0, st, a);
        } else {
            hipLaunchKernelGGL((attn_dense_kernel<D, 1>), grid, dim3(256), 0, st, a);
        }
    }
    return hipGetLastError();
}

// sd_dev: DEVICE pointer to the step descriptor (LM mode), or nullptr: dense (ViT) mode, requires n_splits == 1.
// head_dim: any multiple of 8 up to 128; it runs the 64-, 96- (dense only) or 128-wide template with the surplus
// channels zero-padded on chip (so400m: 72 -> 96).
extern "C" hipError_t aha_attention(const AttnArgs* a_, const StepDesc* sd, int B, int head_dim, hipStream_t st) {
    if (!sd && a_->n_splits != 1) return hipErrorInvalidValue;
    if (a_->n_splits < 1 || a_->n_splits > AHA_MAX_KEY_SPLITS) return hipErrorInvalidValue;   // partial buffers are sized for this many
    if (head_dim < 8 || head_dim > 128 || (head_dim & 7)) return hipErrorInvalidValue;
    AttnArgs a = *a_;
    a.hd = head_dim;
    if (head_dim <= 64) return launch_attn<64>(a, sd, B, st);
    if (!sd && head_dim <= 96 && g_attn_d96) {
        // dense heads of 65..96 channels (so400m: 72): 3 QK^T k-steps and 6 output tiles instead of the 128-wide template's 4 and 8.
        // The padded channels are exact zeros in both templates, so a row's bits do not depend on which one ran.
        // Measured at 32 frames x 729 keys x 16 heads of 72: 279 -> 243 us per layer; two query tiles per wave: 306 us (worse).
        const int RT = ceil_div(a.G * a.T, 16);
        hipLaunchKernelGGL((attn_dense_kernel<96, 1>), dim3(1, a.Hkv * ceil_div(RT, 4), B), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    return launch_attn<128>(a, sd, B, st);
}
