// HBM/L2-bound glue kernels of the streaming path.  All bf16 traffic is 16-B (bf16x8) or 8-B
// vectors; reductions are wave shuffles + one LDS hop.  Rounding points mirror what the torch
// ops of the reference produce in bf16 (see oracle/qwen2_live.py, oracle/vision_tower.py).
#include "aha_kernels.h"

static __device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------
// RMSNorm (modeling_qwen2.py:247-251): fp32 mean of squares, x*rsqrt -> bf16, times bf16 weight.
// One 256-thread block per row; H % 8 == 0, H <= 8192.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ w,
                                                      bf16* __restrict__ out, int ldo, int H, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x, nch = H >> 3;
    bf16x8 v[4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * 256;
        if (c < nch) {
            v[i] = *reinterpret_cast<const bf16x8*>(x + (long)row * ldx + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = bf2f(v[i][e]); ss += f * f; }
        }
    }
    ss = block_sum_256(ss, red);
    const float rstd = rsqrtf(ss / (float)H + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * 256;
        if (c < nch) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + c * 8);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(wv[e]) * rbf(bf2f(v[i][e]) * rstd));
            *reinterpret_cast<bf16x8*>(out + (long)row * ldo + c * 8) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Split-K reduce + residual add + RMSNorm, fused (prologue of the next GEMM):
//   lin = bf16(sum_s partial[s][row][:])          (the Linear's bf16 output)
//   h   = bf16(h + lin)                            (residual stream, updated in place)
//   xn  = w * bf16(h * rsqrt(mean(h^2) + eps))     (next RMSNorm)
// lin_bf16 != null replaces the slab sum (tiled-GEMM fallback path).
// ---------------------------------------------------------------------------------------------

#include "resid_norm_body.h"

__global__ __launch_bounds__(512) void resid_norm_kernel(ResidNormArgs a) {
    __shared__ float red[16];
    resid_norm_row<false>(a, blockIdx.x, red);
}

// ---------------------------------------------------------------------------------------------
// QKV finish: split-K reduce + bias -> bf16 q,k,v ; RoPE on q,k (modeling_qwen2.py:107-131, bf16
// cos/sin table) ; q -> q_rot[M][Hq*D] ; k,v -> the stream's cache slots.  One block per row.
// Columns of the fused QKV GEMM: [ q: Hq*D | k: Hkv*D | v: Hkv*D ].
// ---------------------------------------------------------------------------------------------

template <int D>
__global__ __launch_bounds__(1024) void qkv_finish_kernel(QkvFinishArgs a, const StepDesc* __restrict__ sdp) {
    constexpr int HALF = D / 2, IPH = HALF / 4;           // items (4 rotation pairs) per head
    const int row = blockIdx.x, tid = threadIdx.x;
    const int T = sdp->T, b = row / T, t = row % T;
    const StreamStep ss = sdp->s[b];
    int pos = ss.pos_base + t; if (pos > a.n_pos - 1) pos = a.n_pos - 1;
    const bool store_kv = ss.write_base >= 0 && t < ss.write_count;
    int slot = 0;
    if (store_kv) slot = phys_slot(ss, ss.write_base + t);
    const int qk_items = (a.Hq + a.Hkv) * IPH;
    const int v_items = a.Hkv * (D / 4);

    auto fetch4 = [&](int col, float (&o)[4]) {
        if (a.partial) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* p = a.partial + (long)row * a.ldp + col;
            for (int s0 = 0; s0 < a.S; s0 += 8) {
                f32x4 t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (s0 + j < a.S) t[j] = *reinterpret_cast<const f32x4*>(p + (s0 + j) * a.slab_stride);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (s0 + j < a.S) acc += t[j];
            }
            const bf16x4 bv = *reinterpret_cast<const bf16x4*>(a.bias + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rbf(acc[e] + bf2f(bv[e]));
        } else {
            const bf16x4 xv = *reinterpret_cast<const bf16x4*>(a.qkv_bf16 + (long)row * a.ldq_in + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = bf2f(xv[e]);
        }
    };

    // items are independent (no row-wide reduction), so a row is spread over gridDim.y workgroups: a CU ingests only
    // ~43 GB/s and one workgroup per row pulled 129 KB of slabs (3 us of a 5.6 us kernel)
    for (int it = blockIdx.y * blockDim.x + tid; it < qk_items + v_items; it += gridDim.y * blockDim.x) {
        if (it < qk_items) {
            const int head = it / IPH, d = (it % IPH) * 4;     // head < Hq: query head, else key head
            if (head >= a.Hq && !store_kv) continue;            // frozen static cache: K is neither stored nor (then) projected
            const int col = head * D + d;
            float x1[4], x2[4];
            fetch4(col, x1);
            fetch4(col + HALF, x2);
            const bf16x4 c1 = *reinterpret_cast<const bf16x4*>(a.rope_cos + (long)pos * D + d);
            const bf16x4 s1 = *reinterpret_cast<const bf16x4*>(a.rope_sin + (long)pos * D + d);
            const bf16x4 c2 = *reinterpret_cast<const bf16x4*>(a.rope_cos + (long)pos * D + d + HALF);
            const bf16x4 s2 = *reinterpret_cast<const bf16x4*>(a.rope_sin + (long)pos * D + d + HALF);
            bf16x4 o1, o2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // q*cos + rotate_half(q)*sin, each product and the sum rounded to bf16
                o1[e] = f2bf(rbf(x1[e] * bf2f(c1[e])) + rbf(-x2[e] * bf2f(s1[e])));
                o2[e] = f2bf(rbf(x2[e] * bf2f(c2[e])) + rbf(x1[e] * bf2f(s2[e])));
            }
            if (head < a.Hq) {
                bf16* qp = a.q_rot + (long)row * a.ldq + head * D + d;
                *reinterpret_cast<bf16x4*>(qp) = o1;
                *reinterpret_cast<bf16x4*>(qp + HALF) = o2;
            } else if (store_kv) {
                const int hk = head - a.Hq;
                bf16* kp = ss.k_base + (((long)a.layer * a.Hkv + hk) * ss.cap + slot) * D + d;
                *reinterpret_cast<bf16x4*>(kp) = o1;
                *reinterpret_cast<bf16x4*>(kp + HALF) = o2;
            }
        } else if (store_kv) {
            const int iv = it - qk_items;
            const int hk = iv / (D / 4), d = (iv % (D / 4)) * 4;
            float x[4];
            fetch4((a.Hq + a.Hkv) * D + hk * D + d, x);
            bf16x4 o = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3])};
            *reinterpret_cast<bf16x4*>(ss.v_base + (((long)a.layer * a.Hkv + hk) * ss.cap + slot) * D + d) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Frozen TrulyStaticCache steps with a short prefix (BASELINE configs[1]: the 20-token query turn): qkv_finish AND the
// attention in one launch.  After its first call the static cache returns only the frozen prefix (test/static_cache.py:33-36),
// so a new token's attention needs its own rotated query rows and the <= 64 cached keys - nothing another workgroup
// produces.  One workgroup per (token row, KV head): split-K reduce + bias + RoPE of the G query heads that share the KV head
// (same rounding points as qkv_finish_kernel), then softmax(q K^T) V on the vector ALUs (G x L_kv x D = 18k MACs per
// workgroup: far too small for a tile kernel, whose launch + LDS staging + MFMA drain cost 8.9 us here against 1-2 us).
// The step's K/V projections are not consumed (nothing is stored for a frozen cache); every cached key is visible.
// Arithmetic (this kernel's own; rows independent, so batched, solo and last-token-only steps agree bit for bit):
//   s_j = sum_d q_d k_jd in d order (fp32);  p_j = 2^(c2 s_j - c2 max_j s_j), c2 = scale log2 e;  l = sum_j p_j (fp32);
//   out_d = bf16( (sum_j bf16(p_j) v_jd in j order) / l )  - P rounded to bf16 before PV like the tile kernels.
// ---------------------------------------------------------------------------------------------
struct StaticAttnArgs {
    QkvFinishArgs qa;
    bf16* out; int ldo;              // attention output rows [M][Hq*D]
    float scale; int G;
    int T;                           // tokens per stream of the step (= the descriptor's T)
};

template <int D>
__global__ __launch_bounds__(256) void qkv_finish_attn_static_kernel(StaticAttnArgs p, const StepDesc* __restrict__ sdp) {
    constexpr int HALF = D / 2, IPH = HALF / 4, KST = D + 8, GMAX = 8, LMAX = 64;
    __shared__ __attribute__((aligned(16))) bf16 Ksh[LMAX * KST];
    __shared__ __attribute__((aligned(16))) bf16 Vsh[LMAX * KST];
    __shared__ float qs[GMAX * D];
    __shared__ float ps[GMAX * LMAX];
    const QkvFinishArgs& a = p.qa;
    const int row = blockIdx.x, hk = blockIdx.y, tid = threadIdx.x, G = p.G;
    // Memory round trips are this kernel's time, so they are issued in dependency order, not reading order: (1) this thread's
    // slab loads need only the row; (2) the step descriptor (T comes as an argument: one dependent load less); (3) the prefix
    // K/V, whose addresses come from the descriptor; RoPE (needs the position) runs last on the values already in registers.
    auto fetch4 = [&](int col, float (&o)[4]) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* pp = a.partial + (long)row * a.ldp + col;
        for (int s0 = 0; s0 < a.S; s0 += 8) {
            f32x4 tt[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < a.S) tt[j] = *reinterpret_cast<const f32x4*>(pp + (s0 + j) * a.slab_stride);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < a.S) acc += tt[j];
        }
        const bf16x4 bv = *reinterpret_cast<const bf16x4*>(a.bias + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rbf(acc[e] + bf2f(bv[e]));
    };
    // first (for head_dim 128 and G <= 8: only) item of this thread: 4 rotation pairs of one query head
    const bool first = tid < G * IPH;
    float f1[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
    if (first) {
        const int col = (hk * G + tid / IPH) * D + (tid % IPH) * 4;
        fetch4(col, f1);
        fetch4(col + HALF, f2);
    }
    const int T = p.T, b = row / T, t = row % T;
    const StreamStep ss = sdp->s[b];
    const int Lk = min(ss.len_after, LMAX);
    int pos = ss.pos_base + t; if (pos > a.n_pos - 1) pos = a.n_pos - 1;
    // keys 0 .. nvis-1 are visible to this row: all of them under the sdpa-style rule; j <= causal_off + t under flash-attn-2's
    // bottom-right alignment (causal_off = L - T: the first T - L rows of a frame then see nothing and give 0)
    const int nvis = max(0, min(Lk, ss.causal_off > (1 << 28) ? Lk : ss.causal_off + t + 1));

    // ---- the frozen prefix of this KV head -> LDS
    {
        const long lo = ((long)a.layer * a.Hkv + hk) * ss.cap * D;
        for (int c = tid; c < Lk * (D / 8); c += 256) {
            const int j = c / (D / 8), ch = c % (D / 8);
            const long so = lo + (long)phys_slot(ss, j) * D + ch * 8;
            *reinterpret_cast<bf16x8*>(&Ksh[j * KST + ch * 8]) = *reinterpret_cast<const bf16x8*>(ss.k_base + so);
            *reinterpret_cast<bf16x8*>(&Vsh[j * KST + ch * 8]) = *reinterpret_cast<const bf16x8*>(ss.v_base + so);
        }
    }
    // ---- rotated queries of the G heads (qkv_finish_kernel's arithmetic)
    for (int it = tid; it < G * IPH; it += 256) {
        const int g = it / IPH, dd = (it % IPH) * 4, col = (hk * G + g) * D + dd;
        float x1[4], x2[4];
        if (it == tid && first) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { x1[e] = f1[e]; x2[e] = f2[e]; }
        } else {
            fetch4(col, x1);
            fetch4(col + HALF, x2);
        }
        const bf16x4 c1 = *reinterpret_cast<const bf16x4*>(a.rope_cos + (long)pos * D + dd), s1 = *reinterpret_cast<const bf16x4*>(a.rope_sin + (long)pos * D + dd);
        const bf16x4 c2 = *reinterpret_cast<const bf16x4*>(a.rope_cos + (long)pos * D + dd + HALF), s2 = *reinterpret_cast<const bf16x4*>(a.rope_sin + (long)pos * D + dd + HALF);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qs[g * D + dd + e] = rbf(rbf(x1[e] * bf2f(c1[e])) + rbf(-x2[e] * bf2f(s1[e])));
            qs[g * D + dd + HALF + e] = rbf(rbf(x2[e] * bf2f(c2[e])) + rbf(x1[e] * bf2f(s2[e])));
        }
    }
    __syncthreads();
    // ---- scores
    for (int i = tid; i < G * Lk; i += 256) {
        const int g = i / Lk, j = i % Lk;
        float acc = 0.f;
#pragma unroll 8
        for (int dch = 0; dch < D / 8; ++dch) {
            const bf16x8 kv = *reinterpret_cast<const bf16x8*>(&Ksh[j * KST + dch * 8]);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(qs[g * D + dch * 8 + e], bf2f(kv[e]), acc);
        }
        ps[g * LMAX + j] = acc;
    }
    __syncthreads();
    // ---- softmax numerators (every thread of a head recomputes the head's max: <= 64 LDS reads)
    const float c2 = p.scale * 1.4426950408889634f;
    float pj[2] = {0.f, 0.f};                                   // G * Lk <= 8 * 64 = 512 = two per thread
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 256;
        if (i < G * Lk) {
            const int gi = i / Lk;
            float m = -INFINITY;
            for (int j = 0; j < nvis; j += 8) {              // eight independent LDS reads in flight (the serial loop cost ~1 us of latency)
                float v8[8];
#pragma unroll
                for (int u8 = 0; u8 < 8; ++u8) v8[u8] = j + u8 < nvis ? ps[gi * LMAX + j + u8] : -INFINITY;
#pragma unroll
                for (int u8 = 0; u8 < 8; ++u8) m = fmaxf(m, v8[u8]);
            }
            pj[u] = (i % Lk) < nvis ? __builtin_amdgcn_exp2f(__builtin_fmaf(ps[gi * LMAX + i % Lk], c2, -m * c2)) : 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 256;
        if (i < G * Lk) ps[(i / Lk) * LMAX + i % Lk] = pj[u];
    }
    __syncthreads();
    // ---- out[g][d] = sum_j bf16(p_j) v_jd / sum_j p_j
    for (int i = tid; i < G * (D / 4); i += 256) {
        const int g = i / (D / 4), dd = (i % (D / 4)) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, l = 0.f;
#pragma unroll 4
        for (int j = 0; j < Lk; ++j) {                      // loads of four keys in flight; the sums stay in j order
            const float pv = ps[g * LMAX + j];
            l += pv;
            const float pb = rbf(pv);
            const bf16x4 vv = *reinterpret_cast<const bf16x4*>(&Vsh[j * KST + dd]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(pb, bf2f(vv[e]), acc[e]);
        }
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        bf16x4 o = {f2bf(acc[0] * inv), f2bf(acc[1] * inv), f2bf(acc[2] * inv), f2bf(acc[3] * inv)};
        *reinterpret_cast<bf16x4*>(p.out + (long)row * p.ldo + (hk * G + g) * D + dd) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// SinkCache re-rotation of the kept window keys, in place (test/sink_cache.py:27-33,139-150):
//   k' = bf16( bf16(k*cos_r) + bf16(rotate_half(k)*sin_r) ), table row = rerot_row0 + kept index.
// In the ring layout kept keys do not move, so this is the ONLY per-step traffic on old keys.
// grid: (key blocks, layers*Hkv, B); block 256 threads = (256 / (D/8)) keys x D/8 items.
// ---------------------------------------------------------------------------------------------
template <int HALF>
static __device__ __forceinline__ void rerotate_store4(bf16* kp, bf16x4 x1, bf16x4 x2, bf16x4 c1, bf16x4 s1, bf16x4 c2, bf16x4 s2) {
    bf16x4 o1, o2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a1 = bf2f(x1[e]), a2 = bf2f(x2[e]);
        o1[e] = f2bf(rbf(a1 * bf2f(c1[e])) + rbf(-a2 * bf2f(s1[e])));
        o2[e] = f2bf(rbf(a2 * bf2f(c2[e])) + rbf(a1 * bf2f(s2[e])));
    }
    *reinterpret_cast<bf16x4*>(kp) = o1;
    *reinterpret_cast<bf16x4*>(kp + HALF) = o2;
}

// Re-rotation coefficients of kept key `row` (columns d..d+3), computed from the bf16 RoPE table exactly as
// SinkCache._get_rerotation_cos_sin does (test/sink_cache.py:35-55) for T new tokens:
//   original = table[sink+T + row], shifted = table[sink + row], both widened to fp32
//   cos_r = oc*sc + os*ss ;  sin_r = -os*sc + oc*ss   -> bf16
// torch evaluates each product and each sum as its own fp32 op, so contraction into FMAs is switched off (checked in the
// ISA: v_mul_f32 / v_add_f32 only).  Same arithmetic as rerot_table_kernel below: a step that computes the coefficients
// on the fly (no table registered) and one that reads a table give the same bits.  The RoPE table is L2-resident
// (a few hundred KB), so the kernels stay bound by the K traffic.
static __device__ __forceinline__ void rerot_coef4(const bf16* __restrict__ cosb, const bf16* __restrict__ sinb, int D, int sink, int T,
                                                   int row, int d, bf16x4* c_out, bf16x4* s_out) {
#pragma clang fp contract(off)
    const long o0 = (long)(sink + T + row) * D + d, s0 = (long)(sink + row) * D + d;
    const bf16x4 ocv = *reinterpret_cast<const bf16x4*>(cosb + o0), osv = *reinterpret_cast<const bf16x4*>(sinb + o0);
    const bf16x4 scv = *reinterpret_cast<const bf16x4*>(cosb + s0), ssv = *reinterpret_cast<const bf16x4*>(sinb + s0);
    bf16x4 c, s;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float oc = bf2f(ocv[e]), os = bf2f(osv[e]), sc = bf2f(scv[e]), ss = bf2f(ssv[e]);
        const float p0 = oc * sc, p1 = os * ss, p2 = (-os) * sc, p3 = oc * ss;
        const float cc = p0 + p1, sn = p2 + p3;
        c[e] = f2bf(rbf(cc));
        s[e] = f2bf(rbf(sn));
    }
    *c_out = c;
    *s_out = s;
}

// rcos_/rsin_ == nullptr: no (window, sink, T) table is registered - the coefficients come from the RoPE table (cosb/sinb) on
// the fly, so the first evicting step of a stream allocates nothing and launches no table build.
// A thread owns one (kept key, 4-channel group) and walks the (layer, kv head) planes blockIdx.y, blockIdx.y + gridDim.y, ..:
// the coefficients depend on the key only, so they are fetched / computed ONCE per thread and the loop body is the K row's
// 16 bytes in, 16 bytes out (four planes in flight) - HBM-bound on the 113 MB per stream-step it has to move.
// grid: (key blocks, plane groups, B); block 256 threads = (256 / (D/8)) keys x D/8 items.
template <int D>
__global__ __launch_bounds__(256) void sink_rerotate_kernel(const StepDesc* __restrict__ sdp, unsigned stream_mask,
                                                            const void* __restrict__ rcos_, const void* __restrict__ rsin_,
                                                            const void* __restrict__ cosb_, const void* __restrict__ sinb_,
                                                            int planes) {
    const bf16 *rcos = static_cast<const bf16*>(rcos_), *rsin = static_cast<const bf16*>(rsin_);
    constexpr int HALF = D / 2, IPK = HALF / 4, KPB = 256 / IPK;
    const int b = blockIdx.z;
    if (!((stream_mask >> b) & 1u)) return;                 // streams that share this launch's (W, sink) geometry
    const StreamStep ss = sdp->s[b];
    const int key = blockIdx.x * KPB + threadIdx.x / IPK;
    if (key >= ss.n_rerot) return;
    const int d = (threadIdx.x % IPK) * 4;
    const int slot = phys_slot(ss, ss.n_fixed + key);
    bf16x4 c1, s1, c2, s2;
    if (rcos) {
        const long trow = (long)(ss.rerot_row0 + key) * D;
        c1 = *reinterpret_cast<const bf16x4*>(rcos + trow + d); s1 = *reinterpret_cast<const bf16x4*>(rsin + trow + d);
        c2 = *reinterpret_cast<const bf16x4*>(rcos + trow + d + HALF); s2 = *reinterpret_cast<const bf16x4*>(rsin + trow + d + HALF);
    } else {
        const bf16 *cosb = static_cast<const bf16*>(cosb_), *sinb = static_cast<const bf16*>(sinb_);
        rerot_coef4(cosb, sinb, D, ss.n_fixed, sdp->T, ss.rerot_row0 + key, d, &c1, &s1);
        rerot_coef4(cosb, sinb, D, ss.n_fixed, sdp->T, ss.rerot_row0 + key, d + HALF, &c2, &s2);
    }
    const long pstride = (long)ss.cap * D;                   // elements between consecutive (layer, kv head) planes
    bf16* k0 = ss.k_base + (long)slot * D + d;
    const int step = gridDim.y;
    int p = blockIdx.y;
    for (; p + 3 * step < planes; p += 4 * step) {           // four planes in flight per thread
        bf16* kp[4];
        bf16x4 x1[4], x2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            kp[u] = k0 + (long)(p + u * step) * pstride;
            x1[u] = *reinterpret_cast<const bf16x4*>(kp[u]);
            x2[u] = *reinterpret_cast<const bf16x4*>(kp[u] + HALF);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) rerotate_store4<HALF>(kp[u], x1[u], x2[u], c1, s1, c2, s2);
    }
    for (; p < planes; p += step) {
        bf16* kp = k0 + (long)p * pstride;
        const bf16x4 x1 = *reinterpret_cast<const bf16x4*>(kp), x2 = *reinterpret_cast<const bf16x4*>(kp + HALF);
        rerotate_store4<HALF>(kp, x1, x2, c1, s1, c2, s2);
    }
}

// Operator-level Cache.update of ONE layer (aha_cache_update): the same ring addressing and the same re-rotation arithmetic
// as the fused step (qkv_finish_kernel's K/V store + sink_rerotate_kernel), with the step described by value.
// grid: (max(n_rerot, T) key blocks, Hkv); new K/V: bf16 [Hkv][T][D] (already rotated, as the reference's attention hands
// them to Cache.update, test/sink_cache.py:74-80).
template <int D>
__global__ __launch_bounds__(256) void cache_update_layer_kernel(StreamStep ss, int layer, int Hkv, int T, const void* __restrict__ knew_,
                                                                 const void* __restrict__ vnew_, const void* __restrict__ rcos_,
                                                                 const void* __restrict__ rsin_, const void* __restrict__ cosb_,
                                                                 const void* __restrict__ sinb_) {
    const bf16 *knew = static_cast<const bf16*>(knew_), *vnew = static_cast<const bf16*>(vnew_);
    const bf16 *rcos = static_cast<const bf16*>(rcos_), *rsin = static_cast<const bf16*>(rsin_);
    constexpr int HALF = D / 2, IPK = HALF / 4, KPB = 256 / IPK;
    const int hk = blockIdx.y;
    const int key = blockIdx.x * KPB + threadIdx.x / IPK;
    const int d = (threadIdx.x % IPK) * 4;
    const long lo = ((long)layer * Hkv + hk) * ss.cap;
    if (key < ss.n_rerot) {                                  // kept window keys: re-rotate in place
        bf16* kp = ss.k_base + (lo + phys_slot(ss, ss.n_fixed + key)) * D + d;
        const bf16x4 x1 = *reinterpret_cast<const bf16x4*>(kp), x2 = *reinterpret_cast<const bf16x4*>(kp + HALF);
        bf16x4 c1, s1, c2, s2;
        if (rcos) {
            const long trow = (long)(ss.rerot_row0 + key) * D;
            c1 = *reinterpret_cast<const bf16x4*>(rcos + trow + d); s1 = *reinterpret_cast<const bf16x4*>(rsin + trow + d);
            c2 = *reinterpret_cast<const bf16x4*>(rcos + trow + d + HALF); s2 = *reinterpret_cast<const bf16x4*>(rsin + trow + d + HALF);
        } else {
            const bf16 *cosb = static_cast<const bf16*>(cosb_), *sinb = static_cast<const bf16*>(sinb_);
            rerot_coef4(cosb, sinb, D, ss.n_fixed, T, ss.rerot_row0 + key, d, &c1, &s1);
            rerot_coef4(cosb, sinb, D, ss.n_fixed, T, ss.rerot_row0 + key, d + HALF, &c2, &s2);
        }
        rerotate_store4<HALF>(kp, x1, x2, c1, s1, c2, s2);
    }
    if (key < ss.write_count && ss.write_base >= 0) {        // new tokens: append (distinct slots from the kept keys)
        const long slot = lo + phys_slot(ss, ss.write_base + key);
        const bf16* ks = knew + ((long)hk * T + key) * D + d;
        const bf16* vs = vnew + ((long)hk * T + key) * D + d;
        *reinterpret_cast<bf16x4*>(ss.k_base + slot * D + d) = *reinterpret_cast<const bf16x4*>(ks);
        *reinterpret_cast<bf16x4*>(ss.k_base + slot * D + d + HALF) = *reinterpret_cast<const bf16x4*>(ks + HALF);
        *reinterpret_cast<bf16x4*>(ss.v_base + slot * D + d) = *reinterpret_cast<const bf16x4*>(vs);
        *reinterpret_cast<bf16x4*>(ss.v_base + slot * D + d + HALF) = *reinterpret_cast<const bf16x4*>(vs + HALF);
    }
}

// ---------------------------------------------------------------------------------------------
// The three scoring heads on final-normed hidden rows (video_head_live_llava_qwen.py:185-188) and
// the score post-ops of _encode_frame (test/inference.py:222-227).  heads_w = [info0; info1; rel; unc]
// rows of H.  Row i of the launch reads hidden row (row_first + i*row_step).
//   raw[i]    = bf16-rounded logits (info0, info1, rel_logit, log_var)
//   scores[i] = (softmax(info)[1], sigmoid(rel), exp(log_var))
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void heads_kernel(const bf16* __restrict__ xn, int ldx, int row_first, int row_step,
                                                    const bf16* __restrict__ heads_w, int H, float* __restrict__ scores,
                                                    float* __restrict__ raw, const int* __restrict__ poison) {
    __shared__ float red[4][4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16* x = xn + (long)(row_first + i * row_step) * ldx;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = tid; c < (H >> 3); c += 256) {
        const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + c * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(heads_w + (long)k * H + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[k] += bf2f(xv[e]) * bf2f(wv[e]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave][k] = acc[k];
    __syncthreads();
    if (tid == 0) {
        float l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) l[k] = rbf(red[0][k] + red[1][k] + red[2][k] + red[3][k]);
        if (poison && *poison)                       // a fused kernel's grid barrier timed out earlier: never return plausible numbers
#pragma unroll
            for (int k = 0; k < 4; ++k) l[k] = __builtin_nanf("");
        if (raw) { raw[i * 4 + 0] = l[0]; raw[i * 4 + 1] = l[1]; raw[i * 4 + 2] = l[2]; raw[i * 4 + 3] = l[3]; }
        if (scores) {
            const float m = fmaxf(l[0], l[1]);
            const float e0 = expf(l[0] - m), e1 = expf(l[1] - m);
            scores[i * 3 + 0] = e1 / (e0 + e1);
            scores[i * 3 + 1] = 1.0f / (1.0f + expf(-l[2]));
            scores[i * 3 + 2] = expf(l[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Vision glue
// ---------------------------------------------------------------------------------------------
// uint8 frames [N,3,S,S] -> normalised bf16 im2col rows [N*Np][Kp], column = c*P*P + iy*P + ix,
// value = bf16((u8/255 - .5)/.5)  (preprocess fused into the patch gather; vision_live.py:11-13)
struct PixelNorm { float mean[3], std[3]; };
__global__ void im2col_norm_kernel(const uint8_t* __restrict__ frames, int N, int S, int P, int grid, int Kp, PixelNorm pn,
                                   bf16* __restrict__ out) {
    // torch evaluates frames * (1/255), the subtraction and the division as three separately rounded fp32 ops: no FMA contraction
    // (a fused u * c - mean differs in the last fp32 bit and flips a bf16 rounding now and then: tests/test_gpu_vision_kernels.py)
#pragma clang fp contract(off)
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int kch = Kp >> 3;
    const long total = (long)N * grid * grid * kch;
    if (gid >= total) return;
    const int kc = (int)(gid % kch);
    const long rowi = gid / kch;
    const int px = (int)(rowi % grid), py = (int)((rowi / grid) % grid), n = (int)(rowi / ((long)grid * grid));
    const int PP = P * P, K = 3 * PP;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = kc * 8 + e;
        float v = 0.f;
        if (k < K) {
            const int c = k / PP, rem = k % PP, iy = rem / P, ix = rem % P;
            const uint8_t u = frames[(((long)n * 3 + c) * S + (py * P + iy)) * S + (px * P + ix)];
            v = ((float)u * 0.00392156862745098f - pn.mean[c]) / pn.std[c];   // normalize(frames * 1/255, mean, std) in fp32
        }
        o[e] = f2bf(v);
    }
    *reinterpret_cast<bf16x8*>(out + rowi * Kp + kc * 8) = o;
}

// LayerNorm over the last dim, one wave per row (4 rows per block); D % 8 == 0, D <= 4096.  NC = 16-byte chunks per lane the
// instantiation holds (ceil(D / 512) <= NC): the 8-chunk form needs 132 VGPRs (3 waves per SIMD, two loads in flight per wave
// at D = 1024: 3.8 TB/s on the 32-frame tower); the 2- and 3-chunk forms run at full occupancy.  Same per-lane order: same bits.
//
// Latency path (one to four frames): the launch can carry RIDERS - extra workgroups in front of the row workgroups that do nothing
// but read byte ranges (the weights of GEMMs a few launches ahead, WeightPrefetch in aha_kernels.h) so that those bytes sit in the
// Infinity Cache when their GEMM starts.  At one frame the tower's kernels are latency-bound chains with HBM almost idle (25 MB
// of weights per 85 us layer): with the weights cache-resident the encode measures 1.74 ms instead of 2.09 (tuning "vit_alias",
// tools/diag/vit_alias.py).  Riders change no output bit; with n_riders = 0 the launch is the plain LayerNorm.
// KB: `out` is written k-blocked [D/32][M][32] (ldo unused) for the persistent tile GEMM that reads it next (gemm_tile_p.hip, akb).
template <int NC, bool PF = false, bool KB = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16* __restrict__ x, int ldx, const bf16* __restrict__ w,
                                                        const bf16* __restrict__ bias, bf16* __restrict__ out, int ldo,
                                                        int M, int D, float eps, WeightPrefetch pf) {
    int blk = blockIdx.x;
    if constexpr (PF) {
        if (blk < pf.n_riders) { prefetch_rider(pf, blk); return; }
        blk -= pf.n_riders;
    }
    const int row = blk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int nch = D >> 3;
    bf16x8 v[NC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            v[i] = *reinterpret_cast<const bf16x8*>(x + (long)row * ldx + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += bf2f(v[i][e]);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dlt = bf2f(v[i][e]) - mean; vs += dlt * dlt; }
        }
    }
    const float rstd = rsqrtf(wave_sum(vs) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + c * 8);
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bias + c * 8);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf((bf2f(v[i][e]) - mean) * rstd * bf2f(wv[e]) + bf2f(bv[e]));
            if constexpr (KB) *reinterpret_cast<bf16x8*>(out + ((long)(c >> 2) * M + row) * 32 + (c & 3) * 8) = o;
            else *reinterpret_cast<bf16x8*>(out + (long)row * ldo + c * 8) = o;
        }
    }
}

// post_projector_pooling (video_head_live_llava_qwen.py:117-136): [N, g, g, H] -> [N, go, go, H].
// mode 0 = bilinear (align_corners=False), 1 = average (stride x stride), 2 = max.
__global__ void pool_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int N, int g, int go, int H,
                            int stride, int mode, int frame_rows) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int hch = H >> 3;
    const long total = (long)N * go * go * hch;
    if (gid >= total) return;
    const int hc = (int)(gid % hch);
    const long cell = gid / hch;
    const int ox = (int)(cell % go), oy = (int)((cell / go) % go), n = (int)(cell / ((long)go * go));
    const bf16* base = in + (long)n * frame_rows * H + hc * 8;       // frame_rows >= g*g (CLIP: a class-token row follows the patches)
    float acc[8];
    if (mode == 0) {
        const float sc = (float)g / (float)go;
        float sy = sc * ((float)oy + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
        float sx = sc * ((float)ox + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < g - 1 ? 1 : 0), x1 = x0 + (x0 < g - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const bf16x8 v00 = *reinterpret_cast<const bf16x8*>(base + ((long)y0 * g + x0) * H);
        const bf16x8 v01 = *reinterpret_cast<const bf16x8*>(base + ((long)y0 * g + x1) * H);
        const bf16x8 v10 = *reinterpret_cast<const bf16x8*>(base + ((long)y1 * g + x0) * H);
        const bf16x8 v11 = *reinterpret_cast<const bf16x8*>(base + ((long)y1 * g + x1) * H);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            acc[e] = hy * (hx * bf2f(v00[e]) + lx * bf2f(v01[e])) + ly * (hx * bf2f(v10[e]) + lx * bf2f(v11[e]));
    } else if (mode == 3) {
        // adaptive_avg_pool2d g -> go (models/vision_live.py:18-25): window [floor(o*g/go), ceil((o+1)*g/go))
        const int y0 = (oy * g) / go, y1 = ((oy + 1) * g + go - 1) / go;
        const int x0 = (ox * g) / go, x1 = ((ox + 1) * g + go - 1) / go;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const bf16x8 vv = *reinterpret_cast<const bf16x8*>(base + ((long)y * g + x) * H);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += bf2f(vv[e]);
            }
        const float cnt = (float)((y1 - y0) * (x1 - x0));
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = acc[e] / cnt;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = (mode == 1) ? 0.f : -INFINITY;
        for (int dy = 0; dy < stride; ++dy)
            for (int dx = 0; dx < stride; ++dx) {
                const bf16x8 vv = *reinterpret_cast<const bf16x8*>(base + ((long)(oy * stride + dy) * g + (ox * stride + dx)) * H);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = (mode == 1) ? acc[e] + bf2f(vv[e]) : fmaxf(acc[e], bf2f(vv[e]));
            }
        if (mode == 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] /= (float)(stride * stride);
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
    *reinterpret_cast<bf16x8*>(out + cell * H + hc * 8) = o;
}

// Rows of the tower output that bilinear pooling g -> go actually samples, compacted to a (2 go) x (2 go) grid per frame.
// With an even integer stride s = g / go and align_corners = False the source coordinate of output cell o is s*o + s/2 - 1/2:
// rows s*o + s/2 - 1 and s*o + s/2 with weights exactly 1/2 - so the projector (two GEMMs, 19 GFLOP per frame at full size)
// needs only 4 go^2 of the g^2 patch rows (144 of 576 at 24 -> 6), and pooling the compact grid 2 go -> go reproduces the same
// arithmetic on the same values (stride 2: rows 2o, 2o + 1, weights 1/2): bit-identical embeddings.
__global__ void gather_pool_rows_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int N, int g, int go, int s, int Dv,
                                        int frame_rows) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int dch = Dv >> 3, gc = 2 * go;
    if (gid >= (long)N * gc * gc * dch) return;
    const int c = (int)(gid % dch);
    const long cell = gid / dch;
    const int cx = (int)(cell % gc), cy = (int)((cell / gc) % gc), n = (int)(cell / ((long)gc * gc));
    const int y = s * (cy >> 1) + s / 2 - 1 + (cy & 1), x = s * (cx >> 1) + s / 2 - 1 + (cx & 1);
    *reinterpret_cast<bf16x8*>(out + cell * Dv + c * 8) = *reinterpret_cast<const bf16x8*>(in + ((long)n * frame_rows + (long)y * g + x) * Dv + c * 8);
}

// Token embedding gather: out[i][:] = table[ids[i]][:]
__global__ void embed_gather_kernel(const long* __restrict__ ids, int n, const bf16* __restrict__ table, int H, int vocab,
                                    bf16* __restrict__ out, int ldo) {
    const int i = blockIdx.x;
    long id = ids[i]; if (id < 0) id = 0; if (id > vocab - 1) id = vocab - 1;
    for (int c = threadIdx.x; c < (H >> 3); c += blockDim.x)
        *reinterpret_cast<bf16x8*>(out + (long)i * ldo + c * 8) = *reinterpret_cast<const bf16x8*>(table + id * H + c * 8);
}

// argmax over fp32 logits rows (first maximal index, like torch.argmax)
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int ld, int V, long* __restrict__ out) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int c = tid; c < V; c += 256) {
        const float v = logits[(long)row * ld + c];
        if (v > best || (v == best && c < idx)) { best = v; idx = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[row] = idx;
    }
}

// RepetitionPenaltyLogitsProcessor on one logits row (transformers logits_process.py; used by fast_greedy_generate,
// models/modeling_live.py:73-78): score = gather(logits, ids); score = score < 0 ? score * p : score / p; scatter back.
// Two phases so that duplicate ids all read the ORIGINAL value (gather-then-scatter semantics).  One block.
__global__ __launch_bounds__(1024) void repetition_penalty_kernel(float* __restrict__ logits, int V, const long* __restrict__ hist,
                                                                  const int* __restrict__ n_hist, float penalty, float* __restrict__ tmp) {
    const int n = *n_hist;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        long id = hist[i]; if (id < 0) id = 0; if (id > V - 1) id = V - 1;
        const float sc = logits[id];
        tmp[i] = sc < 0.f ? sc * penalty : sc / penalty;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        long id = hist[i]; if (id < 0) id = 0; if (id > V - 1) id = V - 1;
        logits[id] = tmp[i];
    }
}
// generated_token_ids.append(tok) unless tok is EOS (modeling_live.py:81-82); also mirrors the token for the host poll.
__global__ void generation_bookkeep_kernel(const long* __restrict__ tok, long eos, long* __restrict__ hist, int* __restrict__ n_hist, int cap,
                                           int use_hist, long* __restrict__ out_ids, int i) {
    const long t = *tok;
    out_ids[i] = t;
    if (use_hist && t != eos && *n_hist < cap) { hist[*n_hist] = t; *n_hist = *n_hist + 1; }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
extern "C" {
hipError_t aha_rmsnorm(const bf16* x, int ldx, const bf16* w, bf16* out, int ldo, int M, int H, float eps, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    if ((H & 7) || H > 8192) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(M), dim3(256), 0, st, x, ldx, w, out, ldo, H, eps);
    return hipGetLastError();
}
hipError_t aha_resid_norm(const ResidNormArgs* a, int M, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    if ((a->H & 7) || a->H > 8192) return hipErrorInvalidValue;
    int threads = round_up(a->H >> 3, 64);
    if (threads > 512) threads = 512;
    hipLaunchKernelGGL(resid_norm_kernel, dim3(M), dim3(threads), 0, st, *a);
    return hipGetLastError();
}
hipError_t aha_qkv_finish(const QkvFinishArgs* a, const StepDesc* sd_dev, int M, hipStream_t st) {
    const int items = (a->Hq + a->Hkv) * (a->D / 8) + a->Hkv * (a->D / 4);              // one item per thread
    const int threads = 128;
    int groups = ceil_div(items, threads);
    if (groups > 16) groups = 16;
    if (a->D == 64) hipLaunchKernelGGL((qkv_finish_kernel<64>), dim3(M, groups), dim3(threads), 0, st, *a, sd_dev);
    else if (a->D == 128) hipLaunchKernelGGL((qkv_finish_kernel<128>), dim3(M, groups), dim3(threads), 0, st, *a, sd_dev);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
// G <= 8 query heads per KV head, every stream's frozen prefix <= 64 keys (the caller checks both)
hipError_t aha_qkv_finish_attn_static(const QkvFinishArgs* a, const StepDesc* sd_dev, int M, int T, bf16* out, int ldo, float scale, hipStream_t st) {
    StaticAttnArgs p;
    p.qa = *a; p.out = out; p.ldo = ldo; p.scale = scale; p.G = a->Hq / a->Hkv; p.T = T;
    if (p.G > 8) return hipErrorInvalidValue;
    if (a->D == 64) hipLaunchKernelGGL((qkv_finish_attn_static_kernel<64>), dim3(M, a->Hkv), dim3(256), 0, st, p, sd_dev);
    else if (a->D == 128) hipLaunchKernelGGL((qkv_finish_attn_static_kernel<128>), dim3(M, a->Hkv), dim3(256), 0, st, p, sd_dev);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
static int g_rerot_pg = 0;       // tuning "rerot_pg": plane groups of sink_rerotate_kernel (0 = heuristic)
void aha_sink_rerotate_set_pg(int v) { g_rerot_pg = v; }
hipError_t aha_sink_rerotate(const StepDesc* sd_dev, unsigned stream_mask, int n_streams, int nmax, const bf16* rcos, const bf16* rsin,
                             const bf16* cosb, const bf16* sinb, int layers, int Hkv, int D, hipStream_t st) {
    if (nmax == 0 || stream_mask == 0) return hipSuccess;
    const int planes = layers * Hkv;
    int pg = planes < 8 ? planes : 8;                        // plane groups: each thread walks planes / pg planes with one set of coefficients
    if (n_streams >= 4 && planes >= 4) pg = 4;               // enough workgroups from the streams alone
    if (g_rerot_pg > 0) pg = g_rerot_pg < planes ? g_rerot_pg : planes;
    if (D == 64) {
        const int kpb = 256 / (64 / 8);
        hipLaunchKernelGGL((sink_rerotate_kernel<64>), dim3(ceil_div(nmax, kpb), pg, n_streams), dim3(256), 0, st, sd_dev, stream_mask, rcos, rsin, cosb, sinb, layers * Hkv);
    } else if (D == 128) {
        const int kpb = 256 / (128 / 8);
        hipLaunchKernelGGL((sink_rerotate_kernel<128>), dim3(ceil_div(nmax, kpb), pg, n_streams), dim3(256), 0, st, sd_dev, stream_mask, rcos, rsin, cosb, sinb, layers * Hkv);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t aha_cache_update_layer(const StreamStep* ss, int layer, int Hkv, int D, int T, const bf16* knew, const bf16* vnew,
                                   const bf16* rcos, const bf16* rsin, const bf16* cosb, const bf16* sinb, hipStream_t st) {
    const int n = ss->n_rerot > ss->write_count ? ss->n_rerot : ss->write_count;
    if (n <= 0) return hipSuccess;
    if (D == 64) {
        const int kpb = 256 / (64 / 8);
        hipLaunchKernelGGL((cache_update_layer_kernel<64>), dim3(ceil_div(n, kpb), Hkv), dim3(256), 0, st, *ss, layer, Hkv, T, knew, vnew, rcos, rsin, cosb, sinb);
    } else if (D == 128) {
        const int kpb = 256 / (128 / 8);
        hipLaunchKernelGGL((cache_update_layer_kernel<128>), dim3(ceil_div(n, kpb), Hkv), dim3(256), 0, st, *ss, layer, Hkv, T, knew, vnew, rcos, rsin, cosb, sinb);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t aha_repetition_penalty(float* logits, int V, const long* hist, const int* n_hist, float penalty, float* tmp, hipStream_t st) {
    hipLaunchKernelGGL(repetition_penalty_kernel, dim3(1), dim3(1024), 0, st, logits, V, hist, n_hist, penalty, tmp);
    return hipGetLastError();
}
hipError_t aha_generation_bookkeep(const long* tok, long eos, long* hist, int* n_hist, int cap, int use_hist, long* out_ids, int i, hipStream_t st) {
    hipLaunchKernelGGL(generation_bookkeep_kernel, dim3(1), dim3(1), 0, st, tok, eos, hist, n_hist, cap, use_hist, out_ids, i);
    return hipGetLastError();
}
hipError_t aha_heads(const bf16* xn, int ldx, int row_first, int row_step, int count, const bf16* heads_w, int H,
                     float* scores, float* raw, const int* poison, hipStream_t st) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(heads_kernel, dim3(count), dim3(256), 0, st, xn, ldx, row_first, row_step, heads_w, H, scores, raw, poison);
    return hipGetLastError();
}
hipError_t aha_im2col_norm(const uint8_t* frames, int N, int S, int P, int Kp, const float* mean3, const float* std3, bf16* out,
                           hipStream_t st) {
    const int grid = S / P;
    const long total = (long)N * grid * grid * (Kp >> 3);
    PixelNorm pn;
    for (int c = 0; c < 3; ++c) { pn.mean[c] = mean3[c]; pn.std[c] = std3[c]; }
    hipLaunchKernelGGL(im2col_norm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, frames, N, S, P, grid, Kp, pn, out);
    return hipGetLastError();
}

// CLIP embeddings (CLIPVisionEmbeddings): x = cat([class_embedding, patches]) + position_embedding, one bf16 add per element.
// Internal row order per frame: the Np patch tokens first, the class token LAST (the tower has no causal structure, so the order
// is free as long as every token carries its own position row; `pos` is already stored in that order).
__global__ void clip_assemble_kernel(const bf16* __restrict__ patches, const bf16* __restrict__ cls, const bf16* __restrict__ pos,
                                     bf16* __restrict__ x, int n, int Np, int Dv) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int dch = Dv >> 3, T = Np + 1;
    if (gid >= (long)n * T * dch) return;
    const int c = (int)(gid % dch) * 8;
    const long row = gid / dch;
    const int f = (int)(row / T), t = (int)(row % T);
    const bf16x8 a = t < Np ? *reinterpret_cast<const bf16x8*>(patches + ((long)f * Np + t) * Dv + c) : *reinterpret_cast<const bf16x8*>(cls + c);
    const bf16x8 p = *reinterpret_cast<const bf16x8*>(pos + (long)t * Dv + c);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(a[e]) + bf2f(p[e]));
    *reinterpret_cast<bf16x8*>(x + row * Dv + c) = o;
}
hipError_t aha_clip_assemble(const bf16* patches, const bf16* cls, const bf16* pos, bf16* x, int n, int Np, int Dv, hipStream_t st) {
    const long total = (long)n * (Np + 1) * (Dv >> 3);
    hipLaunchKernelGGL(clip_assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, patches, cls, pos, x, n, Np, Dv);
    return hipGetLastError();
}
hipError_t aha_layernorm(const bf16* x, int ldx, const bf16* w, const bf16* b, bf16* out, int ldo, int M, int D, float eps,
                         hipStream_t st) {
    if ((D & 7) || D > 4096) return hipErrorInvalidValue;
    return aha_layernorm_pf(x, ldx, w, b, out, ldo, M, D, eps, nullptr, st);
}
// LayerNorm whose launch also pulls the byte ranges of *pf_ through the caches (riders in front of the row workgroups)
hipError_t aha_layernorm_pf(const bf16* x, int ldx, const bf16* w, const bf16* b, bf16* out, int ldo, int M, int D, float eps,
                            const WeightPrefetch* pf_, hipStream_t st) {
    if ((D & 7) || D > 4096) return hipErrorInvalidValue;
    const int nc = ceil_div(D >> 3, 64);
    WeightPrefetch pf{};
    if (pf_) pf = *pf_;
    long total = 0;
    for (int i = 0; i < 4; ++i) {
        if (!pf.p[i] || pf.bytes[i] < 16) { pf.p[i] = nullptr; pf.bytes[i] = 0; }
        if ((uintptr_t)pf.p[i] & 15) return hipErrorInvalidValue;
        total += pf.bytes[i];
    }
    const bool on = pf.n_riders > 0 && total > 0;
    if (!on) pf.n_riders = 0;
    const dim3 grid(ceil_div(M, 4) + pf.n_riders), blk(256);
    if (on) {
        if (nc <= 2) hipLaunchKernelGGL((layernorm_kernel<2, true>), grid, blk, 0, st, x, ldx, w, b, out, ldo, M, D, eps, pf);
        else if (nc <= 3) hipLaunchKernelGGL((layernorm_kernel<3, true>), grid, blk, 0, st, x, ldx, w, b, out, ldo, M, D, eps, pf);
        else hipLaunchKernelGGL((layernorm_kernel<8, true>), grid, blk, 0, st, x, ldx, w, b, out, ldo, M, D, eps, pf);
    } else {
        if (nc <= 2) hipLaunchKernelGGL((layernorm_kernel<2>), grid, blk, 0, st, x, ldx, w, b, out, ldo, M, D, eps, pf);
        else if (nc <= 3) hipLaunchKernelGGL((layernorm_kernel<3>), grid, blk, 0, st, x, ldx, w, b, out, ldo, M, D, eps, pf);
        else hipLaunchKernelGGL((layernorm_kernel<8>), grid, blk, 0, st, x, ldx, w, b, out, ldo, M, D, eps, pf);
    }
    return hipGetLastError();
}
hipError_t aha_layernorm_kb(const bf16* x, int ldx, const bf16* w, const bf16* b, bf16* out_kb, int M, int D, float eps, hipStream_t st) {
    if ((D & 31) || D > 4096) return hipErrorInvalidValue;
    const int nc = ceil_div(D >> 3, 64);
    const WeightPrefetch pf{};
    const dim3 grid(ceil_div(M, 4)), blk(256);
    if (nc <= 2) hipLaunchKernelGGL((layernorm_kernel<2, false, true>), grid, blk, 0, st, x, ldx, w, b, out_kb, 0, M, D, eps, pf);
    else if (nc <= 3) hipLaunchKernelGGL((layernorm_kernel<3, false, true>), grid, blk, 0, st, x, ldx, w, b, out_kb, 0, M, D, eps, pf);
    else hipLaunchKernelGGL((layernorm_kernel<8, false, true>), grid, blk, 0, st, x, ldx, w, b, out_kb, 0, M, D, eps, pf);
    return hipGetLastError();
}
hipError_t aha_pool(const bf16* in, bf16* out, int N, int g, int go, int H, int stride, int mode, int frame_rows, hipStream_t st) {
    const long total = (long)N * go * go * (H >> 3);
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, out, N, g, go, H, stride, mode,
                       frame_rows > 0 ? frame_rows : g * g);
    return hipGetLastError();
}
hipError_t aha_gather_pool_rows(const bf16* in, bf16* out, int N, int g, int go, int s, int Dv, int frame_rows, hipStream_t st) {
    const long total = (long)N * 4 * go * go * (Dv >> 3);
    hipLaunchKernelGGL(gather_pool_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, out, N, g, go, s, Dv,
                       frame_rows > 0 ? frame_rows : g * g);
    return hipGetLastError();
}
// k-blocked [K/32][M][32] -> row-major [M][ldo] (parity taps of the mid-M path; 16 B per thread)
__global__ void kblocked_to_rows_kernel(const bf16* __restrict__ in, int M, int rows, int K, bf16* __restrict__ out, int ldo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte piece
    if (i >= (long)M * (K >> 3)) return;
    const int c = (int)(i & 3);
    const long rk = i >> 2;
    const int row = (int)(rk % M), kt = (int)(rk / M);
    if (row < rows) *reinterpret_cast<bf16x8*>(out + (long)row * ldo + kt * 32 + c * 8) = *reinterpret_cast<const bf16x8*>(in + i * 8);
}
// panels of M rows, of which the first `rows` are copied out (the layer engine's hand-off panels are 48 rows high whatever the step's M)
hipError_t aha_kblocked_to_rows_n(const bf16* in, int M, int rows, int K, bf16* out, int ldo, hipStream_t st) {
    if (K % 32 || ldo % 8) return hipErrorInvalidValue;
    const long total = (long)M * (K >> 3);
    hipLaunchKernelGGL(kblocked_to_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, M, rows, K, out, ldo);
    return hipGetLastError();
}
hipError_t aha_kblocked_to_rows(const bf16* in, int M, int K, bf16* out, int ldo, hipStream_t st) { return aha_kblocked_to_rows_n(in, M, M, K, out, ldo, st); }
// row-major [rows][ld] (first K columns, K % 32 == 0) -> k-blocked [K/32][rows][32]: what an LDS-DMA piece of 16 rows x 64 B then reads is
// one contiguous KiB instead of sixteen half cache lines (tile-GEMM weights: gemm_tile_p.hip)
__global__ void rows_to_kblocked_kernel(const bf16* __restrict__ in, int rows, int K, int ld, bf16* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte piece of the output
    if (i >= (long)rows * (K >> 3)) return;
    const int c = (int)(i & 3);
    const long rk = i >> 2;
    const int row = (int)(rk % rows), kt = (int)(rk / rows);
    *reinterpret_cast<bf16x8*>(out + i * 8) = *reinterpret_cast<const bf16x8*>(in + (long)row * ld + kt * 32 + c * 8);
}
hipError_t aha_rows_to_kblocked(const bf16* in, int rows, int K, int ld, bf16* out, hipStream_t st) {
    if (K % 32 || ld % 8) return hipErrorInvalidValue;
    const long total = (long)rows * (K >> 3);
    hipLaunchKernelGGL(rows_to_kblocked_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, rows, K, ld, out);
    return hipGetLastError();
}
hipError_t aha_embed_gather(const long* ids, int n, const bf16* table, int H, int vocab, bf16* out, int ldo, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(embed_gather_kernel, dim3(n), dim3(256), 0, st, ids, n, table, H, vocab, out, ldo);
    return hipGetLastError();
}
hipError_t aha_argmax(const float* logits, int ld, int V, int rows, long* out, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, st, logits, ld, V, out);
    return hipGetLastError();
}
}
