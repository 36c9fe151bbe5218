// Token bookkeeping around the transformer stacks (HBM / index bound):
//   * random masking from a given noise tensor  (reference op K3, model/vit_autoenc.py:141-153)
//   * patch gather = im2col restricted to the KEPT patches (op K1 input side, model/vit.py:72-74;
//     the reference embeds all L patches and then discards 75 % of them at vit_autoenc.py:147 —
//     rows of a GEMM are independent, so embedding only the kept ones is the same numbers)
//   * encoder sequence assembly: +pos_embed, cls token (ops K2, K4; vit_autoenc.py:162-170)
//   * decoder sequence assembly: mask-token fill, un-shuffle, +decoder_pos_embed (op K11/K12;
//     vit_autoenc.py:184-190) and both backward scatters.
#include "common.hpp"
#include "vitae_hip.h"

namespace {

// ---------------------------------------------------------------- masking
// rank[i] = #{j : noise[j] < noise[i] or (== and j < i)} is the stable ascending argsort position, so
// ids_restore[i] = rank[i] (= argsort(argsort(noise))), ids_shuffle[rank[i]] = i and mask[i] = rank[i] >= len_keep (0 = keep,
// 1 = remove).  A workgroup owns 32 elements of one sample; eight lanes share an element (each compares it with every eighth j,
// the row sitting in LDS) and fold their counts with three shuffles.  (One workgroup per SAMPLE, one element per thread and
// pass — the first version — left the first launch of the step's dependent chain on 8 CUs: 14 us at L = 216, 320 us at the
// reference's patch-8 default L = 1728.)
__global__ __launch_bounds__(256) void random_masking_kernel(const float* __restrict__ noise, int* __restrict__ ids_shuffle,
                                                             int* __restrict__ ids_restore, float* __restrict__ mask,
                                                             long long* __restrict__ ids_restore64, int L, int len_keep) {
    extern __shared__ float nz[];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < L; i += 256) nz[i] = noise[(long)b * L + i];
    __syncthreads();
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    const bool live = i < L;
    const float v = nz[live ? i : 0];
    int rank = 0;
    for (int j = part; j < L; j += 8) {
        const float u = nz[j];
        rank += (u < v) || (u == v && j < i);
    }
    rank += __shfl_xor(rank, 1, 64);
    rank += __shfl_xor(rank, 2, 64);
    rank += __shfl_xor(rank, 4, 64);
    if (live && part == 0) {
        ids_restore[(long)b * L + i] = rank;
        if (ids_restore64) ids_restore64[(long)b * L + i] = rank;
        ids_shuffle[(long)b * L + rank] = i;
        mask[(long)b * L + i] = rank >= len_keep ? 1.f : 0.f;
    }
}

// ---------------------------------------------------------------- kept-patch gather (im2col rows)
// out[(b*keep + j), c*p^3 + r*p^2 + s*p + q] = vol[b, c, gl*p + r, gh*p + s, gw*p + q]
// for patch l = ids_shuffle[b, j] = (gl, gh, gw): the Conv3d weight's (C, p, p, p) flattening.
// (two views in one launch: samples b >= B1 come from vol2 — the contrastive model's second view, vit_autoenc.py:272,277)
// POW2: p is a power of two (log2p): the (c, r, s, q4) decode of an element index is shifts and masks; the generic path divides.
// Four 16-byte loads are in flight per thread before the first store: a patch row is only 4 p contiguous bytes (64 at p = 16), so
// the kernel lives on memory-level parallelism (round 5: one load in flight behind ~100 instructions of index division per element
// took 95 us for the 113 MB of a batch-32 step).
template <bool POW2>
__global__ __launch_bounds__(256) void gather_patches_kernel(const float* __restrict__ vol, const float* __restrict__ vol2, int B1,
                                                             const int* __restrict__ ids_shuffle,
                                                             float* __restrict__ out, __bf16* __restrict__ out16,
                                                             int C, int Lz, int Hy, int Wx, int p, int log2p,
                                                             int g1, int g2, int L, int keep) {
    const int j = blockIdx.x, b = blockIdx.y;
    const int l = ids_shuffle[(long)b * L + j];
    const int gl = l / (g1 * g2), gh = (l / g2) % g1, gw = l % g2;
    const int p4 = p / 4;
    const int P4 = C * p * p * p4;
    const long vstride = (long)Lz * Hy * Wx;
    const float* vb = b < B1 ? vol + (long)b * C * vstride : vol2 + (long)(b - B1) * C * vstride;
    const long rowoff = ((long)b * keep + j) * ((long)C * p * p * p);
    const float* pb = vb + ((long)(gl * p) * Hy + gh * p) * Wx + gw * p;      // the patch's first voxel (channel 0)
    auto src_of = [&](int i) -> const float* {
        int q4, sidx, r, c;
        if (POW2) {
            q4 = i & (p4 - 1); const int t = i >> (log2p - 2);
            sidx = t & (p - 1); r = (t >> log2p) & (p - 1); c = t >> (2 * log2p);
        } else {
            q4 = i % p4; sidx = (i / p4) % p; r = (i / (p4 * p)) % p; c = i / (p4 * p * p);
        }
        return pb + c * vstride + ((long)r * Hy + sidx) * Wx + q4 * 4;
    };
    // blockIdx.z splits a patch row over several workgroups (keep * B alone is only ~200 of them)
    constexpr int U = 4;
    const int stride = 256 * gridDim.z;
    for (int i0 = blockIdx.z * 256 + threadIdx.x; i0 < P4; i0 += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f32x4*>(src_of(min(i0 + u * stride, P4 - 1)));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * stride;
            if (i < P4) {
                if (out) *reinterpret_cast<f32x4*>(out + rowoff + (long)i * 4) = v[u];
                if (out16) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[u][e];
                    *reinterpret_cast<bf16x4*>(out16 + rowoff + (long)i * 4) = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- encoder sequence assembly
// x[b, 0]     = cls_token + pos_embed[0]
// x[b, 1 + j] = tok[b*keep + j] + pos_embed[1 + ids_shuffle[b, j]]       (tok already has the conv bias)
// out[d] = a[d] + b[d] over one row; V: D % 4 == 0 and 16-byte aligned rows -> 16 bytes per lane
template <bool V>
__device__ __forceinline__ void row_add(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, int D) {
    if (V) {
        for (int d = threadIdx.x; d < D / 4; d += 256) {
            const f32x4 u = reinterpret_cast<const f32x4*>(a)[d], w = reinterpret_cast<const f32x4*>(b)[d];
            reinterpret_cast<f32x4*>(out)[d] = f32x4{u[0] + w[0], u[1] + w[1], u[2] + w[2], u[3] + w[3]};
        }
    } else {
        for (int d = threadIdx.x; d < D; d += 256) out[d] = a[d] + b[d];
    }
}

// out[d] = src[d] (+ optional bf16 copy) over one row
template <bool V>
__device__ __forceinline__ void row_copy(float* __restrict__ out, __bf16* __restrict__ out16, const float* __restrict__ src, int D) {
    if (V) {
        for (int d = threadIdx.x; d < D / 4; d += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(src)[d];
            if (out) reinterpret_cast<f32x4*>(out)[d] = v;
            if (out16) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                reinterpret_cast<bf16x4*>(out16)[d] = o;
            }
        }
    } else {
        for (int d = threadIdx.x; d < D; d += 256) {
            const float v = src[d];
            if (out) out[d] = v;
            if (out16) out16[d] = (__bf16)v;
        }
    }
}

template <bool V>
__global__ __launch_bounds__(256) void encoder_assemble_fwd_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                                                   const float* __restrict__ pos, const int* __restrict__ ids_shuffle,
                                                                   float* __restrict__ x, int L, int keep, int D) {
    const int t = blockIdx.x, b = blockIdx.y;   // t in [0, keep]
    float* xr = x + ((long)b * (keep + 1) + t) * D;
    if (t == 0) {
        row_add<V>(xr, cls, pos, D);
    } else {
        const int l = ids_shuffle[(long)b * L + t - 1];
        row_add<V>(xr, tok + ((long)b * keep + t - 1) * D, pos + (long)(1 + l) * D, D);
    }
}

// dtok[b*keep + j] = dx[b, 1 + j];  dcls += sum_b dx[b, 0]
template <bool V>
__global__ __launch_bounds__(256) void encoder_assemble_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dtok,
                                                                   __bf16* __restrict__ dtok16,
                                                                   float* __restrict__ dcls, int B, int keep, int D) {
    const int t = blockIdx.x, b = blockIdx.y;
    const float* dr = dx + ((long)b * (keep + 1) + t) * D;
    if (t == 0) {
        if (b != 0) return;   // block (0,0) sums the cls rows of every sample
        // (summed in sample order: deterministic; eight loads in flight — one at a time was a chain of B memory latencies, 47 us at
        // batch 32 for 0.2 MB)
        for (int d = threadIdx.x; d < D; d += 256) {
            float s = 0.f;
            int bb = 0;
            for (; bb + 8 <= B; bb += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = dx[((long)(bb + u) * (keep + 1)) * D + d];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; bb < B; ++bb) s += dx[((long)bb * (keep + 1)) * D + d];
            atomicAdd(dcls + d, s);
        }
    } else {
        const long oo = ((long)b * keep + t - 1) * D;
        row_copy<V>(dtok ? dtok + oo : nullptr, dtok16 ? dtok16 + oo : nullptr, dr, D);
    }
}

// ---------------------------------------------------------------- decoder sequence assembly
// e = decoder_embed(latent) [B, keep+1, Dd].
// xd[b, 0]     = e[b, 0] + dpos[0]
// xd[b, 1 + l] = (r = ids_restore[b, l]) < keep ? e[b, 1 + r] : mask_token) + dpos[1 + l]
template <bool V>
__global__ __launch_bounds__(256) void decoder_assemble_fwd_kernel(const float* __restrict__ e, const float* __restrict__ mask_token,
                                                                   const float* __restrict__ dpos, const int* __restrict__ ids_restore,
                                                                   float* __restrict__ xd, int L, int keep, int Dd) {
    const int t = blockIdx.x, b = blockIdx.y;   // t in [0, L]
    float* xr = xd + ((long)b * (L + 1) + t) * Dd;
    const float* pr = dpos + (long)t * Dd;
    const float* src;
    if (t == 0) src = e + ((long)b * (keep + 1)) * Dd;
    else {
        const int r = ids_restore[(long)b * L + t - 1];
        src = r < keep ? e + ((long)b * (keep + 1) + 1 + r) * Dd : mask_token;
    }
    row_add<V>(xr, src, pr, Dd);
}

// de[b, 0] = dxd[b, 0];  de[b, 1 + r] = dxd[b, 1 + ids_shuffle[b, r]]  (r < keep)
// One launch, two kinds of workgroup.  The first (keep + 1) * B gather the kept rows of dxd back into de (fp32, and
// optionally a bf16 copy: the dy operand of decoder_embed's backward GEMMs).  The rest sum dxd over the masked
// positions into dmask_token: threads own columns, each workgroup takes `per_block` entries of the (b, masked-rank) list.
template <bool V>
__global__ __launch_bounds__(256) void decoder_assemble_bwd_kernel(const float* __restrict__ dxd, const int* __restrict__ ids_shuffle,
                                                                   float* __restrict__ de, __bf16* __restrict__ de16,
                                                                   float* __restrict__ dmask, int B, int L, int keep, int Dd,
                                                                   int per_block, int col_blocks) {
    const int nrow = (keep + 1) * B;
    if ((int)blockIdx.x < nrow) {
        const int t = blockIdx.x % (keep + 1), b = blockIdx.x / (keep + 1);   // t in [0, keep]
        const int srow = t == 0 ? 0 : 1 + ids_shuffle[(long)b * L + t - 1];
        const float* s = dxd + ((long)b * (L + 1) + srow) * Dd;
        const long o = ((long)b * (keep + 1) + t) * Dd;
        row_copy<V>(de + o, de16 ? de16 + o : nullptr, s, Dd);
        return;
    }
    const int mb = blockIdx.x - nrow;
    const int d = (mb % col_blocks) * 256 + threadIdx.x;
    if (d >= Dd) return;
    const int nm = L - keep;
    const int i0 = (mb / col_blocks) * per_block, i1 = min(B * nm, i0 + per_block);
    float s = 0.f;
    int i = i0;
    for (; i + 4 <= i1; i += 4) {                     // four rows in flight (summed in row order)
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = (i + u) / nm, r = keep + (i + u) % nm;
            const int l = ids_shuffle[(long)b * L + r];
            v[u] = dxd[((long)b * (L + 1) + 1 + l) * Dd + d];
        }
        s += v[0]; s += v[1]; s += v[2]; s += v[3];
    }
    for (; i < i1; ++i) {
        const int b = i / nm, r = keep + i % nm;
        const int l = ids_shuffle[(long)b * L + r];
        s += dxd[((long)b * (L + 1) + 1 + l) * Dd + d];
    }
    atomicAdd(dmask + d, s);
}

// out[b, d] = mean over tokens n = first .. N-1 of x[b, n, d]  (global pool without the cls token,
// reference model/vit.py:277-278).  One thread per (b, d); rows are read coalesced, summed in token order.
__global__ __launch_bounds__(256) void mean_pool_tokens_kernel(const float* __restrict__ x, float* __restrict__ out, int N,
                                                               int D, int first) {
    const int d = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (d >= D) return;
    const float* xb = x + ((long)b * N + first) * D + d;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int cnt = N - first;
    int n = 0;
    for (; n + 4 <= cnt; n += 4) {
        s0 += xb[(long)n * D]; s1 += xb[(long)(n + 1) * D]; s2 += xb[(long)(n + 2) * D]; s3 += xb[(long)(n + 3) * D];
    }
    for (; n < cnt; ++n) s0 += xb[(long)n * D];
    out[(long)b * D + d] = ((s0 + s1) + (s2 + s3)) / (float)cnt;
}

}  // namespace

extern "C" int vitae_random_masking(const float* noise, int* ids_shuffle, int* ids_restore, float* mask,
                                    long long* ids_restore_i64, int B, int L, int len_keep, void* stream) {
    if (!noise || !ids_shuffle || !ids_restore || !mask || B <= 0 || L <= 0 || len_keep < 0 || len_keep > L)
        return VITAE_ERR_INVALID_ARG;
    if ((size_t)L * 4 > 60000) return VITAE_ERR_UNSUPPORTED_SHAPE;
    hipLaunchKernelGGL(random_masking_kernel, dim3(cdiv(L, 32), B), dim3(256), (size_t)L * 4, (hipStream_t)stream, noise,
                       ids_shuffle, ids_restore, mask, ids_restore_i64, L, len_keep);
    return vitae_launch_status();
}

extern "C" int vitae_gather_patches(const float* vol, const int* ids_shuffle, float* out, void* out_bf16, int B, int C,
                                    int Lz, int Hy, int Wx, int p, int keep, void* stream) {
    if (!vol || !ids_shuffle || (!out && !out_bf16) || B <= 0 || C <= 0 || p <= 0 || keep <= 0) return VITAE_ERR_INVALID_ARG;
    if ((p & 3) || Lz % p || Hy % p || Wx % p || ((uintptr_t)vol & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const int g0 = Lz / p, g1 = Hy / p, g2 = Wx / p;
    const int P4 = C * p * p * (p / 4);
    int zsplit = (keep * B < 1024) ? cdiv(1024, keep * B) : 1;
    if (zsplit > cdiv(P4, 256)) zsplit = cdiv(P4, 256);
    const bool pow2 = (p & (p - 1)) == 0;
    const int log2p = pow2 ? __builtin_ctz(p) : 0;
    if (pow2) hipLaunchKernelGGL(gather_patches_kernel<true>, dim3(keep, B, zsplit), dim3(256), 0, (hipStream_t)stream, vol, vol, B, ids_shuffle, out,
                                 reinterpret_cast<__bf16*>(out_bf16), C, Lz, Hy, Wx, p, log2p, g1, g2, g0 * g1 * g2, keep);
    else hipLaunchKernelGGL(gather_patches_kernel<false>, dim3(keep, B, zsplit), dim3(256), 0, (hipStream_t)stream, vol, vol, B, ids_shuffle, out,
                            reinterpret_cast<__bf16*>(out_bf16), C, Lz, Hy, Wx, p, log2p, g1, g2, g0 * g1 * g2, keep);
    return vitae_launch_status();
}

// both views of the contrastive model in one launch: rows [0, B*keep) from vol1 with ids_shuffle[0:B], rows [B*keep, 2*B*keep)
// from vol2 with ids_shuffle[B:2B]
extern "C" int vitae_gather_patches_2views(const float* vol1, const float* vol2, const int* ids_shuffle, float* out, void* out_bf16,
                                           int B, int C, int Lz, int Hy, int Wx, int p, int keep, void* stream) {
    if (!vol1 || !vol2 || !ids_shuffle || (!out && !out_bf16) || B <= 0 || C <= 0 || p <= 0 || keep <= 0) return VITAE_ERR_INVALID_ARG;
    if ((p & 3) || Lz % p || Hy % p || Wx % p || ((uintptr_t)vol1 & 15) || ((uintptr_t)vol2 & 15)) return VITAE_ERR_UNSUPPORTED_SHAPE;
    const int g0 = Lz / p, g1 = Hy / p, g2 = Wx / p;
    const int P4 = C * p * p * (p / 4);
    int zsplit = (keep * 2 * B < 1024) ? cdiv(1024, keep * 2 * B) : 1;
    if (zsplit > cdiv(P4, 256)) zsplit = cdiv(P4, 256);
    const bool pow2 = (p & (p - 1)) == 0;
    const int log2p = pow2 ? __builtin_ctz(p) : 0;
    if (pow2) hipLaunchKernelGGL(gather_patches_kernel<true>, dim3(keep, 2 * B, zsplit), dim3(256), 0, (hipStream_t)stream, vol1, vol2, B, ids_shuffle,
                                 out, reinterpret_cast<__bf16*>(out_bf16), C, Lz, Hy, Wx, p, log2p, g1, g2, g0 * g1 * g2, keep);
    else hipLaunchKernelGGL(gather_patches_kernel<false>, dim3(keep, 2 * B, zsplit), dim3(256), 0, (hipStream_t)stream, vol1, vol2, B, ids_shuffle,
                            out, reinterpret_cast<__bf16*>(out_bf16), C, Lz, Hy, Wx, p, log2p, g1, g2, g0 * g1 * g2, keep);
    return vitae_launch_status();
}

extern "C" int vitae_encoder_assemble_fwd(const float* tok, const float* cls_token, const float* pos_embed,
                                          const int* ids_shuffle, float* x, int B, int L, int keep, int D,
                                          void* stream) {
    if (!tok || !cls_token || !pos_embed || !ids_shuffle || !x || B <= 0) return VITAE_ERR_INVALID_ARG;
    const bool v = !(D & 3) && !(((uintptr_t)tok | (uintptr_t)cls_token | (uintptr_t)pos_embed | (uintptr_t)x) & 15);
    if (v) hipLaunchKernelGGL(encoder_assemble_fwd_kernel<true>, dim3(keep + 1, B), dim3(256), 0, (hipStream_t)stream, tok,
                              cls_token, pos_embed, ids_shuffle, x, L, keep, D);
    else hipLaunchKernelGGL(encoder_assemble_fwd_kernel<false>, dim3(keep + 1, B), dim3(256), 0, (hipStream_t)stream, tok,
                            cls_token, pos_embed, ids_shuffle, x, L, keep, D);
    return vitae_launch_status();
}

extern "C" int vitae_encoder_assemble_bwd(const float* dx, float* dtok, void* dtok_bf16, float* dcls, int B, int keep,
                                          int D, void* stream) {
    if (!dx || (!dtok && !dtok_bf16) || !dcls || B <= 0) return VITAE_ERR_INVALID_ARG;
    const bool v = !(D & 3) && !(((uintptr_t)dx | (uintptr_t)dtok) & 15) && !((uintptr_t)dtok_bf16 & 7);
    if (v) hipLaunchKernelGGL(encoder_assemble_bwd_kernel<true>, dim3(keep + 1, B), dim3(256), 0, (hipStream_t)stream, dx, dtok,
                              reinterpret_cast<__bf16*>(dtok_bf16), dcls, B, keep, D);
    else hipLaunchKernelGGL(encoder_assemble_bwd_kernel<false>, dim3(keep + 1, B), dim3(256), 0, (hipStream_t)stream, dx, dtok,
                            reinterpret_cast<__bf16*>(dtok_bf16), dcls, B, keep, D);
    return vitae_launch_status();
}

extern "C" int vitae_decoder_assemble_fwd(const float* e, const float* mask_token, const float* dpos,
                                          const int* ids_restore, float* xd, int B, int L, int keep, int Dd,
                                          void* stream) {
    if (!e || !mask_token || !dpos || !ids_restore || !xd || B <= 0) return VITAE_ERR_INVALID_ARG;
    const bool v = !(Dd & 3) && !(((uintptr_t)e | (uintptr_t)mask_token | (uintptr_t)dpos | (uintptr_t)xd) & 15);
    if (v) hipLaunchKernelGGL(decoder_assemble_fwd_kernel<true>, dim3(L + 1, B), dim3(256), 0, (hipStream_t)stream, e, mask_token,
                              dpos, ids_restore, xd, L, keep, Dd);
    else hipLaunchKernelGGL(decoder_assemble_fwd_kernel<false>, dim3(L + 1, B), dim3(256), 0, (hipStream_t)stream, e, mask_token,
                            dpos, ids_restore, xd, L, keep, Dd);
    return vitae_launch_status();
}

extern "C" int vitae_decoder_assemble_bwd(const float* dxd, const int* ids_shuffle, float* de, void* de_bf16,
                                          float* dmask_token, int B, int L, int keep, int Dd, void* stream) {
    if (!dxd || !ids_shuffle || !de || !dmask_token || B <= 0) return VITAE_ERR_INVALID_ARG;
    const int total = B * (L - keep), per_block = 32, col_blocks = cdiv(Dd, 256);
    const int blocks = (keep + 1) * B + col_blocks * cdiv(total, per_block);
    const bool v = !(Dd & 3) && !(((uintptr_t)dxd | (uintptr_t)de) & 15) && !((uintptr_t)de_bf16 & 7);
    if (v) hipLaunchKernelGGL(decoder_assemble_bwd_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dxd, ids_shuffle, de,
                              reinterpret_cast<__bf16*>(de_bf16), dmask_token, B, L, keep, Dd, per_block, col_blocks);
    else hipLaunchKernelGGL(decoder_assemble_bwd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dxd, ids_shuffle, de,
                            reinterpret_cast<__bf16*>(de_bf16), dmask_token, B, L, keep, Dd, per_block, col_blocks);
    return vitae_launch_status();
}

extern "C" int vitae_mean_pool_tokens(const float* x, float* out, int B, int N, int D, int first, void* stream) {
    if (!x || !out || B <= 0 || D <= 0 || first < 0 || first >= N) return VITAE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(mean_pool_tokens_kernel, dim3(cdiv(D, 256), B), dim3(256), 0, (hipStream_t)stream, x, out, N, D, first);
    return vitae_launch_status();
}
