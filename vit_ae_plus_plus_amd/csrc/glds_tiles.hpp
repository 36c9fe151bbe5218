// LDS-DMA operand tiles shared by the bf16 GEMM kernels (gemm_glds.hip) and the fused MLP kernels (mlp_fused.hip):
// 64-deep k-tiles moved HBM -> LDS with global_load_lds (16 B per lane, no registers), bank-conflict-free through a
// permutation of the SOURCE addresses, read back as MFMA fragments with ds_read_b128 (k-contiguous operands) or the
// transposing ds_read_b64_tr_b16 (row-contiguous operands).
#pragma once
#include "common.hpp"

namespace vglds {

typedef short s16x4 __attribute__((ext_vector_type(4)));
#ifndef VITAE_GLDS_NS
#define VITAE_GLDS_NS 3
#endif
#ifndef VITAE_DMA_BUFFER
#define VITAE_DMA_BUFFER 1       // LDS-DMA pieces as buffer_load ... lds (wave-uniform descriptor + ONE 32-bit byte offset per lane) instead of
                                 // global_load_lds with 64-bit lane addresses: batch 32 10.73 -> 10.50 ms, batch 4 4.19 -> 4.16 (alternating);
                                 // operands must stay below 2 GiB (checked by the launchers)
#endif
constexpr int BK = 64, NS = VITAE_GLDS_NS;

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }


// XOR applied to a line's 16-byte chunk index (by the DMA through the SOURCE address, by the fragment reads directly).
// The hardware serves a wave's LDS read in phases of 256 B, so the lanes of one phase must cover all 64 banks once:
//   * k-contiguous tile (128-B lines, ds_read_b128, a phase = 16 lanes = 16 consecutive rows of one chunk column):
//     rows of equal parity share their 32 banks, so the 8 such rows of a phase need 8 different slots -> (line >> 1) & 7
//     (the earlier `line & 7` repeated every 8 rows: 2-way conflicts, 50 % of the LDS cycles by SQ_LDS_BANK_CONFLICT);
//   * row-contiguous tile read with ds_read_b64_tr_b16 (a phase = 32 lanes = 4 consecutive k lines x 64 contiguous
//     bytes): 128-B lines -> lines k, k + 2 share banks, flip the 64-byte half with bit 1 of k; 256-B lines -> all four
//     lines share the 64 banks, give each its own 64-byte quarter, (k & 3) << 2.
template <bool KC, int LINE_CH> __device__ __forceinline__ int swz(int line) {
    if (KC) return (line >> 1) & 7;
    return LINE_CH == 8 ? ((line >> 1) & 1) << 2 : (line & 3) << 2;
}

// One LDS-DMA instruction (1 KB per wave) of an operand tile (ROWS rows x 64 k, bf16) into `lds` (byte address, tile base):
// piece j of this wave, j < pieces<ROWS, KC, NW>().  KC tile image: [row][8 chunks]; !KC image: [k][ROWS/8 chunks];
// chunk slot = chunk ^ swz(line).  A kernel with ONE workgroup per CU interleaves the pieces with its MFMAs by hand: a
// piece occupies the wave for ~16-64 clocks while the texture addresser takes its 64 lanes, the matrix pipe works beside it.
template <int ROWS, bool KC, int NW> constexpr int pieces() { return (KC ? ROWS : BK) / (64 / (KC ? BK / 8 : ROWS / 8)) / NW; }

template <int ROWS, bool KC, int NW>
__device__ __forceinline__ void dma_piece(const __bf16* __restrict__ P, long ld, int rows, int r0, int k0,
                                          unsigned char* lds, int wave, int lane, int j) {
    constexpr int LINE_CH = KC ? BK / 8 : ROWS / 8;       // 8 (128 B) or 16 (256 B) chunks of 16 B per LDS line
    constexpr int LPI = 64 / LINE_CH;                     // lines per wave-instruction (1 KB)
    constexpr int NI = pieces<ROWS, KC, NW>();
    static_assert(NI >= 1, "tile too small for this many waves");
    const int inst = wave * NI + j;
    const int line = inst * LPI + lane / LINE_CH;
    const int slot = lane % LINE_CH;
    const int chunk = slot ^ swz<KC, LINE_CH>(line);
    long off;
    if (KC) {
        const int gr = min(r0 + line, rows - 1);          // rows past the operand: any valid row (never stored)
        off = (long)gr * ld + k0 + chunk * 8;
    } else {
        const int gr = min(r0 + chunk * 8, rows - 8);
        off = (long)(k0 + line) * ld + gr;
    }
#if VITAE_DMA_BUFFER
    // buffer form: a wave-uniform descriptor + ONE 32-bit byte offset per lane (half the address traffic of the 64-bit flat form)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(P), 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + inst * 1024), 16, (int)(off * 2), 0, 0, 0);
#else
    __builtin_amdgcn_global_load_lds(P + off, (__attribute__((address_space(3))) void*)(lds + inst * 1024), 16, 0, 0);
#endif
}

// The whole tile: all pieces of this wave back to back.
template <int ROWS, bool KC, int NW>
__device__ __forceinline__ void dma_tile(const __bf16* __restrict__ P, long ld, int rows, int r0, int k0,
                                         unsigned char* lds, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < pieces<ROWS, KC, NW>(); ++j) dma_piece<ROWS, KC, NW>(P, ld, rows, r0, k0, lds, wave, lane, j);
}

// Transposing LDS read (8 bytes per lane), issued as INLINE ASM on purpose.  Through the builtin
// (__builtin_amdgcn_ds_read_tr16_b64_v4i16) hipcc (ROCm 7.2) treats the read as possibly aliasing every LDS-DMA in
// flight and puts `s_waitcnt vmcnt(0)` in front of it — in a k-loop that keeps two or three tiles in flight this drains
// the whole DMA queue every step (seen in the ISA of every row-contiguous-operand GEMM of round 1: dgrad, wgrad, the paired
// launch).  The compiler cannot see an asm read, so (a) no such wait is generated, and (b) the CALLER must order the result:
// call frags_ready() once after the last fragment read of a step and tie every fragment with frag_tie() before its first use.
__device__ __forceinline__ s16x4 lds_read_tr16_b64(unsigned addr) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
__device__ __forceinline__ void frags_ready() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void frag_tie(bf16x8& f) { asm volatile("" : "+v"(f)); }

// MFMA operand fragment (32 rows x 16 k): rows row0 + (lane & 31), k-slots kk*16 + 8*hi + e
template <int ROWS, bool KC>
__device__ __forceinline__ bf16x8 frag(const unsigned char* T, int row0, int kk, int lane) {
    if (KC) {
        const int r = row0 + (lane & 31), c = 2 * kk + (lane >> 5);
        return *reinterpret_cast<const bf16x8*>(T + r * 128 + ((c ^ swz<true, 8>(r)) << 4));
    } else {
        constexpr int LB = ROWS * 2;
        const int gg = lane >> 4, li = lane & 15;
        const int kl = 8 * (gg >> 1) + (li >> 2);                 // lane part of k; kk * 16 and the +4 of the second read
        const int col = row0 + 16 * (gg & 1) + 4 * (li & 3);      // leave the swizzle (bits 0-1 of k) unchanged
        const int c = col >> 3, w = (col & 7) * 2;
        typedef __attribute__((address_space(3))) const unsigned char lds_u8;
        const unsigned base = (unsigned)(uintptr_t)(lds_u8*)T + kl * LB + ((c ^ swz<false, ROWS / 8>(kl)) << 4) + w;
        union { s16x4 s[2]; bf16x8 b; } u;
        u.s[0] = lds_read_tr16_b64(base + kk * 16 * LB);
        u.s[1] = lds_read_tr16_b64(base + kk * 16 * LB + 4 * LB);
        return u.b;
    }
}

// The same fragment with EVERY LDS read as inline asm (the k-contiguous ds_read_b128 too): hipcc then tracks none of a loop's
// reads and inserts no waits of its own — in a loop whose header merges a preheader with scalar loads still pending, its lgkmcnt
// bookkeeping falls back to lgkmcnt(0) in front of the first use, which drains the reads issued for the NEXT k-slices (seen in the
// ISA of the wave-specialised kernel).  The caller counts: frag_asm_reads<KC>() LDS instructions per fragment, in issue order.
__device__ __forceinline__ bf16x8 lds_read_b128_asm(unsigned addr) {
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
template <bool KC> constexpr int frag_asm_reads() { return KC ? 1 : 2; }
template <int ROWS, bool KC>
__device__ __forceinline__ bf16x8 frag_asm(const unsigned char* T, int row0, int kk, int lane) {
    if constexpr (KC) {
        typedef __attribute__((address_space(3))) const unsigned char lds_u8;
        const int r = row0 + (lane & 31), c = 2 * kk + (lane >> 5);
        return lds_read_b128_asm((unsigned)(uintptr_t)(lds_u8*)T + r * 128 + ((c ^ swz<true, 8>(r)) << 4));
    } else {
        return frag<ROWS, false>(T, row0, kk, lane);
    }
}

// The same fragments addressed as (stage base) + (per-lane offset that does not depend on the stage: computed ONCE) + (immediate):
// one v_add per pair of reads in the k-loop instead of 1.5-2 address instructions per read — the wave-specialised consumers issue
// their reads in the shadow of an MFMA, where only ~5 instructions fit (round 5).
//   KC image ([row][8 chunks], slot = chunk ^ swz(row)): the k-slice enters through the XOR, so one offset per slice (4 VGPRs);
//     the fragment 64 rows further down is + 8192 bytes (swz has a period of 16 rows): an immediate.
//   row-contiguous image ([k][ROWS / 8 chunks]): the k-slice is additive (kk * 16 lines, + 4 lines for the second read): immediates;
//     the fragment 64 columns further flips a lane-dependent bit of the slot: a second offset (2 VGPRs).
template <int ROWS, bool KC> struct FragOff { unsigned o[KC ? 4 : 2]; };
template <int ROWS, bool KC>
__device__ __forceinline__ FragOff<ROWS, KC> frag_offsets(int row0, int lane) {
    FragOff<ROWS, KC> f;
    if constexpr (KC) {
        const int r = row0 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) f.o[kk] = r * 128 + (((2 * kk + (lane >> 5)) ^ swz<true, 8>(r)) << 4);
    } else {
        constexpr int LB = ROWS * 2;
        const int gg = lane >> 4, li = lane & 15;
        const int kl = 8 * (gg >> 1) + (li >> 2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = row0 + 64 * h + 16 * (gg & 1) + 4 * (li & 3);
            f.o[h] = kl * LB + (((col >> 3) ^ swz<false, ROWS / 8>(kl)) << 4) + (col & 7) * 2;
        }
    }
    return f;
}
// fragment (k-slice KK, row half H) of the tile at LDS byte address `base`; KC: base must already include f.o[KK] (frag_base)
template <int ROWS, bool KC, int KK>
__device__ __forceinline__ unsigned frag_base(unsigned base, const FragOff<ROWS, KC>& f, int h) { return base + (KC ? f.o[KK] : f.o[h]); }
template <int ROWS, bool KC, int KK, int H>
__device__ __forceinline__ bf16x8 frag_rd(unsigned a) {
    if constexpr (KC) {
        bf16x8 r;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(H * 8192));
        return r;
    } else {
        constexpr int LB = ROWS * 2;
        union { s16x4 s[2]; bf16x8 b; } u;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.s[0]) : "v"(a), "n"(KK * 16 * LB));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.s[1]) : "v"(a), "n"(KK * 16 * LB + 4 * LB));
        return u.b;
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


}  // namespace vglds
