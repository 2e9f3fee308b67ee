// Coset LDE kernels (see the header of ntt.hip for the block-twiddle formulation).
//
// Execution: up to 4 stages at a time are done in registers (a thread owns the 16 elements of a
// radix-16 group), with one LDS exchange between such rounds.  LDS image: element i lives at word
// i + (i >> 5).  On gfx950 a ds_read_b32 / ds_write_b32 is served in two 32-lane groups over 32
// banks ((byte address / 4) mod 32), so an access is conflict-free when the 32 lanes of a half-wave
// hit 32 distinct words mod 32.  With one pad word per 32:
//   - 32 consecutive elements from a multiple of 32 (tile loads, rounds at distance >= 32): one pad
//     value for the run -> 32 consecutive banks;
//   - the 16 g + q pattern of the distance-1 round (lane = group g): 16 g + (g >> 1) mod 32 takes
//     every value once over 32 consecutive g;
//   - the distance-16 round (16 lanes = consecutive elements, the other 16 = another block `hi`):
//     a block step moves the bank by 2^(K-1) mod 32, so the two halves of a half-wave take blocks
//     2^(5-K) apart (radix_round swaps two bits of the lane -> group map for that);
//   - the 16-byte chunk loads / stores (lane t: words 4 t + j): 4 t + j + (t >> 3) mod 32, distinct.
// (Rounds 1-3 padded one word per 16, right for 16-lane groups: rocprofv3 showed
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.33 / 0.50 / 0.40 for the three big kernels,
// profiles/r03_config3_sq_counters.txt -- lanes 0 and 31 of every 32-element run met on one bank.)
// Three kernels:
//   k_intt_contig    stages log_n-1 .. sA of the inverse on 4096-element chunks   (only n > 4096)
//   k_lde_mid        the strided stages of the inverse (sA-1 .. 0), then for every coset: scale
//                    coefficient k by s_beta^k / n and run the strided stages of the forward
//                    transform, writing coset block beta -- the coefficients never touch HBM
//   k_lde_fwd_contig stages sA .. log_n-1 of the forward transform, in place on 4096-element chunks
// For n <= 4096 k_lde_mid alone does everything.
//
// The contiguous chunk is 2^LM elements, LM = 12 by default.  For n = 2^21 and 2^22 it grows to
// 2^13 / 2^14 (LM = log_n - 8) so that the strided pass keeps the shape it has at n = 2^20 -- 8
// strided stages on 256-row x 128-byte tiles, two 512-thread workgroups per CU, one LDS round per
// coset (PLAN 1) -- instead of 9 / 10 stages on tiles whose rows are 64 bytes wide and whose
// 1024-thread workgroup has a CU to itself (round 2, 2^22 x 64 at log_blowup 4: that pass ran at VALU
// busy 0.67 with 44 % of its wave-cycles parked at barriers and scale-table loads).  The extra one
// or two stages go to the contiguous passes as radix-32 register rounds (13 = 5+4+4, 14 = 5+5+4):
// still three LDS round trips per chunk.
#include <stdio.h>
#include <stdlib.h>

#include <initializer_list>

#include "kernels.hpp"

namespace ts {

constexpr int LOG_M = 12;          // default contiguous chunk = 4096 elements
constexpr int NT_MID = 512;        // threads per workgroup, middle kernel
// threads per workgroup of the contiguous kernels: 16 elements per thread at LM = 12, 32 above
constexpr int chunk_threads(int lm) { return lm == 14 ? 512 : 256; }
constexpr int TILE_ELEMS = 8192;   // strided tile (generic plan)

__device__ __forceinline__ uint32_t pad(uint32_t i) { return i + (i >> 5); }
constexpr int padded(int n) { return n + (n >> 5); }

// The LDS image `s` (padded) holds 2^log_total elements: a transform of 2^log_len points whose
// element e occupies the 2^log_T consecutive slots [e << log_T, (e+1) << log_T) (log_T = 0 for a
// contiguous chunk; for a strided tile the 2^log_T slots are independent side-by-side transforms
// that share their twiddles).  Local stage u (distance 2^(log_total-1-u) slots) is global stage
// s_base + u, whose block `blk` uses W[2^(s_base+u) + (c << u) + blk].
//
// One round = K consecutive stages u0 .. u0+K-1 on register groups of 2^K elements spaced by the
// distance 2^log_dl of the round's last stage.  LOG_DL >= 0 fixes that distance at compile time
// (LDS addresses become base + immediate offsets); LOG_DL = -1 takes it from the arguments.
// The K stages of one register group v[0 .. 2^K): global stages s_base + u0 .. + K - 1 on the group
// whose block index at stage u0 is `hi` (chunk c).  Values stay in the lazy range [0, 2p).
template <int K, bool INV, bool TOP>
__device__ __forceinline__ void radix_butterflies(uint32_t (&v)[1 << K], unsigned s_base, unsigned u0,
                                                  uint32_t c, uint32_t hi,
                                                  const uint32_t* __restrict__ W) {
    constexpr int R = 1 << K;
    if (!INV) {
#pragma unroll
        for (int d = 0; d < K; d++) {
            const int half = R >> (d + 1);
            // ONE address per stage: the 2^d twiddles of this group are consecutive words, read at
            // immediate offsets from wp (indexing W[wb + j] made the compiler rebuild a 64-bit
            // address for every j: ~3 VALU instructions per twiddle, 8 % of a contiguous pass)
            const uint32_t* __restrict__ wp = W + ((1u << (s_base + u0 + d)) + (c << (u0 + d)) + (hi << d));
#pragma unroll
            for (int q = 0; q < R; q++) {
                if ((q & half) == 0) {
                    // lazy range: inputs and outputs in [0, 2p) (2p < 2^32), 10 VALU
                    // instructions instead of 11: a -> [0, p), t in [0, p), a + t and a - t + p
                    const uint32_t a = red2p(v[q]);
                    uint32_t t = v[q + half];
                    if (TOP && (q >> (K - d)) == 0) t = red2p(t);
                    else t = mont_mul(t, wp[q >> (K - d)]);
                    v[q] = a + t;
                    v[q + half] = a - t + P;
                }
            }
        }
    } else {
#pragma unroll
        for (int d = K - 1; d >= 0; d--) {
            const int half = R >> (d + 1);
            const uint32_t* __restrict__ wp = W + ((1u << (s_base + u0 + d)) + (c << (u0 + d)) + (hi << d));
#pragma unroll
            for (int q = 0; q < R; q++) {
                if ((q & half) == 0) {
                    // lazy range [0, 2p) in and out: the product is left uncorrected
                    const uint32_t a = red2p(v[q]), b = red2p(v[q + half]);
                    v[q] = a + b;
                    uint32_t dlt = a - b + P;
                    if (!(TOP && (q >> (K - d)) == 0)) dlt = mont_mul_lazy(dlt, wp[q >> (K - d)]);
                    v[q + half] = dlt;
                }
            }
        }
    }
}

// TOP = true: the round starts at global stage 0 of a whole transform (s_base = u0 = c = 0, one group
// block): block 0 of every stage has the twiddle w^0 = 1, i.e. butterflies with q < 2^(K-d) at local
// stage d need no multiplication (all of stage 0, half of stage 1, ...: 47 % of a radix-16 round).
template <int K, bool INV, int LOG_DL, int NTH, bool TOP = false>
__device__ __forceinline__ void radix_round(uint32_t* s, unsigned log_total, unsigned u0,
                                            unsigned s_base, uint32_t c,
                                            const uint32_t* __restrict__ W) {
    constexpr int R = 1 << K;
    const unsigned log_dl = LOG_DL >= 0 ? (unsigned)LOG_DL : log_total - u0 - K;
    const uint32_t n_groups = 1u << (log_total - K);
    for (uint32_t g0 = threadIdx.x; g0 < n_groups; g0 += NTH) {
        uint32_t g = g0;
        if (LOG_DL == 4 && K < 5) {
            // distance 16: lanes 0-15 of a half-wave are the 16 `lo` of one block, lanes 16-31 those
            // of the block 2^(5-K) further (header comment): swap bits 4 and 9-K of the group index
            constexpr uint32_t B = 9 - K;
            const uint32_t x = ((g >> 4) ^ (g >> B)) & 1u;
            g ^= (x << 4) | (x << B);
        }
        const uint32_t lo = g & ((1u << log_dl) - 1);
        const uint32_t hi = g >> log_dl;  // block index at stage u0
        const uint32_t base = (hi << (K + log_dl)) + lo;
        const uint32_t pbase = pad(base);
        uint32_t addr[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            if (LOG_DL >= 5)  // base + (q << d) with d >= 5: the pad grows by q << (d - 5)
                addr[q] = pbase + (uint32_t)q * ((1u << (LOG_DL >= 5 ? LOG_DL : 5)) +
                                                 (1u << (LOG_DL >= 5 ? LOG_DL - 5 : 0)));
            else if (LOG_DL == 4)  // lo < 16 and the block offset is a multiple of 32
                addr[q] = pbase + 16u * (uint32_t)q + ((uint32_t)q >> 1);
            else if (LOG_DL == 0 && K == 4)  // the group's 16 words share one pad value
                addr[q] = pbase + (uint32_t)q;
            else
                addr[q] = pad(base + ((uint32_t)q << log_dl));
        }
        uint32_t v[R];
#pragma unroll
        for (int q = 0; q < R; q++) v[q] = s[addr[q]];
        radix_butterflies<K, INV, TOP>(v, s_base, u0, c, hi, W);
#pragma unroll
        for (int q = 0; q < R; q++) s[addr[q]] = v[q];
    }
    __syncthreads();
}

// Last forward round of a strided tile: groups read from the LDS, results written straight to the
// tile's positions in HBM (element e of the tile lives at o[((e >> log_T) << row_shift) + (e & mask)]),
// (values stay in [0, 2p): the contiguous forward pass follows).  A group's elements are 2^LOG_DL >= 2^log_T apart... and consecutive lanes hold
// consecutive `lo`, so every store instruction writes whole 2^log_T-word row pieces, exactly like the
// copy loop it replaces; no barrier is needed after it.
template <int K, int LOG_DL, int NTH>
__device__ __forceinline__ void radix_round_fwd_to_global(const uint32_t* s, unsigned log_total,
                                                          unsigned u0, const uint32_t* __restrict__ W,
                                                          uint32_t* __restrict__ o, unsigned log_T,
                                                          unsigned row_shift) {
    constexpr int R = 1 << K;
    const uint32_t n_groups = 1u << (log_total - K);
    const uint32_t tmask = (1u << log_T) - 1;
    for (uint32_t g = threadIdx.x; g < n_groups; g += NTH) {
        const uint32_t lo = g & ((1u << LOG_DL) - 1);
        const uint32_t hi = g >> LOG_DL;
        const uint32_t base = (hi << (K + LOG_DL)) + lo;
        uint32_t v[R];
#pragma unroll
        for (int q = 0; q < R; q++) v[q] = s[pad(base + ((uint32_t)q << LOG_DL))];
        radix_butterflies<K, false, false>(v, 0, u0, 0, hi, W);
#pragma unroll
        for (int q = 0; q < R; q++) {
            const uint32_t e = base + ((uint32_t)q << LOG_DL);
            o[((uint64_t)(e >> log_T) << row_shift) + (e & tmask)] = v[q];  // lazy: k_lde_fwd_contig reads it next
        }
    }
}

template <bool INV, int NTH>
__device__ __forceinline__ void radix_round_rt(int k, uint32_t* s, unsigned log_total, unsigned u0,
                                               unsigned s_base, uint32_t c,
                                               const uint32_t* __restrict__ W) {
    switch (k) {
        case 4: radix_round<4, INV, -1, NTH>(s, log_total, u0, s_base, c, W); break;
        case 3: radix_round<3, INV, -1, NTH>(s, log_total, u0, s_base, c, W); break;
        case 2: radix_round<2, INV, -1, NTH>(s, log_total, u0, s_base, c, W); break;
        default: radix_round<1, INV, -1, NTH>(s, log_total, u0, s_base, c, W); break;
    }
}

// Generic plan: log_len stages in ceil(log_len/4) rounds of nearly equal size (10 = 4+3+3): the
// first `rem` rounds take q+1 stages, the others q.
template <int NTH>
__device__ __forceinline__ void tile_forward_rt(uint32_t* s, unsigned log_len, unsigned log_T,
                                                unsigned s_base, uint32_t c,
                                                const uint32_t* __restrict__ W) {
    if (log_len == 0) return;
    const unsigned nr = (log_len + 3) / 4, q = log_len / nr, rem = log_len % nr;
    unsigned u = 0;
    for (unsigned r = 0; r < nr; r++) {
        const unsigned k = q + (r < rem ? 1u : 0u);
        radix_round_rt<false, NTH>((int)k, s, log_len + log_T, u, s_base, c, W);
        u += k;
    }
}
template <int NTH>
__device__ __forceinline__ void tile_inverse_rt(uint32_t* s, unsigned log_len, unsigned log_T,
                                                unsigned s_base, uint32_t c,
                                                const uint32_t* __restrict__ Winv) {
    if (log_len == 0) return;
    const unsigned nr = (log_len + 3) / 4, q = log_len / nr, rem = log_len % nr;
    unsigned u = log_len;
    for (int r = (int)nr - 1; r >= 0; r--) {
        const unsigned k = q + ((unsigned)r < rem ? 1u : 0u);
        u -= k;
        radix_round_rt<true, NTH>((int)k, s, log_len + log_T, u, s_base, c, Winv);
    }
}

// ------------------------------------------------------------------ contiguous passes (static plan)
// 2^LM-element chunk: LM = 12: 3 radix-16 rounds (last-stage distances 256, 16, 1); LM = 13: radix-32
// (distance 256), radix-16 (16), radix-16 (1); LM = 14: radix-32 (512), radix-32 (16), radix-16 (1).
template <int LM>
__device__ __forceinline__ void chunk_load(uint32_t* s, const uint32_t* __restrict__ g) {
    constexpr int NT = chunk_threads(LM);
    const uint4* g4 = reinterpret_cast<const uint4*>(g);
    uint4 v[(1 << LM) / 4 / NT];
#pragma unroll
    for (int k = 0; k < (1 << LM) / 4 / NT; k++) v[k] = g4[threadIdx.x + (uint32_t)k * NT];
#pragma unroll
    for (int k = 0; k < (1 << LM) / 4 / NT; k++) {
        const uint32_t i4 = threadIdx.x + (uint32_t)k * NT;
        const uint32_t a = 4 * i4 + (i4 >> 3);  // pad(4*i4); the 4 words stay inside one 32-group
        s[a] = v[k].x;
        s[a + 1] = v[k].y;
        s[a + 2] = v[k].z;
        s[a + 3] = v[k].w;
    }
    __syncthreads();
}
// The butterflies leave values in [0, 2p).  CANON: canonical form on the way to HBM (the LDE itself).
// Between the passes of one transform the next pass reduces its inputs anyway (red2p on `a`, a
// product with b < 2p is still < p 2^32), so intermediate images stay lazy: 2 VALU instructions per
// element less, in kernels that are bound by VALU issue.
template <int LM, bool CANON>
__device__ __forceinline__ void chunk_store(const uint32_t* s, uint32_t* __restrict__ g) {
    constexpr int NT = chunk_threads(LM);
    uint4* g4 = reinterpret_cast<uint4*>(g);
#pragma unroll
    for (int k = 0; k < (1 << LM) / 4 / NT; k++) {
        const uint32_t i4 = threadIdx.x + (uint32_t)k * NT;
        const uint32_t a = 4 * i4 + (i4 >> 3);
        if (CANON)
            g4[i4] = make_uint4(red2p(s[a]), red2p(s[a + 1]), red2p(s[a + 2]), red2p(s[a + 3]));
        else
            g4[i4] = make_uint4(s[a], s[a + 1], s[a + 2], s[a + 3]);
    }
}

// the LM local stages of a chunk, forward (u = 0 .. LM-1) or inverse (backwards)
// SKIP_R16 (inverse only): the round at distance 1 was done by k_transpose_bitrev_r16 already.
template <int LM, bool INV, bool SKIP_R16 = false>
__device__ __forceinline__ void chunk_rounds(uint32_t* s, unsigned sb, uint32_t c, const uint32_t* __restrict__ W) {
    constexpr int NT = chunk_threads(LM);
    constexpr int K0 = LM == 12 ? 4 : 5;   // stages 0 .. K0-1, distance 2^(LM - K0)
    constexpr int K1 = LM == 14 ? 5 : 4;   // stages K0 .. K0+K1-1, distance 16
    static_assert(K0 + K1 + 4 == LM, "chunk plan");
    if (!INV) {
        radix_round<K0, false, LM - K0, NT>(s, LM, 0, sb, c, W);
        radix_round<K1, false, 4, NT>(s, LM, K0, sb, c, W);
        radix_round<4, false, 0, NT>(s, LM, K0 + K1, sb, c, W);
    } else {
        if (!SKIP_R16) radix_round<4, true, 0, NT>(s, LM, K0 + K1, sb, c, W);
        radix_round<K1, true, 4, NT>(s, LM, K0, sb, c, W);
        radix_round<K0, true, LM - K0, NT>(s, LM, 0, sb, c, W);
    }
}

// chunk `c` of column blockIdx.y: global stages log_n-1 .. log_n-LM of the inverse (n > 2^LM)
template <int LM, bool SKIP_R16 = false>
__global__ void __launch_bounds__(chunk_threads(LM))
k_intt_contig(uint32_t* __restrict__ data, uint64_t col_stride, unsigned log_n,
              const uint32_t* __restrict__ Winv, uint32_t* __restrict__ data2, uint32_t gw) {
    __shared__ uint32_t s[padded(1 << LM)];
    const uint32_t c = blockIdx.x;
    // columns gw .. of a two-matrix launch live in the second matrix (coset_lde: evals2)
    uint32_t* col = blockIdx.y < gw ? data + (uint64_t)blockIdx.y * col_stride
                                    : data2 + (uint64_t)(blockIdx.y - gw) * col_stride;
    uint32_t* g = col + ((uint64_t)c << LM);
    chunk_load<LM>(s, g);
    chunk_rounds<LM, true, SKIP_R16>(s, log_n - LM, c, Winv);
    chunk_store<LM, false>(s, g);  // read next by k_lde_mid's inverse rounds
}

// The transpose of a row-major trace (src[r][c], natural rows -> dst[c][p], p = bitrev(r)) with the
// first round of the inverse transform folded in.  That round works on groups of 16 consecutive p,
// i.e. on the source rows bitrev(16 G + q): whole 256-byte row pieces whatever the group, so the
// thread that owns (group, column) loads its 16 values coalesced across the 64 columns of the tile,
// runs the four stages in registers and hands the results to the 64 x 64 LDS tile the transpose goes
// through anyway.  The transpose alone is bound by memory (its VALU idles); k_intt_contig then has
// two rounds and two LDS round trips left instead of three.
__global__ void __launch_bounds__(256)
k_transpose_bitrev_r16(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, unsigned log_n,
                       uint32_t w, uint64_t dst_col_stride, uint32_t src_width,
                       const uint32_t* __restrict__ Winv) {
    __shared__ uint32_t tile[64][65];
    const uint32_t p0 = blockIdx.x << 6;
    const uint32_t c0 = blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // bitrev(p0 + i) = bitrev(p0) + (bitrev6(i) << (log_n - 6)): p0 is a multiple of 64
    const uint32_t rb = bitrev32(p0, log_n);
    const uint64_t row_step = (uint64_t)src_width << (log_n - 6);
    uint32_t v[16];
    if (c0 + tx < w) {
        const uint32_t* sp = src + (uint64_t)rb * src_width + c0 + tx;
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = sp[(uint64_t)(__brev(16 * ty + (uint32_t)q) >> 26) * row_step];
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = 0;
    }
    // global stages log_n-1 .. log_n-4 on group (p0 >> 4) + ty (lazy [0, 2p) values out)
    radix_butterflies<4, true, false>(v, log_n - 4, 0, 0, (p0 >> 4) + ty, Winv);
#pragma unroll
    for (int q = 0; q < 16; q++) tile[16 * ty + q][tx] = v[q];
    __syncthreads();
    uint32_t t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = tile[tx][ty + 4 * k];
    uint32_t* dp = dst + (uint64_t)(c0 + ty) * dst_col_stride + p0 + tx;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (c0 + ty + 4 * k < w) dp[(uint64_t)(4 * k) * dst_col_stride] = t[k];
}

// in place on chunk `c` of coset block `beta` of column blockIdx.y: forward stages sA .. log_n-1.
// CPW > 1: a workgroup takes CPW consecutive chunks and loads chunk i+1 into registers while the
// rounds of chunk i run (the 16384-element chunk leaves room for two workgroups per CU only, and with
// 16 waves per CU nothing else covers a chunk's load latency: rocprofv3 SQ counters on 2^22 x 64,
// log_blowup 4 showed VALU busy 0.86 with 28 % of the wave-cycles parked).
template <int LM, int CPW = 1>
__global__ void __launch_bounds__(chunk_threads(LM), CPW > 1 ? 4 : 1)  // CPW > 1: keep two 512-thread groups per CU
k_lde_fwd_contig(uint32_t* __restrict__ out, uint64_t out_col_stride, unsigned log_n,
                 const uint32_t* __restrict__ W) {
    __shared__ uint32_t s[padded(1 << LM)];
    constexpr int NT = chunk_threads(LM);
    constexpr int NV = (1 << LM) / 4 / NT;
    const uint32_t beta = blockIdx.z;
    uint32_t* col = out + (uint64_t)blockIdx.y * out_col_stride + ((uint64_t)beta << log_n);
    if constexpr (CPW == 1) {
        const uint32_t c = blockIdx.x;
        uint32_t* o = col + ((uint64_t)c << LM);
        chunk_load<LM>(s, o);
        chunk_rounds<LM, false>(s, log_n - LM, c, W);
        chunk_store<LM, true>(s, o);
    } else {
        const uint32_t c0 = blockIdx.x * CPW;
        uint4 v[NV];
        {
            const uint4* g4 = reinterpret_cast<const uint4*>(col + ((uint64_t)c0 << LM));
#pragma unroll
            for (int k = 0; k < NV; k++) v[k] = g4[threadIdx.x + (uint32_t)k * NT];
        }
#pragma unroll 1
        for (int i = 0; i < CPW; i++) {
            const uint32_t c = c0 + i;
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const uint32_t i4 = threadIdx.x + (uint32_t)k * NT;
                const uint32_t a = 4 * i4 + (i4 >> 3);
                s[a] = v[k].x;
                s[a + 1] = v[k].y;
                s[a + 2] = v[k].z;
                s[a + 3] = v[k].w;
            }
            __syncthreads();
            if (i + 1 < CPW) {
                const uint4* g4 = reinterpret_cast<const uint4*>(col + ((uint64_t)(c + 1) << LM));
#pragma unroll
                for (int k = 0; k < NV; k++) v[k] = g4[threadIdx.x + (uint32_t)k * NT];
            }
            chunk_rounds<LM, false>(s, log_n - LM, c, W);
            chunk_store<LM, true>(s, col + ((uint64_t)c << LM));
            __syncthreads();  // the image is rewritten at the top of the loop
        }
    }
}

// ------------------------------------------------------------------ middle kernel
// Tile = slots {row << row_shift + j2_0 + jj : row < 2^log_len, jj < 2^log_T} of one column
// (row_shift = LOG_M for n > 4096, where rows are 4096 apart; 0 for n <= 4096 with log_T = 0, where
// the tile is the whole column).  Finishes the inverse transform (stages log_len-1 .. 0), then for
// each coset: scaled copy -> forward stages 0 .. log_len-1 -> block beta of `out`.
// PLAN 0: generic (runtime round plan).  PLAN 1: log_len = 8, log_T = 5 (n = 2^(LM + 8): 2^20 with
// 4096-element chunks, 2^21 / 2^22 with LM = 13 / 14): two radix-16 rounds with compile-time
// distances 2^9 and 2^5.  PLAN 2: log_len = 10 (n = 2^22), rounds of 4, 3
// and 3 stages with compile-time distances; TILE elements per workgroup (log_T = log2(TILE) - 10).
template <int PLAN, int TILE = TILE_ELEMS, int NTM = NT_MID, int LM = LOG_M>
__global__ void __launch_bounds__(NTM)
k_lde_mid(const uint32_t* __restrict__ evals, uint64_t in_col_stride, uint32_t* __restrict__ out,
          uint64_t out_col_stride, unsigned log_n, unsigned log_len, unsigned log_T,
          unsigned row_shift, unsigned beta0, unsigned n_cosets, const uint32_t* __restrict__ W,
          const uint32_t* __restrict__ Winv, const uint32_t* __restrict__ scale_a,
          const uint32_t* __restrict__ evals2, const uint32_t* __restrict__ scale_b, uint32_t gw) {
    __shared__ uint32_t s[padded(TILE)];
    constexpr int PER_THREAD = TILE / NTM;
    constexpr int LOG_TILE = TILE == 8192 ? 13 : (TILE == 16384 ? 14 : 15);
    if (PLAN == 1) {
        log_len = 8;
        log_T = 5;
    }
    if (PLAN == 2) {
        log_len = 10;
        log_T = LOG_TILE - 10;
    }
    // Tiles narrower than 128 B (log_T < 5) share every cache line they touch with their neighbours:
    // neighbouring tiles go to workgroups of the same XCD (ids 8 apart), whose L2 then merges the
    // pieces of a line (measured on 2^22 x 64, log_blowup 4: 19.6 -> 15.7 ms at 32 B, 14.5 -> 13.8 at 64 B)
    const bool remap = PLAN == 0 && log_T < 5 && gridDim.x >= 8;
    uint32_t bx = remap ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    uint32_t col_id = blockIdx.y;
    if constexpr (PLAN == 1) {
        // 1-D grid in the order (32 tiles, all columns, next 32 tiles, ...): the workgroups that use
        // one tile's slice of the scale table (and its twiddles) run close together and -- 32 being
        // a multiple of the 8 XCDs -- on the same XCD, so its L2 serves the slice to every column
        // instead of the Infinity Cache (measured fetch + write traffic 1.39x -> 1.11x of the
        // algorithmic bytes, same kernel time, 3.03-3.06 -> 3.00 ms/step for whole proofs)
        const uint32_t ncols = gridDim.x >> (LM - 5);  // 2^(LM - 5) tiles per column (128 at LM = 12)
        bx = (blockIdx.x & 31) + 32 * (blockIdx.x / (32 * ncols));  // < 2^(LM - 5)
        col_id = (blockIdx.x >> 5) % ncols;
    }
    if constexpr (PLAN == 2) {
        // 1-D grid, n_tiles x ncols workgroups.  The scale table of this shape (n x 2^log_blowup
        // words: 268 MB for 2^22 rows at log_blowup 4) no longer fits the Infinity Cache; with a
        // (tiles, columns) grid every column swept the whole table from HBM (rocprofv3 FETCH_SIZE:
        // 20.5 GB per proof against 1.2 GB of coefficients, profiles/r02_config4_pmc_traffic.json).
        // Order: workgroup ids go round the 8 XCDs; XCD x owns a contiguous range of tile PAIRS and,
        // for each pair, runs (tile 2p, col), (tile 2p+1, col) for all columns in turn -- the pair's
        // 2 MB of table stay in that XCD's L2 for every column, and the two 64-byte halves of every
        // output line (neighbouring tiles) are still written back to back on one XCD.
        const uint32_t n_tiles = (uint32_t)(1u << LOG_M) >> log_T;
        const uint32_t ncols = gridDim.x / n_tiles;
        const uint32_t x = blockIdx.x & 7, k = blockIdx.x >> 3;
        const uint32_t pair_local = k / (2 * ncols), r = k % (2 * ncols);
        col_id = r >> 1;
        bx = (x * (n_tiles >> 4) + pair_local) * 2 + (r & 1);
    }
    const uint32_t j2_0 = bx << log_T;
    // a two-matrix launch (coset_lde: evals2): columns gw .. come from the second matrix and take its
    // coset's scale table; the output columns follow each other either way
    const uint32_t* g = (col_id < gw ? evals + (uint64_t)col_id * in_col_stride
                                     : evals2 + (uint64_t)(col_id - gw) * in_col_stride) + j2_0;
    const uint32_t* __restrict__ scale = col_id < gw ? scale_a : scale_b;
    const uint32_t total = 1u << (log_len + log_T);
    const uint32_t tmask = (1u << log_T) - 1;
    if constexpr (PLAN == 1 || PLAN == 2) {
        // fixed shape: PER_THREAD loads issued back to back (the generic loop below is a run-time
        // loop whose iterations the compiler keeps in order: load, wait, LDS store)
        uint32_t t[PER_THREAD];
#pragma unroll
        for (int k = 0; k < PER_THREAD; k++) {
            const uint32_t i = threadIdx.x + (uint32_t)k * NTM;
            t[k] = g[((uint64_t)(i >> log_T) << row_shift) + (i & tmask)];
        }
#pragma unroll
        for (int k = 0; k < PER_THREAD; k++) s[pad(threadIdx.x + (uint32_t)k * NTM)] = t[k];
    } else {
        for (uint32_t i = threadIdx.x; i < total; i += NTM)
            s[pad(i)] = g[((uint64_t)(i >> log_T) << row_shift) + (i & tmask)];
    }
    __syncthreads();
    if (PLAN == 1) {
        radix_round<4, true, 5, NTM>(s, 13, 4, 0, 0, Winv);
        // the last inverse round (distance 2^9) works on elements tid + 512 q: exactly the share of
        // the tile this thread keeps as coefficients, so it runs in registers (below)
    } else if (PLAN == 2) {
        radix_round<3, true, LOG_TILE - 10, NTM>(s, LOG_TILE, 7, 0, 0, Winv);
        radix_round<3, true, LOG_TILE - 7, NTM>(s, LOG_TILE, 4, 0, 0, Winv);
        // last inverse round: in registers, as in PLAN 1 (below)
    } else {
        tile_inverse_rt<NTM>(s, log_len, log_T, 0, 0, Winv);
    }
    // natural-order coefficients (times n): keep this thread's share in registers
    uint32_t coef[PER_THREAD];
#pragma unroll
    for (int k = 0; k < PER_THREAD; k++) {
        const uint32_t i = threadIdx.x + (uint32_t)k * NTM;
        coef[k] = i < total ? s[pad(i)] : 0u;
    }
    // PLAN 1 / 2: the radix-16 round at the largest distance (TILE/16) has TILE/16 groups, GP =
    // PER_THREAD/16 per thread, and group j of a thread is {tid + (j + GP q) NTM : q < 16}: its own
    // coefficients coef[j + GP q].  That round therefore runs in registers on either side.
    constexpr int GP = PER_THREAD / 16;
    if constexpr (PLAN == 1 || PLAN == 2) {
#pragma unroll
        for (int j = 0; j < GP; j++) {
            uint32_t v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = coef[j + GP * q];
            radix_butterflies<4, true, true>(v, 0, 0, 0, 0, Winv);
#pragma unroll
            for (int q = 0; q < 16; q++) coef[j + GP * q] = v[q];
        }
    }
    // cosets beta0 .. beta0 + n_cosets - 1 go to blocks 0 .. n_cosets - 1 of `out` (a rank of a
    // sharded prover owns a contiguous range of cosets)
    // PLAN 1: the 16 scale-table entries of the NEXT coset are loaded while this coset is computed
    // (the table of a big shape -- 268 MB at 2^22 rows, log_blowup 4 -- is served from L2 at best, and
    // with one LDS round per coset there is little else to hide that latency behind): 16 more VGPRs
    // (109 -> 125, still two workgroups per CU).
    // (nontemporal stores of the coset blocks, to keep the table slices in L2, changed nothing at
    // 2^22 x 64, log_blowup 4: 7.92 against 7.90 ms per launch.  Nor did two LDS images used in turn,
    // which make the barrier at the top of the coset loop unnecessary (3 -> 2 barriers per coset, but
    // 114 VGPRs and run-time image addresses): 0.61 against 0.59-0.60 ms at C3, 8.19 against 7.9-8.1.)
    constexpr bool PREFETCH = PLAN == 1;
    uint32_t scv[PREFETCH ? 16 : 1];
    const uint32_t* scp = nullptr;  // this thread's first entry in coset block 0 of the table
    if constexpr (PREFETCH) {
        const uint32_t i0 = threadIdx.x;
        scp = scale + ((uint64_t)(i0 >> 5) << LM) + (i0 & 31) + j2_0;
#pragma unroll
        for (int q = 0; q < 16; q++)
            scv[q] = (scp + ((uint64_t)beta0 << log_n))[(uint64_t)q * (NTM >> 5) << LM];
    }
    for (uint32_t bl = 0; bl < n_cosets; bl++) {
        const uint32_t beta = beta0 + bl;
        const uint32_t* sc = scale + ((uint64_t)beta << log_n);  // s_beta^k / n, one entry per coefficient
        __syncthreads();
        if constexpr (PLAN == 1 || PLAN == 2) {
            // scaled coefficients and the first forward round (the thread's own elements) stay in
            // registers; LDS is written once, for the following rounds
            constexpr unsigned LT = PLAN == 1 ? 5 : LOG_TILE - 10;
#pragma unroll
            for (int j = 0; j < GP; j++) {
                uint32_t v[16];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const uint32_t i = threadIdx.x + (uint32_t)(j + GP * q) * NTM;
                    const uint32_t kk = ((i >> LT) << LM) + (i & ((1u << LT) - 1)) + j2_0;
                    // (lazy product: the butterflies that follow take [0, 2p) -- two instructions fewer
                    // per coefficient and coset)
                    if constexpr (PREFETCH) v[q] = mont_mul_lazy(coef[j + GP * q], scv[q]);
                    else v[q] = mont_mul_lazy(coef[j + GP * q], sc[kk]);
                }
                if constexpr (PREFETCH) {
                    if (bl + 1 < n_cosets) {
                        const uint32_t* nx = scp + ((uint64_t)(beta + 1) << log_n);
#pragma unroll
                        for (int q = 0; q < 16; q++) scv[q] = nx[(uint64_t)q * (NTM >> 5) << LM];
                    }
                }
                radix_butterflies<4, false, true>(v, 0, 0, 0, 0, W);
#pragma unroll
                for (int q = 0; q < 16; q++) s[pad(threadIdx.x + (uint32_t)(j + GP * q) * NTM)] = v[q];
            }
            __syncthreads();
            uint32_t* og = out + (uint64_t)col_id * out_col_stride + ((uint64_t)bl << log_n) + j2_0;
            if constexpr (PLAN == 1) {
                // (writing this round's results straight to HBM, as PLAN 2 does below: round 2 it needed
                // 137 VGPRs -- one workgroup per CU -- and ran 0.85 ms against 0.64; since the scale
                // prefetch simplified the addressing it fits in 123 and runs 0.584-0.591 ms per proof
                // against 0.588-0.601: within the noise of whole proofs, so the LDS round stays)
                radix_round<4, false, 5, NTM>(s, 13, 4, 0, 0, W);
                // lazy values: k_lde_fwd_contig reads them next.  Unrolled: 16 LDS reads in flight, then
                // 16 stores whose addresses differ by constants (as a run-time loop each iteration
                // cost 7 VALU instructions and waited for its own LDS read)
                uint32_t t[PER_THREAD];
#pragma unroll
                for (int k = 0; k < PER_THREAD; k++) t[k] = s[pad(threadIdx.x + (uint32_t)k * NTM)];
                uint32_t* ot = og + ((uint64_t)(threadIdx.x >> 5) << LM) + (threadIdx.x & 31);
#pragma unroll
                for (int k = 0; k < PER_THREAD; k++) ot[(uint64_t)k * (NTM >> 5) << LM] = t[k];
            } else {
                radix_round<3, false, LOG_TILE - 7, NTM>(s, LOG_TILE, 4, 0, 0, W);
                radix_round_fwd_to_global<3, LOG_TILE - 10, NTM>(s, LOG_TILE, 7, W, og, LOG_TILE - 10, LOG_M);
            }
            continue;  // stored; the barrier at the top of the loop protects the LDS image
        } else {
#pragma unroll
            for (int k = 0; k < PER_THREAD; k++) {
                const uint32_t i = threadIdx.x + (uint32_t)k * NTM;
                if (i < total) {
                    // coefficient index of this slot
                    const uint32_t kk = ((i >> log_T) << row_shift) + (i & tmask) + j2_0;
                    s[pad(i)] = mont_mul_lazy(coef[k], sc[kk]);
                }
            }
            __syncthreads();
            tile_forward_rt<NTM>(s, log_len, log_T, 0, 0, W);
        }
        uint32_t* o = out + (uint64_t)col_id * out_col_stride + ((uint64_t)bl << log_n) + j2_0;
        if (row_shift != 0) {  // n > 4096: k_lde_fwd_contig follows and takes lazy values
            for (uint32_t i = threadIdx.x; i < total; i += NTM)
                o[((uint64_t)(i >> log_T) << row_shift) + (i & tmask)] = s[pad(i)];
        } else {  // the whole transform was done here: canonical output
            for (uint32_t i = threadIdx.x; i < total; i += NTM)
                o[((uint64_t)(i >> log_T) << row_shift) + (i & tmask)] = red2p(s[pad(i)]);
        }
    }
}

// ------------------------------------------------------------------ host driver
// Tuning knobs of the LDE plans (A/B measurements; the defaults are what the tables in DESIGN.md were
// measured with).  Parsed ONCE per process, strictly: a value outside a knob's list is reported on
// stderr and ignored, so a typo cannot silently select another plan.
//   TS_LDE_LM=12                 n = 2^21 / 2^22 with the 4096-element chunk plans of round 2
//   TS_LDE_NO_FUSED_TRANSPOSE=1  transpose without the fused first inverse round
//   TS_LDE_TILE=0|8192|16384|32768, TS_LDE_THREADS=512|1024   PLAN 2 tile / workgroup (with TS_LDE_LM=12)
//   TS_LDE_FWD_CPW=1|2|4         chunks per workgroup of the 16384-element forward pass
struct LdeKnobs {
    int lm = 0;                   // 0 = default plan choice
    bool no_fused_transpose = false;
    int plan2_tile = 16384, plan2_threads = 1024, fwd_cpw = 4;
};
static int knob_int(const char* name, int dflt, std::initializer_list<int> allowed) {
    const char* e = getenv(name);
    if (!e) return dflt;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    bool ok = end != e && *end == 0;
    bool listed = false;
    for (int a : allowed) listed = listed || a == v;
    if (!ok || !listed) {
        fprintf(stderr, "tapstark: %s=%s ignored (allowed:", name, e);
        for (int a : allowed) fprintf(stderr, " %d", a);
        fprintf(stderr, "); using %d\n", dflt);
        return dflt;
    }
    return (int)v;
}
static const LdeKnobs& lde_knobs() {
    static const LdeKnobs k = [] {
        LdeKnobs x;
        x.lm = knob_int("TS_LDE_LM", 0, {12});
        x.no_fused_transpose = knob_int("TS_LDE_NO_FUSED_TRANSPOSE", 0, {0, 1}) != 0;
        x.plan2_tile = knob_int("TS_LDE_TILE", 16384, {0, 8192, 16384, 32768});
        x.plan2_threads = knob_int("TS_LDE_THREADS", 1024, {512, 1024});
        x.fwd_cpw = knob_int("TS_LDE_FWD_CPW", 4, {1, 2, 4});
        return x;
    }();
    return k;
}

// chunk size of the contiguous passes: 2^12, or log_n - 8 for n = 2^21 / 2^22 (header comment).
// The one place that decides it: the fused transpose and coset_lde both ask here.
static unsigned lde_chunk_log(unsigned log_n) {
    return ((log_n == 21 || log_n == 22) && lde_knobs().lm != 12) ? log_n - 8 : (unsigned)LOG_M;
}

bool launch_transpose_bitrev_r16(Context& ctx, const uint32_t* src, uint32_t* dst, unsigned log_n, uint32_t w,
                                 uint64_t dst_col_stride, uint32_t src_width) {
    if (lde_knobs().no_fused_transpose || w == 0 || log_n <= lde_chunk_log(log_n)) return false;  // no contiguous inverse pass to shorten
    ctx.ensure_twiddles(log_n);
    TS_LAUNCH(ctx, k_transpose_bitrev_r16, dim3(1u << (log_n - 6), (w + 63) / 64), dim3(256), 0, src, dst, log_n, w,
              dst_col_stride, src_width ? src_width : w, (const uint32_t*)ctx.d_twiddle_inv);
    TS_HIP(hipGetLastError());
    return true;
}

void coset_lde(Context& ctx, uint32_t* evals, uint64_t in_col_stride, uint32_t ncols, unsigned log_n,
               unsigned log_blowup, uint32_t shift, uint32_t* out, uint64_t out_col_stride,
               uint32_t beta0, uint32_t n_beta, bool first_round_done, uint32_t* evals2, uint32_t shift2,
               uint32_t gw) {
    if (ncols == 0) return;
    TS_REQUIRE(evals2 == nullptr || (gw >= 1 && gw < ncols && shift2 != 0), TS_ERR_INVALID,
               "coset_lde: second matrix needs its first column and its shift");
    if (evals2 == nullptr) gw = 0xffffffffu;
    TS_REQUIRE(log_n + log_blowup <= 27, TS_ERR_INVALID, "coset_lde: log_n + log_blowup > 27");
    TS_REQUIRE(ncols >= 1 && ncols <= 65535, TS_ERR_INVALID, "coset_lde: bad column count");
    const unsigned LM = lde_chunk_log(log_n);
    TS_REQUIRE(!first_round_done || log_n > LM, TS_ERR_INVARIANT, "coset_lde: no contiguous inverse pass at this size");
    const bool two_pass = log_n > LM;
    const unsigned sA = two_pass ? log_n - LM : 0;  // stages done by the strided (middle) kernel
    // sA <= 13 fits the 8192-element tile; sA = 14 (n = 2^26, the longest trace a blowup of 2 leaves
    // room for below the two-adicity 27) takes a 16384-element tile (68 KB of LDS)
    TS_REQUIRE(sA <= 14, TS_ERR_INVALID, "coset_lde: log_n > 26");
    ctx.ensure_twiddles(log_n == 0 ? 1 : log_n);
    const uint32_t* W = ctx.d_twiddle_fwd;
    const uint32_t* Winv = ctx.d_twiddle_inv;
    unsigned log_T = 0;
    // PLAN 2 tile (TS_LDE_TILE).  Measured on 2^22 x 64, log_blowup 4 (ms per launch): generic 19.1,
    // 8192 15.7, 16384 13.8, 32768 30.3 (one workgroup per CU); with the register-resident outer rounds
    // and the last round stored straight to HBM 16384 went 13.8 -> 11.7, and to 11.1 with 1024 threads
    // (TS_LDE_THREADS: 114 VGPRs, 16 waves per CU; 512 threads: 191 VGPRs, 8 waves)
    const int plan2_tile = lde_knobs().plan2_tile, plan2_threads = lde_knobs().plan2_threads;
    const bool plan2 = two_pass && sA == 10 && plan2_tile != 0;
    if (two_pass && sA <= 13) {
        while ((1u << (sA + log_T + 1)) <= (unsigned)TILE_ELEMS && log_T < 6) log_T++;
        if (plan2) log_T = plan2_tile == 8192 ? 3 : (plan2_tile == 16384 ? 4 : 5);
    }

    // per-coset scale table s_beta^k / n (cached per context: the trace's is the same every proof)
    const uint32_t n_cosets = 1u << log_blowup;
    if (n_beta == 0) n_beta = n_cosets - beta0;
    TS_REQUIRE(beta0 < n_cosets && n_beta <= n_cosets - beta0, TS_ERR_INVALID, "coset_lde: coset range");
    const uint32_t* scale = coset_scale_table(ctx, log_n, log_blowup, shift);
    const uint32_t* scale2 = evals2 ? coset_scale_table(ctx, log_n, log_blowup, shift2) : nullptr;

    if (two_pass) {
        // the vectorised chunk loads need 16-byte aligned columns
        TS_REQUIRE(in_col_stride % 4 == 0 && out_col_stride % 4 == 0, TS_ERR_INVALID,
                   "coset_lde: column strides must be multiples of 4 elements");
        // (ctx.lde_pass_mask: measurement only -- ts_bench_stage runs one of the three passes alone, on
        // whatever the buffers hold, to sample its clock and power; every product path leaves it at 7)
        if (ctx.lde_pass_mask & 1u) {
            // (stage names: the sharded prover reports where a rank's time goes)
            StageTimer t(&ctx, "lde: inverse NTT, contiguous stages");
            const dim3 g(1u << sA, ncols);
#define TS_INTT(LMV)                                                                                          \
    do {                                                                                                      \
        if (first_round_done)                                                                                 \
            TS_LAUNCH(ctx, (k_intt_contig<LMV, true>), g, dim3(chunk_threads(LMV)), 0, evals, in_col_stride, log_n, \
                      Winv, evals2, gw);                                                                      \
        else                                                                                                  \
            TS_LAUNCH(ctx, k_intt_contig<LMV>, g, dim3(chunk_threads(LMV)), 0, evals, in_col_stride, log_n, Winv, \
                      evals2, gw);                                                                            \
    } while (0)
            if (LM == 12) TS_INTT(12);
            else if (LM == 13) TS_INTT(13);
            else TS_INTT(14);
#undef TS_INTT
        }
        StageTimer t_rest(&ctx, "lde: strided pass + forward NTT of the owned cosets");
        const dim3 grid(1u << (LM - log_T), ncols);
        const dim3 grid1((1u << (LM - log_T)) * ncols);  // PLAN 1 / 2: 1-D, see the kernel
#define TS_MID_ARGS                                                                            \
    (const uint32_t*)evals, in_col_stride, out, out_col_stride, log_n, sA, log_T, (unsigned)LM, \
        beta0, n_beta, W, Winv, scale, (const uint32_t*)evals2, scale2, gw
        if (!(ctx.lde_pass_mask & 2u)) {
        } else if (sA == 14)
            TS_LAUNCH(ctx, (k_lde_mid<0, 16384>), grid, dim3(NT_MID), 0, TS_MID_ARGS);
        else if (plan2 && plan2_tile == 8192)
            TS_LAUNCH(ctx, (k_lde_mid<2, 8192>), grid1, dim3(NT_MID), 0, TS_MID_ARGS);
        else if (plan2 && plan2_tile == 16384 && plan2_threads == 1024)
            TS_LAUNCH(ctx, (k_lde_mid<2, 16384, 1024>), grid1, dim3(1024), 0, TS_MID_ARGS);
        else if (plan2 && plan2_tile == 16384)
            TS_LAUNCH(ctx, (k_lde_mid<2, 16384>), grid1, dim3(NT_MID), 0, TS_MID_ARGS);
        else if (plan2)
            TS_LAUNCH(ctx, (k_lde_mid<2, 32768, 1024>), grid1, dim3(1024), 0, TS_MID_ARGS);
        else if (sA == 8 && log_T == 5 && LM == 12)
            TS_LAUNCH(ctx, k_lde_mid<1>, grid1, dim3(NT_MID), 0, TS_MID_ARGS);
        else if (sA == 8 && log_T == 5 && LM == 13)
            TS_LAUNCH(ctx, (k_lde_mid<1, 8192, 512, 13>), grid1, dim3(NT_MID), 0, TS_MID_ARGS);
        else if (sA == 8 && log_T == 5 && LM == 14)
            TS_LAUNCH(ctx, (k_lde_mid<1, 8192, 512, 14>), grid1, dim3(NT_MID), 0, TS_MID_ARGS);
        else
            TS_LAUNCH(ctx, k_lde_mid<0>, grid, dim3(NT_MID), 0, TS_MID_ARGS);
        const dim3 gf(1u << sA, ncols, n_beta);
        // chunks per workgroup of the 16384-element forward pass: measured 12.49 (1) / 13.34 (2: spills) /
        // 12.22 ms (4) per proof
        const int fwd_cpw = lde_knobs().fwd_cpw;
        if (!(ctx.lde_pass_mask & 4u)) {
        } else if (LM == 12)
            TS_LAUNCH(ctx, k_lde_fwd_contig<12>, gf, dim3(chunk_threads(12)), 0, out, out_col_stride, log_n, W);
        else if (LM == 13)
            TS_LAUNCH(ctx, k_lde_fwd_contig<13>, gf, dim3(chunk_threads(13)), 0, out, out_col_stride, log_n, W);
        else if (fwd_cpw == 1)
            TS_LAUNCH(ctx, k_lde_fwd_contig<14>, gf, dim3(chunk_threads(14)), 0, out, out_col_stride, log_n, W);
        else if (fwd_cpw == 2)
            TS_LAUNCH(ctx, (k_lde_fwd_contig<14, 2>), dim3(gf.x / 2, gf.y, gf.z), dim3(chunk_threads(14)), 0, out,
                      out_col_stride, log_n, W);
        else
            TS_LAUNCH(ctx, (k_lde_fwd_contig<14, 4>), dim3(gf.x / 4, gf.y, gf.z), dim3(chunk_threads(14)), 0, out,
                      out_col_stride, log_n, W);
    } else {
        TS_LAUNCH(ctx, k_lde_mid<0>, dim3(1, ncols), dim3(NT_MID), 0, (const uint32_t*)evals,
                  in_col_stride, out, out_col_stride, log_n, log_n, 0u, 0u, beta0, n_beta, W, Winv, scale,
                  (const uint32_t*)evals2, scale2, gw);
    }
    TS_HIP(hipGetLastError());
}

}  // namespace ts
