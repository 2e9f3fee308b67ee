// One BLAKE3 compression spread over FOUR adjacent lanes (device only).
//
// Near the top of a Merkle tree (and in the FRI tail) there are fewer nodes than lanes and the time
// is the dependency chain of one compression: ~680 VALU instructions back to back per tree level.
// The four column G functions of a round are independent, and so are the four diagonal ones, so
// four lanes can each run one of them: lane j of a quad holds column j of the 4x4 state
// (a = s[j], b = s[4+j], c = s[8+j], d = s[12+j]); the diagonal step rotates b, c, d by 1, 2, 3
// lanes inside the quad (DPP quad_perm: a modifier on the consuming instruction, no LDS) and back.
// That is 14 G per lane instead of 56: the chain is ~3.5x shorter.
//
// Message words: each lane fetches the words its G functions need from where the block already lies
// (LDS), through 28 offsets it derives once from the message schedule (7 rounds x {column x, column y,
// diagonal x, diagonal y}).  Callers on a latency chain fetch all 28 in ONE batch before the rounds and
// pass the registers (one LDS latency per compression; fetched round by round it was four) -- with a
// single wave on a SIMD every instruction costs its four cycles, on the dependent chain or not, so such
// callers also keep every address a per-lane constant (merkle_tree.hpp: reduce_levels, fri.hip: tail).
#pragma once
#include "blake3.hpp"

namespace ts {
namespace b3 {

#if defined(__HIPCC__)

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the
// wave's outstanding global stores (s_waitcnt vmcnt(0): 1-2 us each time near the top of a tree,
// where every level stores its few digests and then meets at a barrier); the digests of a level
// travel to the next one through LDS, so the stores may stay in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// message schedule, 4 bits per position: round r, positions 0..7 in LO[r], 8..15 in HI[r]
// (the rows of TS_B3_ROUND in blake3.hpp; row r+1 = row r permuted by the BLAKE3 message permutation)
__device__ __forceinline__ uint32_t sched_word(int r, uint32_t pos) {
    constexpr uint8_t S[7][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
        {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8},
        {3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1},
        {10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6},
        {12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4},
        {9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7},
        {11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13}};
    uint64_t packed = 0;
    for (int i = 0; i < 16; i++) packed |= (uint64_t)S[r][i] << (4 * i);
    return (uint32_t)(packed >> (4 * pos)) & 15u;
}

// idx[4 r + {0,1,2,3}] = message word index lane j needs in round r for (column x, column y,
// diagonal x, diagonal y)
__device__ __forceinline__ void quad_schedule(uint32_t j, uint32_t idx[28]) {
#pragma unroll
    for (int r = 0; r < 7; r++) {
        idx[4 * r + 0] = sched_word(r, 2 * j);
        idx[4 * r + 1] = sched_word(r, 2 * j + 1);
        idx[4 * r + 2] = sched_word(r, 8 + 2 * j);
        idx[4 * r + 3] = sched_word(r, 9 + 2 * j);
    }
}

template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}

// (rotr(d ^ a, 16) as xor + alignbit here, not the two-SDWA form of blake3.hpp: the SDWA pair draws two
// hazard s_nops, which cost a lone wave an issue slot each -- 28 per compression)
#define TS_B3_GQ(mx, my)       \
    a = a + b + (mx);          \
    d = rotr(d ^ a, 16);       \
    c = c + d;                 \
    b = rotr(b ^ c, 12);       \
    a = a + b + (my);          \
    d = rotr(d ^ a, 8);        \
    c = c + d;                 \
    b = rotr(b ^ c, 7);

// Column j of the initial state: (cv[j], cv[4+j], IV[j], {counter lo, counter hi, block_len, flags}[j]).
// It depends on the lane only: a kernel that compresses level after level derives it once (as a
// chain of selects inside every call it cost ~ 20 instructions and two branches).
struct QuadIv {
    uint32_t a, b, c, d;
};

// Lane j (0..3 inside its quad; all four lanes must be active) passes column j of the initial state
// and gets back word j and word 4+j of the output.  msg(k) returns the message word for schedule slot
// k (0..27, the slot order of quad_schedule).
template <class Msg>
__device__ __forceinline__ void compress_quad(const QuadIv& s0, Msg msg, uint32_t& out_lo, uint32_t& out_hi) {
    uint32_t a = s0.a, b = s0.b, c = s0.c, d = s0.d;
#pragma unroll
    for (int r = 0; r < 7; r++) {
        const uint32_t m0 = msg(4 * r + 0), m1 = msg(4 * r + 1), m2 = msg(4 * r + 2), m3 = msg(4 * r + 3);
        TS_B3_GQ(m0, m1)
        // diagonalise: this lane's diagonal is (a_j, b_{j+1}, c_{j+2}, d_{j+3}).  (Oldest value first -- a G
        // ends d, c, b: a DPP read needs two wait states after the write of its source.)
        d = quad_perm<0x93>(d);  // from lane (j+3) & 3
        c = quad_perm<0x4E>(c);  // from lane (j+2) & 3
        b = quad_perm<0x39>(b);  // from lane (j+1) & 3
        TS_B3_GQ(m2, m3)
        d = quad_perm<0x39>(d);
        c = quad_perm<0x4E>(c);
        b = quad_perm<0x93>(b);
    }
    out_lo = a ^ c;
    out_hi = b ^ d;
}

__device__ __forceinline__ uint32_t iv_word(uint32_t k) {
    constexpr uint32_t IVW[8] = {TS_B3_IV0, TS_B3_IV1, TS_B3_IV2, TS_B3_IV3,
                                 TS_B3_IV4, TS_B3_IV5, TS_B3_IV6, TS_B3_IV7};
    uint32_t v = IVW[0];
#pragma unroll
    for (int i = 1; i < 8; i++) v = k == (uint32_t)i ? IVW[i] : v;
    return v;
}

// the initial state of a one-block hash (chaining value = IV, counter = 0)
__device__ __forceinline__ QuadIv quad_iv(uint32_t j, uint32_t block_len, uint32_t flags) {
    return QuadIv{iv_word(j), iv_word(4 + j), iv_word(j), j == 2 ? block_len : j == 3 ? flags : 0u};
}

template <class Msg>
__device__ __forceinline__ void compress_quad(uint32_t j, uint32_t cv_lo, uint32_t cv_hi, Msg msg,
                                              uint32_t block_len, uint32_t flags, uint32_t& out_lo,
                                              uint32_t& out_hi) {
    compress_quad(QuadIv{cv_lo, cv_hi, iv_word(j), j == 2 ? block_len : j == 3 ? flags : 0u}, msg, out_lo, out_hi);
}

#endif

}  // namespace b3
}  // namespace ts
