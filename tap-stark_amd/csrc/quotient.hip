// Quotient evaluation (reference uni-stark/src/prover.rs:122-194 `quotient_values` with
// ProverConstraintFolder, folder.rs:11-19,60-64; selectors: p3-commit selectors_on_coset,
// SURVEY.md App. A.4; chunking: prover.rs:78-80 flatten_to_base/split_evals).
//
// One thread per row of the quotient domain 31*H_{n*qd}, addressed in the LDE's own storage order
// (bit-reversed), so that `local` is a coalesced read and `next` (natural index + qd) is a second
// coalesced read for all but one wavefront in 2^(L-6).  The constraint program is interpreted with
// wave-uniform control flow; its register file lives in LDS ([reg][thread], conflict-free) or, for
// programs with more live registers than LDS holds, in a global slab per workgroup.
// folder.rs:60-64 accumulates acc = acc*alpha + c_i; here the same value is formed as
// sum_i c_i * alpha^(K-1-i) with precomputed powers (4 base multiplications per constraint instead
// of an EF4 x EF4 product) -- exact field arithmetic, identical result.
#include <stdlib.h>

#include <algorithm>

#include "air.hpp"
#include "kernels.hpp"

namespace ts {

// ------------------------------------------------------------------ selectors
// Storage index r <-> natural index i = bitrev_L(r); x_r = shift * omega_{2^L}^i (shift = 31, the
// quotient domain of prover.rs:65-66; a rank of the sharded prover that evaluates the quotient on its
// own cosets passes their shift, sharded.cpp "local quotient").
// W is the block-twiddle table: omega_{2^L}^bitrev_L(r) = (r odd ? -1 : 1) * W[2^(L-1) + (r >> 1)].
constexpr int SEL_BATCH = 8;

struct SelConsts {
    uint32_t zh_mont[MAX_QUOTIENT_CHUNKS];  // Z_H on the qd cosets: 31^n * omega_qd^c - 1  (Montgomery)
};

__global__ void __launch_bounds__(256)
k_selectors(unsigned L, unsigned log_qd, const uint32_t* __restrict__ W, uint32_t gen_mont,
            uint32_t gn_inv_mont, SelConsts sc, uint32_t* __restrict__ is_first,
            uint32_t* __restrict__ is_last, uint32_t* __restrict__ is_transition) {
    const uint64_t total = 1ull << L;
    const uint64_t r0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * SEL_BATCH;
    if (r0 >= total) return;
    const unsigned log_n = L - log_qd;
    uint32_t d[2 * SEL_BATCH];
    uint32_t pre[2 * SEL_BATCH];
    uint32_t run = R_MOD_P;
    const int cnt = (total - r0) < (uint64_t)SEL_BATCH ? (int)(total - r0) : SEL_BATCH;
#pragma unroll
    for (int k = 0; k < SEL_BATCH; k++) {
        if (k < cnt) {
            uint64_t r = r0 + k;
            uint32_t w = L == 0 ? R_MOD_P : W[(total >> 1) + (r >> 1)];
            if (r & 1) w = neg(w);
            uint32_t x = mont_mul(gen_mont, w);
            d[2 * k] = sub(x, R_MOD_P);         // x - 1
            d[2 * k + 1] = sub(x, gn_inv_mont);  // x - omega_n^-1
        } else {
            d[2 * k] = d[2 * k + 1] = R_MOD_P;
        }
        pre[2 * k] = run;
        run = mont_mul(run, d[2 * k]);
        pre[2 * k + 1] = run;
        run = mont_mul(run, d[2 * k + 1]);
    }
    uint32_t inv = mont_inv(run);
#pragma unroll
    for (int k = SEL_BATCH - 1; k >= 0; k--) {
        uint32_t inv_last = mont_mul(inv, pre[2 * k + 1]);
        inv = mont_mul(inv, d[2 * k + 1]);
        uint32_t inv_first = mont_mul(inv, pre[2 * k]);
        inv = mont_mul(inv, d[2 * k]);
        if (k < cnt) {
            uint64_t r = r0 + k;
            // coset index c = natural i mod qd = bitrev_lqd(r >> log_n)
            uint32_t c = bitrev32((uint32_t)(r >> log_n), log_qd);
            uint32_t zh = sc.zh_mont[c];
            is_first[r] = mont_mul(zh, inv_first);
            is_last[r] = mont_mul(zh, inv_last);
            is_transition[r] = d[2 * k + 1];
        }
    }
}

void launch_selectors(Context& ctx, unsigned log_n, unsigned log_qd, uint32_t* is_first,
                      uint32_t* is_last, uint32_t* is_transition, uint32_t shift) {
    const unsigned L = log_n + log_qd;
    TS_REQUIRE((1u << log_qd) <= (unsigned)MAX_QUOTIENT_CHUNKS, TS_ERR_UNSUPPORTED, "quotient degree > 64 not supported");
    ctx.ensure_twiddles(L == 0 ? 1 : L);
    SelConsts sc;
    const uint32_t s_pow_n = pow_canon(shift, 1ull << log_n);
    const uint32_t gqd = two_adic_generator(log_qd);
    for (uint32_t c = 0; c < (1u << log_qd); c++)
        sc.zh_mont[c] = to_mont(sub(mul(s_pow_n, pow_canon(gqd, c)), 1));
    const uint32_t gn_inv = inv_canon(two_adic_generator(log_n));
    const uint64_t threads = (((uint64_t)1 << L) + SEL_BATCH - 1) / SEL_BATCH;
    TS_LAUNCH(ctx, k_selectors, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, L, log_qd, ctx.d_twiddle_fwd, to_mont(shift), to_mont(gn_inv), sc, is_first,
                       is_last, is_transition);
    TS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ interpreter
struct QuotConsts {
    uint32_t inv_zh_canonical[MAX_QUOTIENT_CHUNKS];  // 1/Z_H per coset, CANONICAL: acc(Mont) * it -> canonical
};

// Register file of the interpreter: [reg][thread], in LDS while it fits (conflict-free, the common
// case), otherwise in a global scratch slab of the workgroup ([workgroup][reg][thread]: coalesced, L2
// resident for moderate programs) with a persistent grid walking the row tiles, so that the slab is
// sized by the grid and not by the domain.  There is no cap on live registers: a program the JIT
// declines (jit.cpp: instruction budget) still runs on the device.
struct RegFilePlan {
    int nthreads;
    bool global;
    size_t lds_bytes;
    unsigned grid;
    size_t scratch_words;
};

static RegFilePlan plan_reg_file(Context& ctx, uint32_t n_regs, uint64_t rows) {
    RegFilePlan pl;
    pl.nthreads = 256;
    while (pl.nthreads > 64 && (size_t)n_regs * pl.nthreads * 4 > 48 * 1024) pl.nthreads >>= 1;
    pl.lds_bytes = (size_t)n_regs * pl.nthreads * 4;
    // LDS only while a 256-lane workgroup's file stays small (several workgroups per CU); beyond that the
    // global slab is FASTER, not just possible: 424 registers in LDS leave one wave per CU (16.6 ms on a
    // 2^18-row domain against 3.9 from the slab, profiles/r06_quotient_paths.txt).  Knobs for measurements.
    const size_t lds_max_regs = [] { const char* e = getenv("TS_INTERP_LDS_MAX_REGS"); return e ? (size_t)atoi(e) : (size_t)48; }();
    const unsigned waves_per_cu = [] { const char* e = getenv("TS_INTERP_WAVES_PER_CU"); return e ? (unsigned)atoi(e) : 16u; }();
    pl.global = n_regs > lds_max_regs || pl.lds_bytes > ctx.max_lds_per_block || getenv("TS_INTERP_GLOBAL_REGS") != nullptr;
    const uint64_t tiles = (rows + pl.nthreads - 1) / pl.nthreads;
    if (pl.global) {
        pl.nthreads = 64;
        pl.lds_bytes = 0;
        const uint64_t t64 = (rows + 63) / 64;
        // a few waves per SIMD hide the slab's latency (16 per CU: 1.5x over 8, 32 adds nothing); the slab stays
        // below 4 GiB of the 288 (a smaller, cache-resident slab is slower: the grid is what matters)
        uint64_t grid = std::min<uint64_t>(t64, (uint64_t)ctx.num_cus * waves_per_cu);
        const uint64_t slab_mb = [] { const char* e = getenv("TS_INTERP_SLAB_MB"); return e ? (uint64_t)atoi(e) : (uint64_t)4096; }();
        const uint64_t cap = (slab_mb << 18) / ((uint64_t)n_regs * 64);
        grid = std::max<uint64_t>(1, std::min(grid, cap));
        pl.grid = (unsigned)grid;
        pl.scratch_words = (size_t)grid * n_regs * 64;
    } else {
        pl.grid = (unsigned)tiles;
        pl.scratch_words = 0;
    }
    return pl;
}

template <int NTHREADS, bool GLOBAL_REGS>
__global__ void __launch_bounds__(NTHREADS)
k_quotient(const uint32_t* __restrict__ code, uint32_t n_instr, uint32_t n_regs,
           const uint32_t* __restrict__ lde, uint64_t col_stride, unsigned log_n, unsigned log_qd,
           const uint32_t* __restrict__ consts_mont, const uint32_t* __restrict__ alpha_pows,
           const uint32_t* __restrict__ is_first, const uint32_t* __restrict__ is_last,
           const uint32_t* __restrict__ is_transition, QuotConsts qc, QuotOut out,
           uint32_t row_begin, uint32_t row_end, uint32_t* __restrict__ reg_slabs, uint32_t n_tiles) {
    extern __shared__ uint32_t lds_regs[];  // [n_regs][NTHREADS] unless GLOBAL_REGS
    const unsigned L = log_n + log_qd;
    const uint32_t total = 1u << L;
    uint32_t* my = GLOBAL_REGS ? reg_slabs + (size_t)blockIdx.x * n_regs * NTHREADS + threadIdx.x
                               : lds_regs + threadIdx.x;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t r = row_begin + tile * NTHREADS + threadIdx.x;
        const bool active = r < row_end;
        const uint32_t rr = active ? r : row_begin;
        const uint32_t i = bitrev32(rr, L);
        const uint32_t i_next = (i + (1u << log_qd)) & (total - 1);  // prover.rs:139-140,165
        const uint32_t r_next = bitrev32(i_next, L);
        const uint32_t sel0 = is_first[rr], sel1 = is_last[rr], sel2 = is_transition[rr];
        const uint32_t* row_local = lde + rr;
        const uint32_t* row_next = lde + r_next;
        uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;

        // Wave-uniform instruction fetch (scalar loads), one instruction ahead.  The value an instruction
        // produces stays in a VGPR for the next one (`fwd`): in a post-order evaluation the next instruction
        // nearly always consumes it, and reading it back from the register file would put a memory round
        // trip (LDS or, worse, the slab) on every link of the dependency chain.
        const uint4* code4 = reinterpret_cast<const uint4*>(code);
        uint4 ins = code4[0];
        uint32_t fwd_reg = ~0u, fwd_val = 0;
        auto rd = [&](uint32_t reg) { return reg == fwd_reg ? fwd_val : my[(size_t)reg * NTHREADS]; };
        for (uint32_t pc = 0; pc < n_instr; pc++) {
            const uint4 nxt = code4[pc + 1 < n_instr ? pc + 1 : pc];
            const uint32_t op = ins.x, dst = ins.y, a = ins.z, b = ins.w;
            ins = nxt;
            uint32_t v;
            switch (op) {
                case D_LOAD: {
                    const uint32_t* base = a ? row_next : row_local;
                    v = to_mont(base[(uint64_t)b * col_stride]);
                    break;
                }
                case D_CONST: v = consts_mont[a]; break;
                case D_SEL: v = a == 0 ? sel0 : (a == 1 ? sel1 : sel2); break;
                case D_ADD: v = add(rd(a), rd(b)); break;
                case D_SUB: v = sub(rd(a), rd(b)); break;
                case D_NEG: v = neg(rd(a)); break;
                case D_MUL: v = mont_mul(rd(a), rd(b)); break;
                default: {  // D_ASSERT
                    const uint32_t c = rd(a);
                    const uint32_t* ap = alpha_pows + 4 * b;
                    acc0 = add(acc0, mont_mul(c, ap[0]));
                    acc1 = add(acc1, mont_mul(c, ap[1]));
                    acc2 = add(acc2, mont_mul(c, ap[2]));
                    acc3 = add(acc3, mont_mul(c, ap[3]));
                    continue;
                }
            }
            my[(size_t)dst * NTHREADS] = v;
            fwd_reg = dst;
            fwd_val = v;
        }
        if (!active) continue;
        // quotient(x) = constraints(x) / Z_H(x)  (prover.rs:183); flatten + split (prover.rs:78-80):
        // natural row i -> chunk i % qd, position i / qd; stored bit-reversed = r & (n-1)
        const uint32_t c = bitrev32(r >> log_n, log_qd);
        const uint32_t iz = qc.inv_zh_canonical[c];
        const uint64_t n = 1ull << log_n;
        uint32_t* o = out.chunk[c] + (r & (n - 1));
        o[0] = mont_mul(acc0, iz);
        o[n] = mont_mul(acc1, iz);
        o[2 * n] = mont_mul(acc2, iz);
        o[3 * n] = mont_mul(acc3, iz);
    }
}

template <class K>
static void allow_lds(K kernel, size_t lds) {
    // above 64 KiB the dynamic LDS size has to be granted per function (gfx950: 160 KiB per workgroup)
    if (lds > 48 * 1024)
        TS_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
}

void launch_quotient(Context& ctx, const AirProgram& air, const ColMat& trace_lde, unsigned log_n,
                     unsigned log_qd, const uint32_t* d_consts_mont, const uint32_t* d_alpha_pows_mont,
                     const uint32_t* is_first, const uint32_t* is_last, const uint32_t* is_transition,
                     const QuotOut& out, uint64_t row_begin, uint64_t row_end, uint32_t shift) {
    TS_REQUIRE(air.d_code != nullptr, TS_ERR_INVALID, "air program not uploaded");
    TS_REQUIRE(log_n + log_qd <= 31, TS_ERR_INVALID, "quotient domain too large");
    QuotConsts qc;
    const uint32_t s_pow_n = pow_canon(shift, 1ull << log_n);
    const uint32_t gqd = two_adic_generator(log_qd);
    for (uint32_t c = 0; c < (1u << log_qd); c++)
        qc.inv_zh_canonical[c] = inv_canon(sub(mul(s_pow_n, pow_canon(gqd, c)), 1));
    const uint32_t n_instr = (uint32_t)(air.code.size() / 4);
    if (row_end == 0) row_end = 1ull << (log_n + log_qd);
    TS_REQUIRE(row_begin < row_end && row_end <= (1ull << (log_n + log_qd)), TS_ERR_INVALID,
               "quotient: row range");
    const uint64_t total = row_end - row_begin;  // rows to do
    uint32_t rb = (uint32_t)row_begin, re = (uint32_t)row_end;
    if (air.jit_fn) {
        // specialised straight-line kernel (jit.cpp); same arguments, same results
        const uint32_t* lde_p = trace_lde.d;
        uint64_t stride = trace_lde.col_stride;
        QuotOut qo = out;
        void* args[] = {&lde_p, &stride, &log_n, &log_qd, &d_consts_mont, &d_alpha_pows_mont,
                        &is_first, &is_last, &is_transition, &qc, &qo, &rb, &re};
        KernelTimer kt(&ctx, "k_quotient_jit");
        TS_HIP(hipModuleLaunchKernel((hipFunction_t)air.jit_fn, (unsigned)((total + 255) / 256), 1, 1,
                                     256, 1, 1, 0, ctx.stream, args, nullptr));
        return;
    }
    const RegFilePlan pl = plan_reg_file(ctx, air.n_regs, total);
    DevBuf<uint32_t> slabs;
    if (pl.global) slabs = DevBuf<uint32_t>(&ctx, pl.scratch_words);
    const uint32_t n_tiles = (uint32_t)((total + pl.nthreads - 1) / pl.nthreads);
#define TS_LAUNCH_Q(NTH, GLOB)                                                                          \
    do {                                                                                                \
        allow_lds(k_quotient<NTH, GLOB>, pl.lds_bytes);                                                 \
        TS_LAUNCH(ctx, (k_quotient<NTH, GLOB>), dim3(pl.grid), dim3(NTH), pl.lds_bytes, air.d_code, n_instr, \
                  air.n_regs, trace_lde.d, trace_lde.col_stride, log_n, log_qd, d_consts_mont,          \
                  d_alpha_pows_mont, is_first, is_last, is_transition, qc, out, rb, re, slabs.p, n_tiles); \
    } while (0)
    if (pl.global) TS_LAUNCH_Q(64, true);
    else if (pl.nthreads == 256) TS_LAUNCH_Q(256, false);
    else if (pl.nthreads == 128) TS_LAUNCH_Q(128, false);
    else TS_LAUNCH_Q(64, false);
#undef TS_LAUNCH_Q
    TS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ chunk mix (sharded local quotient)
// One thread per (row, base column j < 4) of the slab: v[c'] = chunk c' column j at this row, then
// chunk c <- sum_c' mix[c][c'] v[c'].  Canonical values x Montgomery constants -> canonical.
template <int QD>
__global__ void __launch_bounds__(256)
k_chunk_mix(uint32_t* const* __restrict__ chunks, uint64_t rows, uint64_t col_stride,
            const uint32_t* __restrict__ mix) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 4 * rows) return;
    const uint64_t row = t % rows, j = t / rows;
    uint32_t v[QD];
#pragma unroll
    for (int c = 0; c < QD; c++) v[c] = chunks[c][j * col_stride + row];
#pragma unroll
    for (int c = 0; c < QD; c++) {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < QD; k++) acc = add(acc, mont_mul(v[k], mix[c * QD + k]));
        chunks[c][j * col_stride + row] = acc;
    }
}

void launch_chunk_mix(Context& ctx, uint32_t* const* d_chunk_ptrs, uint32_t qd, uint64_t rows, uint64_t col_stride,
                      const uint32_t* d_mix_mont) {
    const dim3 grid((unsigned)((4 * rows + 255) / 256));
#define TS_MIX(Q) TS_LAUNCH(ctx, k_chunk_mix<Q>, grid, dim3(256), 0, d_chunk_ptrs, rows, col_stride, d_mix_mont)
    switch (qd) {
        case 2: TS_MIX(2); break;
        case 4: TS_MIX(4); break;
        case 8: TS_MIX(8); break;
        case 16: TS_MIX(16); break;
        case 32: TS_MIX(32); break;
        case 64: TS_MIX(64); break;
        default: TS_REQUIRE(false, TS_ERR_INVALID, "chunk mix: quotient degree must be 2 .. 64");
    }
#undef TS_MIX
    TS_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ check_constraints
// reference uni-stark/src/check_constraints.rs:11-39 (debug builds of prove(), prover.rs:40-41):
// every constraint must vanish on every row of the trace itself, with is_first_row = (i == 0),
// is_last_row = (i == h-1), is_transition = (i != h-1) and the next row wrapping around.
// One thread per row of the row-major trace; the first violation (row * 2^16 + constraint index,
// smallest wins) is left in *violation.
template <int NTHREADS, bool GLOBAL_REGS>
__global__ void __launch_bounds__(NTHREADS)
k_check_constraints(const uint32_t* __restrict__ code, uint32_t n_instr, uint32_t n_regs,
                    const uint32_t* __restrict__ trace, uint32_t width, uint64_t n,
                    const uint32_t* __restrict__ consts_mont, unsigned long long* __restrict__ violation,
                    uint32_t* __restrict__ reg_slabs, uint32_t n_tiles) {
    extern __shared__ uint32_t lds_regs[];
    uint32_t* my = GLOBAL_REGS ? reg_slabs + (size_t)blockIdx.x * n_regs * NTHREADS + threadIdx.x
                               : lds_regs + threadIdx.x;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t i = (uint64_t)tile * NTHREADS + threadIdx.x;
        const bool active = i < n;
        const uint64_t ii = active ? i : 0;
        const uint32_t* row_local = trace + ii * width;
        const uint32_t* row_next = trace + ((ii + 1) % n) * width;
        const uint32_t sel0 = ii == 0 ? R_MOD_P : 0u;
        const uint32_t sel1 = ii == n - 1 ? R_MOD_P : 0u;
        const uint32_t sel2 = ii != n - 1 ? R_MOD_P : 0u;
        unsigned long long bad = ~0ull;
        const uint4* code4 = reinterpret_cast<const uint4*>(code);
        uint4 ins = code4[0];
        uint32_t fwd_reg = ~0u, fwd_val = 0;  // as in k_quotient
        auto rd = [&](uint32_t reg) { return reg == fwd_reg ? fwd_val : my[(size_t)reg * NTHREADS]; };
        for (uint32_t pc = 0; pc < n_instr; pc++) {
            const uint4 nxt = code4[pc + 1 < n_instr ? pc + 1 : pc];
            const uint32_t op = ins.x, dst = ins.y, a = ins.z, b = ins.w;
            ins = nxt;
            uint32_t v;
            switch (op) {
                case D_LOAD: v = to_mont((a ? row_next : row_local)[b]); break;
                case D_CONST: v = consts_mont[a]; break;
                case D_SEL: v = a == 0 ? sel0 : (a == 1 ? sel1 : sel2); break;
                case D_ADD: v = add(rd(a), rd(b)); break;
                case D_SUB: v = sub(rd(a), rd(b)); break;
                case D_NEG: v = neg(rd(a)); break;
                case D_MUL: v = mont_mul(rd(a), rd(b)); break;
                default: {  // D_ASSERT
                    if (rd(a) != 0 && bad == ~0ull) bad = ii * 65536ull + b;
                    continue;
                }
            }
            my[(size_t)dst * NTHREADS] = v;
            fwd_reg = dst;
            fwd_val = v;
        }
        if (active && bad != ~0ull) atomicMin(violation, bad);
    }
}

void launch_check_constraints(Context& ctx, const AirProgram& air, const uint32_t* trace_row_major,
                              uint64_t n, const uint32_t* d_consts_mont,
                              unsigned long long* d_violation) {
    TS_REQUIRE(air.d_code != nullptr, TS_ERR_INVALID, "air program not uploaded");
    // the report is row * 2^16 + constraint index (the oracle's and stark.py's format)
    TS_REQUIRE(air.n_constraints <= 65536, TS_ERR_UNSUPPORTED, "check_constraints: more than 65536 constraints");
    const uint32_t n_instr = (uint32_t)(air.code.size() / 4);
    const RegFilePlan pl = plan_reg_file(ctx, air.n_regs, n);
    DevBuf<uint32_t> slabs;
    if (pl.global) slabs = DevBuf<uint32_t>(&ctx, pl.scratch_words);
    const uint32_t n_tiles = (uint32_t)((n + pl.nthreads - 1) / pl.nthreads);
#define TS_LAUNCH_C(NTH, GLOB)                                                                         \
    do {                                                                                               \
        allow_lds(k_check_constraints<NTH, GLOB>, pl.lds_bytes);                                       \
        TS_LAUNCH(ctx, (k_check_constraints<NTH, GLOB>), dim3(pl.grid), dim3(NTH), pl.lds_bytes,       \
                  air.d_code, n_instr, air.n_regs, trace_row_major, air.width, n, d_consts_mont,       \
                  d_violation, slabs.p, n_tiles);                                                      \
    } while (0)
    if (pl.global) TS_LAUNCH_C(64, true);
    else if (pl.nthreads == 256) TS_LAUNCH_C(256, false);
    else if (pl.nthreads == 128) TS_LAUNCH_C(128, false);
    else TS_LAUNCH_C(64, false);
#undef TS_LAUNCH_C
    TS_HIP(hipGetLastError());
}

}  // namespace ts
