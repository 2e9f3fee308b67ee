// Leaves AND tree of a Merkle commitment in one launch (merkle_tree.hpp: leaf_tree_body): the kernel
// template and its launcher, shared by merkle.hip (committed matrices) and fri.hip (commit-phase
// rounds, with the fold in the leaf).  Device code; include from .hip files only.
#pragma once
#include <stdlib.h>

#include "kernels.hpp"
#include "merkle_tree.hpp"

namespace ts {

// (Occupancy is not what holds this kernel: built for 5 and 6 waves per SIMD -- 96 / 80 VGPRs with 80 /
// 192 bytes of scratch -- it measured the same 199 / 482 us on 2^22 rows of 2 / 64 elements as at the 4
// waves its 116 VGPRs allow.  Its issue fill is 0.95; what is left is the clock.)
template <int LOG_R, class Leaf>
__global__ void __launch_bounds__(mt::NTH, 4)  // four waves per SIMD: 128 VGPRs (it needs 127-130 as the inlining falls)
k_leaf_tree(Leaf leaf, uint32_t* __restrict__ tree, unsigned log_leaves, int finish,
            uint32_t* __restrict__ ticket, DevChallenger* __restrict__ ch, uint32_t* __restrict__ root_out,
            Ef* __restrict__ beta_out) {
    __shared__ mt::T8::Lds lds;
    __shared__ uint32_t s_last;
    const mt::Levels lv{tree, 0, (uint64_t)1 << log_leaves};
    mt::leaf_tree_body<LOG_R>(lds, s_last, leaf, lv, log_leaves, finish != 0, ticket, ch, root_out, beta_out);
}

// merkle.hip: the whole-tree kernel on the levels from `first_level` up (first_level's nodes are in
// the tree); at most 2^MAX_LOG_TREE of them
void launch_merkle_tree_from(Context& ctx, uint32_t* tree, unsigned log_leaves, unsigned first_level,
                             DevChallenger* ch, uint32_t* root_out, Ef* beta_out);

// Leaf digests and every level of a tree of 2^log_leaves >= 2^LEAF_TREE_MIN_LOG leaves.  One launch
// while the workgroups leave at most 2^LEAF_TREE_MAX_LOG_SUB sub-roots (trees up to 2^16 .. 2^18
// leaves, by leaves per lane); above, the leaf launch stops at its sub-roots and the whole-tree kernel
// finishes.  With `ch`, the workgroup that makes the root observes it and samples (as
// launch_merkle_levels).
template <class Leaf>
void launch_leaf_tree(Context& ctx, const Leaf& leaf, uint32_t* tree, unsigned log_leaves, DevChallenger* ch,
                      uint32_t* root_out, Ef* beta_out) {
    TS_REQUIRE(log_leaves >= mt::LEAF_TREE_MIN_LOG && log_leaves <= 27, TS_ERR_INVALID,
               "leaf_tree: between 2^8 and 2^27 leaves");
    // measurement knobs: TS_LEAF_TREE_R = leaves per lane (log2, 0..3) whatever the height;
    // TS_LEAF_TREE_FINISH=0: the sub-roots always go to a second launch (the whole-tree kernel)
    static const int knob_r = [] { const char* e = getenv("TS_LEAF_TREE_R"); return e ? atoi(e) : -1; }();
    static const int knob_finish = [] { const char* e = getenv("TS_LEAF_TREE_FINISH"); return e ? atoi(e) : 1; }();
    unsigned log_r = mt::leaf_tree_log_r(log_leaves);
    if (knob_r >= 0 && knob_r <= 3 && log_leaves >= 8u + (unsigned)knob_r) log_r = (unsigned)knob_r;
    const unsigned log_b = 8 + log_r;
    const unsigned log_sub = log_leaves - log_b;
    const int finish = (log_sub <= mt::LEAF_TREE_MAX_LOG_SUB && (knob_finish || log_sub == 0)) ? 1 : 0;
    const dim3 grid(1u << log_sub), block(mt::NTH);
    DevChallenger* kch = finish ? ch : nullptr;
    // (kernel timers: one name per leaf kind and R, Leaf::name(log_r))
#define TS_LEAF_TREE_CASE(LR)                                                                              \
    case LR: {                                                                                             \
        ts::KernelTimer _kt(&ctx, Leaf::name(LR));                                                         \
        hipLaunchKernelGGL((k_leaf_tree<LR, Leaf>), grid, block, 0, ctx.stream, leaf, tree, log_leaves,    \
                           finish, ctx.ticket(), kch, root_out, beta_out);                                 \
    } break;
    switch (log_r) {
        TS_LEAF_TREE_CASE(0)
        TS_LEAF_TREE_CASE(1)
        TS_LEAF_TREE_CASE(2)
        TS_LEAF_TREE_CASE(3)
    }
#undef TS_LEAF_TREE_CASE
    TS_HIP(hipGetLastError());
    if (!finish) launch_merkle_tree_from(ctx, tree, log_leaves, log_b, ch, root_out, beta_out);
}

}  // namespace ts
