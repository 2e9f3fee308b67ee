#include "air.hpp"

#include <algorithm>
#include <map>
#include <string>

#include "bb.hpp"
#include "context.hpp"

namespace ts {

AirProgram compile_air(const uint32_t* tape, size_t n_words) {
    TS_REQUIRE(tape && n_words >= 6 && tape[0] == TAPE_MAGIC && tape[1] == 1, TS_ERR_INVALID,
               "air tape: bad header");
    AirProgram p;
    p.width = tape[2];
    p.n_public = tape[3];
    const uint32_t n_nodes = tape[4];
    p.n_constraints = tape[5];
    TS_REQUIRE((size_t)6 + 3 * (size_t)n_nodes + p.n_constraints == n_words, TS_ERR_INVALID,
               "air tape: length does not match header");
    TS_REQUIRE(p.width >= 1, TS_ERR_INVALID, "air tape: zero width");
    const uint32_t* nodes = tape + 6;
    const uint32_t* cons = tape + 6 + 3 * (size_t)n_nodes;

    // validation + degree_multiple (symbolic_expression.rs:41-61)
    std::vector<uint32_t> deg(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) {
        uint32_t op = nodes[3 * i], a = nodes[3 * i + 1], b = nodes[3 * i + 2];
        switch (op) {
            case T_CONST:
                TS_REQUIRE(a < P, TS_ERR_INVALID, "air tape: non-canonical constant");
                deg[i] = 0;
                break;
            case T_MAIN:
                TS_REQUIRE(a <= 1 && b < p.width, TS_ERR_INVALID, "air tape: bad main variable");
                deg[i] = 1;
                break;
            case T_PUBLIC:
                TS_REQUIRE(a < p.n_public, TS_ERR_INVALID, "air tape: bad public index");
                deg[i] = 0;
                break;
            case T_IS_FIRST:
            case T_IS_LAST:
                deg[i] = 1;
                break;
            case T_IS_TRANSITION:
                deg[i] = 0;
                break;
            case T_ADD:
            case T_SUB:
                TS_REQUIRE(a < i && b < i, TS_ERR_INVALID, "air tape: forward reference");
                deg[i] = std::max(deg[a], deg[b]);
                break;
            case T_NEG:
                TS_REQUIRE(a < i, TS_ERR_INVALID, "air tape: forward reference");
                deg[i] = deg[a];
                break;
            case T_MUL:
                TS_REQUIRE(a < i && b < i, TS_ERR_INVALID, "air tape: forward reference");
                deg[i] = deg[a] + deg[b];
                break;
            default:
                throw Error(TS_ERR_INVALID, "air tape: unknown op");
        }
    }
    uint32_t mx = 0;
    for (uint32_t c = 0; c < p.n_constraints; c++) {
        TS_REQUIRE(cons[c] < n_nodes, TS_ERR_INVALID, "air tape: bad constraint id");
        mx = std::max(mx, deg[cons[c]]);
    }
    p.max_degree = mx;
    uint32_t d = std::max(mx, 2u);  // symbolic_builder.rs:24-26
    uint32_t lq = 0;
    while ((1u << lq) < d - 1) lq++;  // log2_ceil(d - 1), :31
    p.log_quotient_degree = lq;
    p.tape.assign(tape, tape + n_words);

    // ---- lowering: liveness + linear-scan register allocation --------------------------------
    // reachable nodes
    std::vector<uint8_t> live(n_nodes, 0);
    for (uint32_t c = 0; c < p.n_constraints; c++) live[cons[c]] = 1;
    for (int64_t i = (int64_t)n_nodes - 1; i >= 0; i--) {
        if (!live[i]) continue;
        uint32_t op = nodes[3 * i], a = nodes[3 * i + 1], b = nodes[3 * i + 2];
        if (op == T_ADD || op == T_SUB || op == T_MUL) live[a] = live[b] = 1;
        if (op == T_NEG) live[a] = 1;
    }
    // Schedule: a node is emitted right before its first use (demand-driven, depth first in
    // constraint order), so leaf loads are not all hoisted to the top and register pressure stays
    // close to the expression depth.  Each constraint is asserted as soon as its root is computed.
    // use counts
    std::vector<uint32_t> uses(n_nodes, 0);
    for (uint32_t i = 0; i < n_nodes; i++) {
        if (!live[i]) continue;
        uint32_t op = nodes[3 * i], a = nodes[3 * i + 1], b = nodes[3 * i + 2];
        if (op == T_ADD || op == T_SUB || op == T_MUL) { uses[a]++; uses[b]++; }
        if (op == T_NEG) uses[a]++;
    }
    for (uint32_t c = 0; c < p.n_constraints; c++) uses[cons[c]]++;

    std::vector<int32_t> reg_of(n_nodes, -1);
    std::vector<uint32_t> free_regs;
    uint32_t next_reg = 0;
    auto alloc_reg = [&]() -> uint32_t {
        if (!free_regs.empty()) {
            uint32_t r = free_regs.back();
            free_regs.pop_back();
            return r;
        }
        return next_reg++;
    };
    auto emit = [&](uint32_t op, uint32_t dst, uint32_t a, uint32_t b) {
        p.code.push_back(op);
        p.code.push_back(dst);
        p.code.push_back(a);
        p.code.push_back(b);
    };
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> const_slot;  // (public index | ~0u, value) -> slot
    auto add_const = [&](uint32_t canonical, uint32_t public_idx) -> uint32_t {
        auto [it, fresh] = const_slot.try_emplace({public_idx, canonical}, (uint32_t)p.const_canonical.size());
        if (fresh) {
            p.const_canonical.push_back(canonical);
            p.const_public_idx.push_back(public_idx);
        }
        return it->second;
    };
    auto release = [&](uint32_t node) {
        if (--uses[node] == 0) {
            free_regs.push_back((uint32_t)reg_of[node]);
            reg_of[node] = -1;
        }
    };

    // iterative post-order evaluation
    std::vector<std::pair<uint32_t, int>> stack;
    auto eval_node = [&](uint32_t root) {
        if (reg_of[root] >= 0) return;
        stack.push_back({root, 0});
        while (!stack.empty()) {
            auto [nd, state] = stack.back();
            uint32_t op = nodes[3 * nd], a = nodes[3 * nd + 1], b = nodes[3 * nd + 2];
            bool binary = (op == T_ADD || op == T_SUB || op == T_MUL);
            if (reg_of[nd] >= 0) { stack.pop_back(); continue; }
            if (state == 0) {
                stack.back().second = 1;
                if ((binary || op == T_NEG) && reg_of[a] < 0) { stack.push_back({a, 0}); continue; }
            }
            if (state <= 1) {
                stack.back().second = 2;
                if (binary && reg_of[b] < 0) { stack.push_back({b, 0}); continue; }
            }
            // operands ready (note: evaluating b cannot have freed a, a still has this pending use)
            uint32_t ra = 0, rb = 0;
            if (binary || op == T_NEG) ra = (uint32_t)reg_of[a];
            if (binary) rb = (uint32_t)reg_of[b];
            // free operands before allocating dst so that dst may reuse an operand register
            if (binary || op == T_NEG) release(a);
            if (binary) release(b);
            uint32_t dst = alloc_reg();
            reg_of[nd] = (int32_t)dst;
            switch (op) {
                case T_CONST: emit(D_CONST, dst, add_const(a, ~0u), 0); break;
                case T_PUBLIC: emit(D_CONST, dst, add_const(0, a), 0); break;
                case T_MAIN: emit(D_LOAD, dst, a, b); break;
                case T_IS_FIRST: emit(D_SEL, dst, 0, 0); break;
                case T_IS_LAST: emit(D_SEL, dst, 1, 0); break;
                case T_IS_TRANSITION: emit(D_SEL, dst, 2, 0); break;
                case T_ADD: emit(D_ADD, dst, ra, rb); break;
                case T_SUB: emit(D_SUB, dst, ra, rb); break;
                case T_MUL: emit(D_MUL, dst, ra, rb); break;
                case T_NEG: emit(D_NEG, dst, ra, 0); break;
            }
            stack.pop_back();
        }
    };
    for (uint32_t c = 0; c < p.n_constraints; c++) {
        eval_node(cons[c]);
        emit(D_ASSERT, 0, (uint32_t)reg_of[cons[c]], c);
        release(cons[c]);
    }
    p.n_regs = std::max(next_reg, 1u);
    return p;
}

}  // namespace ts
