/*
 * oracle/air.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * The AIR enters as a serialised symbolic-constraint DAG ("tape", format in oracle.h), the
 * restatement of what the reference obtains by running Air::eval on a SymbolicAirBuilder:
 *   uni-stark/src/symbolic_builder.rs:52-64   get_symbolic_constraints
 *   uni-stark/src/symbolic_builder.rs:15-32   get_log_quotient_degree
 *   uni-stark/src/symbolic_expression.rs:41-61,137,182,227  degree_multiple rules
 *   uni-stark/src/symbolic_variable.rs:33-38  variable degree (main 1, public 0)
 *   uni-stark/src/check_constraints.rs:11-39  row-by-row constraint check
 */
#include "oracle_internal.h"
#include <stdlib.h>

int ts_or_tape_parse(const uint32_t* tape, size_t n_words, ts_or_tape* t) {
    if (n_words < 6 || tape[0] != TS_TAPE_MAGIC || tape[1] != 1) return -1;
    t->width = tape[2];
    t->n_public = tape[3];
    t->n_nodes = tape[4];
    t->n_constraints = tape[5];
    if ((size_t)6 + 3 * (size_t)t->n_nodes + t->n_constraints != n_words) return -1;
    t->nodes = tape + 6;
    t->constraints = tape + 6 + 3 * (size_t)t->n_nodes;
    for (uint32_t i = 0; i < t->n_nodes; i++) {
        uint32_t op = t->nodes[3 * i], a = t->nodes[3 * i + 1], b = t->nodes[3 * i + 2];
        switch (op) {
            case TS_OP_CONST:
                if (a >= BB_P) return -1;
                break;
            case TS_OP_MAIN:
                if (a > 1 || b >= t->width) return -1;
                break;
            case TS_OP_PUBLIC:
                if (a >= t->n_public) return -1;
                break;
            case TS_OP_IS_FIRST:
            case TS_OP_IS_LAST:
            case TS_OP_IS_TRANSITION:
                break;
            case TS_OP_ADD:
            case TS_OP_SUB:
            case TS_OP_MUL:
                if (a >= i || b >= i) return -1;
                break;
            case TS_OP_NEG:
                if (a >= i) return -1;
                break;
            default:
                return -1;
        }
    }
    for (uint32_t i = 0; i < t->n_constraints; i++)
        if (t->constraints[i] >= t->n_nodes) return -1;
    return 0;
}

int ts_or_tape_validate(const uint32_t* tape, size_t n_words) {
    ts_or_tape t;
    return ts_or_tape_parse(tape, n_words, &t);
}

/* symbolic_expression.rs:41-61 + the max/sum rules at :137 (add), :182 (sub), :227 (mul) */
int ts_or_air_max_constraint_degree(const uint32_t* tape, size_t n_words) {
    ts_or_tape t;
    if (ts_or_tape_parse(tape, n_words, &t)) return -1;
    uint32_t* deg = (uint32_t*)malloc((t.n_nodes + 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < t.n_nodes; i++) {
        uint32_t op = t.nodes[3 * i], a = t.nodes[3 * i + 1], b = t.nodes[3 * i + 2];
        switch (op) {
            case TS_OP_MAIN: deg[i] = 1; break;
            case TS_OP_IS_FIRST: deg[i] = 1; break;
            case TS_OP_IS_LAST: deg[i] = 1; break;
            case TS_OP_IS_TRANSITION: deg[i] = 0; break;
            case TS_OP_CONST: deg[i] = 0; break;
            case TS_OP_PUBLIC: deg[i] = 0; break;
            case TS_OP_ADD:
            case TS_OP_SUB: deg[i] = deg[a] > deg[b] ? deg[a] : deg[b]; break;
            case TS_OP_NEG: deg[i] = deg[a]; break;
            case TS_OP_MUL: deg[i] = deg[a] + deg[b]; break;
            default: deg[i] = 0;
        }
    }
    uint32_t mx = 0;
    for (uint32_t i = 0; i < t.n_constraints; i++)
        if (deg[t.constraints[i]] > mx) mx = deg[t.constraints[i]];
    free(deg);
    return (int)mx;
}

/* symbolic_builder.rs:24-31: max(degree, 2), then log2_ceil(degree - 1) */
int ts_or_air_log_quotient_degree(const uint32_t* tape, size_t n_words) {
    int d = ts_or_air_max_constraint_degree(tape, n_words);
    if (d < 0) return -1;
    if (d < 2) d = 2;
    int k = 0;
    while ((1 << k) < d - 1) k++;
    return k;
}

void ts_or_tape_eval_base(const ts_or_tape* t, const uint32_t* local, const uint32_t* next,
                          const uint32_t* pis, uint32_t is_first, uint32_t is_last,
                          uint32_t is_transition, uint32_t* v) {
    for (uint32_t i = 0; i < t->n_nodes; i++) {
        uint32_t op = t->nodes[3 * i], a = t->nodes[3 * i + 1], b = t->nodes[3 * i + 2];
        switch (op) {
            case TS_OP_CONST: v[i] = a; break;
            case TS_OP_MAIN: v[i] = a ? next[b] : local[b]; break;
            case TS_OP_PUBLIC: v[i] = pis[a]; break;
            case TS_OP_IS_FIRST: v[i] = is_first; break;
            case TS_OP_IS_LAST: v[i] = is_last; break;
            case TS_OP_IS_TRANSITION: v[i] = is_transition; break;
            case TS_OP_ADD: v[i] = bb_add(v[a], v[b]); break;
            case TS_OP_SUB: v[i] = bb_sub(v[a], v[b]); break;
            case TS_OP_NEG: v[i] = bb_neg(v[a]); break;
            case TS_OP_MUL: v[i] = bb_mul(v[a], v[b]); break;
        }
    }
}

void ts_or_tape_eval_ext(const ts_or_tape* t, const ef4* local, const ef4* next,
                         const uint32_t* pis, ef4 is_first, ef4 is_last, ef4 is_transition,
                         ef4* v) {
    for (uint32_t i = 0; i < t->n_nodes; i++) {
        uint32_t op = t->nodes[3 * i], a = t->nodes[3 * i + 1], b = t->nodes[3 * i + 2];
        switch (op) {
            case TS_OP_CONST: v[i] = ef4_from_base(a); break;
            case TS_OP_MAIN: v[i] = a ? next[b] : local[b]; break;
            case TS_OP_PUBLIC: v[i] = ef4_from_base(pis[a]); break;
            case TS_OP_IS_FIRST: v[i] = is_first; break;
            case TS_OP_IS_LAST: v[i] = is_last; break;
            case TS_OP_IS_TRANSITION: v[i] = is_transition; break;
            case TS_OP_ADD: v[i] = ef4_add(v[a], v[b]); break;
            case TS_OP_SUB: v[i] = ef4_sub(v[a], v[b]); break;
            case TS_OP_NEG: v[i] = ef4_neg(v[a]); break;
            case TS_OP_MUL: v[i] = ef4_mul(v[a], v[b]); break;
        }
    }
}

/* The value of every constraint on `m` independent (local, next, selectors) inputs, by direct
 * evaluation of the DAG: what folder.rs:60-64 `assert_zero` receives, before any folding with alpha.
 * sels = m x {is_first, is_last, is_transition}; out = m x n_constraints.  Used to check the
 * product's lowered register program (tests interpret it) constraint by constraint. */
int ts_or_tape_constraint_values(const uint32_t* tape, size_t n_words, const uint32_t* local,
                                 const uint32_t* next, size_t m, const uint32_t* pis,
                                 const uint32_t* sels, uint32_t* out) {
    ts_or_tape t;
    if (ts_or_tape_parse(tape, n_words, &t)) return -1;
    uint32_t* v = (uint32_t*)malloc((t.n_nodes + 1) * sizeof(uint32_t));
    for (size_t i = 0; i < m; i++) {
        ts_or_tape_eval_base(&t, local + i * t.width, next + i * t.width, pis, sels[3 * i],
                             sels[3 * i + 1], sels[3 * i + 2], v);
        for (uint32_t c = 0; c < t.n_constraints; c++)
            out[i * t.n_constraints + c] = v[t.constraints[c]];
    }
    free(v);
    return 0;
}

/* check_constraints.rs:18-38: is_first = (i==0), is_last = (i==h-1), is_transition = (i!=h-1),
 * next row wraps around */
int64_t ts_or_check_constraints(const uint32_t* tape, size_t n_words, const uint32_t* trace,
                                size_t n, const uint32_t* pis) {
    ts_or_tape t;
    if (ts_or_tape_parse(tape, n_words, &t)) return -2;
    uint32_t* v = (uint32_t*)malloc((t.n_nodes + 1) * sizeof(uint32_t));
    int64_t bad = -1;
    for (size_t i = 0; i < n && bad < 0; i++) {
        size_t inext = (i + 1) % n;
        ts_or_tape_eval_base(&t, trace + i * t.width, trace + inext * t.width, pis, i == 0,
                             i == n - 1, i != n - 1, v);
        for (uint32_t c = 0; c < t.n_constraints; c++)
            if (v[t.constraints[c]] != 0) {
                bad = (int64_t)i * 65536 + c;
                break;
            }
    }
    free(v);
    return bad;
}
