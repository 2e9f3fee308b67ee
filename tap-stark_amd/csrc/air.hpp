// Constraint tape -> device program.
//
// The AIR crosses the C ABI as the serialised symbolic-constraint DAG that the reference obtains
// from `get_symbolic_constraints` (uni-stark/src/symbolic_builder.rs:52-64; node kinds of
// symbolic_expression.rs:12-37).  `compile_air` validates it, applies the reference's degree rules
// (symbolic_expression.rs:41-61,137,182,227; symbolic_builder.rs:15-32) and lowers it to a linear
// register program that the quotient kernel interprets (quotient.hip), one thread per row.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace ts {

// tape format (include/tapstark.h)
constexpr uint32_t TAPE_MAGIC = 0x54415354u;
enum TapeOp : uint32_t {
    T_CONST = 0, T_MAIN = 1, T_PUBLIC = 2, T_IS_FIRST = 3, T_IS_LAST = 4, T_IS_TRANSITION = 5,
    T_ADD = 6, T_SUB = 7, T_NEG = 8, T_MUL = 9
};

// device instruction: 4 x u32 {op, dst, a, b}.  Operands of ADD/SUB/MUL/NEG/ASSERT are register ids.
enum DevOp : uint32_t {
    D_LOAD = 0,     // dst <- to_mont(main[a = offset][b = column])
    D_CONST = 1,    // dst <- consts[a]           (Montgomery; constants and public values)
    D_SEL = 2,      // dst <- selector a (0 first, 1 last, 2 transition)
    D_ADD = 3,
    D_SUB = 4,
    D_NEG = 5,
    D_MUL = 6,
    D_ASSERT = 7,   // acc += reg[a] * alpha_pow[b]   (b = constraint index)
};

struct AirProgram {
    uint32_t width = 0;
    uint32_t n_public = 0;
    uint32_t n_constraints = 0;
    uint32_t max_degree = 0;
    uint32_t log_quotient_degree = 0;
    uint32_t n_regs = 0;
    std::vector<uint32_t> code;             // 4 words per instruction
    std::vector<uint32_t> const_canonical;  // constants (canonical); publics are appended per proof
    std::vector<uint32_t> const_public_idx; // for entries that are public values: index, else ~0u
    std::vector<uint32_t> tape;             // the validated input (kept for the verifier side)
    // device copy of `code`, owned by the context that compiled it
    uint32_t* d_code = nullptr;
    // hiprtc-specialised quotient kernel (jit.cpp); null => the interpreter in quotient.hip is used
    void* jit_module = nullptr;
    void* jit_fn = nullptr;
};

// throws ts::Error(TS_ERR_INVALID) on a malformed tape
AirProgram compile_air(const uint32_t* tape, size_t n_words);

}  // namespace ts
