// tapstark_air.hpp -- header-only C++ capture of an AIR into the constraint tape ts_air_compile takes.
//
// The compiled-language counterpart of tap-stark_amd/air.py, mirroring how the reference turns an
// `Air::eval` body into symbolic constraints:
//   SymbolicVariable / Entry      uni-stark/src/symbolic_variable.rs:9-38
//   SymbolicExpression            uni-stark/src/symbolic_expression.rs:12-61 (degree rules :41-61)
//   SymbolicAirBuilder            uni-stark/src/symbolic_builder.rs:68-148
//   get_symbolic_constraints      uni-stark/src/symbolic_builder.rs:52-64
//   FilteredAirBuilder            p3-air (when_first_row / when_transition / when_last_row)
// Write `eval(ts::air::Builder&)` the way the reference writes `eval(&self, builder: &mut AB)`, then
// hand `builder.tape()` to ts_air_compile.  No dependency beyond the standard library.
#pragma once
#include <stdint.h>

#include <map>
#include <stdexcept>
#include <tuple>
#include <vector>

namespace ts {
namespace air {

constexpr uint32_t P = 0x78000001u;  // basic/src/field/mod.rs:45
constexpr uint32_t TAPE_MAGIC = 0x54415354u;
enum Op : uint32_t { CONST = 0, MAIN = 1, PUBLIC = 2, IS_FIRST = 3, IS_LAST = 4, IS_TRANSITION = 5,
                     ADD = 6, SUB = 7, NEG = 8, MUL = 9 };

class Builder;

// a node of the constraint DAG (hash-consed per builder: shared sub-expressions are emitted once)
class Expr {
public:
    Expr() = default;
    Expr(Builder* b, uint32_t id) : b_(b), id_(id) {}
    uint32_t id() const { return id_; }
    Builder* builder() const { return b_; }

private:
    Builder* b_ = nullptr;
    uint32_t id_ = 0;
};

class Filtered;

class Builder {
public:
    Builder(uint32_t width, uint32_t num_public_values) : width_(width), n_public_(num_public_values) {
        for (uint32_t off = 0; off < 2; off++)
            for (uint32_t c = 0; c < width; c++) rows_[off].push_back(node(MAIN, off, c, 1));
        for (uint32_t i = 0; i < num_public_values; i++) public_.push_back(node(PUBLIC, i, 0, 0));
    }
    // builder.main().row_slice(0 | 1)
    const std::vector<Expr>& local() const { return rows_[0]; }
    const std::vector<Expr>& next() const { return rows_[1]; }
    const std::vector<Expr>& public_values() const { return public_; }
    Expr constant(uint64_t v) { return node(CONST, (uint32_t)(v % P), 0, 0); }
    Expr is_first_row() { return node(IS_FIRST, 0, 0, 1); }    // symbolic_expression.rs:45
    Expr is_last_row() { return node(IS_LAST, 0, 0, 1); }      // :46
    Expr is_transition() { return node(IS_TRANSITION, 0, 0, 0); }  // :47 (window size 2)
    void assert_zero(Expr x) { constraints_.push_back(x.id()); }  // symbolic_builder.rs:136-138
    void assert_eq(Expr x, Expr y);
    Filtered when(Expr c);
    Filtered when_first_row();
    Filtered when_last_row();
    Filtered when_transition();

    // symbolic_builder.rs:15-50
    uint32_t max_constraint_degree() const {
        uint32_t d = 0;
        for (uint32_t c : constraints_) d = degs_[c] > d ? degs_[c] : d;
        return d;
    }
    uint32_t log_quotient_degree() const {
        uint32_t d = max_constraint_degree();
        if (d < 2) d = 2;
        uint32_t k = 0;
        while ((1u << k) < d - 1) k++;  // log2_ceil(d - 1)
        return k;
    }
    // [magic, version, width, n_public, n_nodes, n_constraints, nodes (op, a, b)..., constraint ids...]
    std::vector<uint32_t> tape() const {
        std::vector<uint32_t> t = {TAPE_MAGIC, 1, width_, n_public_, (uint32_t)nodes_.size(),
                                   (uint32_t)constraints_.size()};
        for (auto& n : nodes_) {
            t.push_back(std::get<0>(n));
            t.push_back(std::get<1>(n));
            t.push_back(std::get<2>(n));
        }
        t.insert(t.end(), constraints_.begin(), constraints_.end());
        return t;
    }

    Expr node(uint32_t op, uint32_t a, uint32_t b, uint32_t deg) {
        const auto key = std::make_tuple(op, a, b);
        auto it = cse_.find(key);
        if (it != cse_.end()) return Expr(this, it->second);
        const uint32_t id = (uint32_t)nodes_.size();
        nodes_.push_back(key);
        degs_.push_back(deg);
        cse_[key] = id;
        return Expr(this, id);
    }
    uint32_t degree(Expr e) const { return degs_[e.id()]; }

private:
    uint32_t width_, n_public_;
    std::vector<std::tuple<uint32_t, uint32_t, uint32_t>> nodes_;
    std::vector<uint32_t> degs_;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> cse_;
    std::vector<uint32_t> constraints_;
    std::vector<Expr> rows_[2], public_;
};

// degree rules: symbolic_expression.rs:137 (add), :182 (sub), :227 (mul)
inline Expr operator+(Expr x, Expr y) {
    Builder* b = x.builder();
    const uint32_t dx = b->degree(x), dy = b->degree(y);
    return b->node(ADD, x.id(), y.id(), dx > dy ? dx : dy);
}
inline Expr operator-(Expr x, Expr y) {
    Builder* b = x.builder();
    const uint32_t dx = b->degree(x), dy = b->degree(y);
    return b->node(SUB, x.id(), y.id(), dx > dy ? dx : dy);
}
inline Expr operator-(Expr x) { return x.builder()->node(NEG, x.id(), 0, x.builder()->degree(x)); }
inline Expr operator*(Expr x, Expr y) {
    Builder* b = x.builder();
    return b->node(MUL, x.id(), y.id(), b->degree(x) + b->degree(y));
}
inline Expr operator+(Expr x, uint64_t c) { return x + x.builder()->constant(c); }
inline Expr operator-(Expr x, uint64_t c) { return x - x.builder()->constant(c); }
inline Expr operator*(Expr x, uint64_t c) { return x * x.builder()->constant(c); }

// p3-air FilteredAirBuilder: when(c).assert_zero(x) => assert_zero(c * x)
class Filtered {
public:
    Filtered(Builder* b, Expr cond) : b_(b), cond_(cond) {}
    void assert_zero(Expr x) { b_->assert_zero(cond_ * x); }
    void assert_eq(Expr x, Expr y) { assert_zero(x - y); }
    void assert_one(Expr x) { assert_zero(x - 1); }
    Filtered when(Expr c) { return Filtered(b_, cond_ * c); }

private:
    Builder* b_;
    Expr cond_;
};
inline void Builder::assert_eq(Expr x, Expr y) { assert_zero(x - y); }
inline Filtered Builder::when(Expr c) { return Filtered(this, c); }
inline Filtered Builder::when_first_row() { return when(is_first_row()); }
inline Filtered Builder::when_last_row() { return when(is_last_row()); }
inline Filtered Builder::when_transition() { return when(is_transition()); }

}  // namespace air
}  // namespace ts
