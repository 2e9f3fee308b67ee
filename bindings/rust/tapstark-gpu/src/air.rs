//! AIR -> constraint tape (`ts_air_compile`, include/tapstark.h "AIR").
//!
//! `get_symbolic_constraints` (uni-stark/src/symbolic_builder.rs:52-64) runs `Air::eval` once on
//! symbolic variables; the resulting `Vec<SymbolicExpression<F>>` (symbolic_expression.rs:12-37) is
//! a DAG of `Rc` nodes.  The tape lists every distinct node once, operands before users
//! ({op, a, b} triples), then the constraint roots in `assert_zero` call order -- the order
//! `ProverConstraintFolder::assert_zero` folds them with alpha (folder.rs:60-64).
use std::collections::HashMap;
use std::rc::Rc;

use p3_air::Air;
use p3_field::{Field, PrimeField32};
// (uni-stark keeps its modules private and re-exports their items at the crate root, src/lib.rs:24-35)
use uni_stark::{get_symbolic_constraints, Entry, SymbolicAirBuilder, SymbolicExpression};

const TAPE_MAGIC: u32 = 0x5441_5354;
const OP_CONST: u32 = 0;
const OP_MAIN: u32 = 1;
const OP_PUBLIC: u32 = 2;
const OP_IS_FIRST_ROW: u32 = 3;
const OP_IS_LAST_ROW: u32 = 4;
const OP_IS_TRANSITION: u32 = 5;
const OP_ADD: u32 = 6;
const OP_SUB: u32 = 7;
const OP_NEG: u32 = 8;
const OP_MUL: u32 = 9;

struct TapeBuilder<F: Field> {
    nodes: Vec<[u32; 3]>,
    by_ptr: HashMap<*const SymbolicExpression<F>, u32>, // shared sub-expressions (Rc) are emitted once
    by_leaf: HashMap<[u32; 3], u32>,
}

impl<F: PrimeField32> TapeBuilder<F> {
    fn push(&mut self, n: [u32; 3]) -> u32 {
        self.nodes.push(n);
        (self.nodes.len() - 1) as u32
    }
    fn leaf(&mut self, n: [u32; 3]) -> u32 {
        if let Some(&id) = self.by_leaf.get(&n) {
            return id;
        }
        let id = self.push(n);
        self.by_leaf.insert(n, id);
        id
    }
    fn rc(&mut self, e: &Rc<SymbolicExpression<F>>) -> u32 {
        let key = Rc::as_ptr(e);
        if let Some(&id) = self.by_ptr.get(&key) {
            return id;
        }
        let id = self.expr(e);
        self.by_ptr.insert(key, id);
        id
    }
    fn expr(&mut self, e: &SymbolicExpression<F>) -> u32 {
        match e {
            SymbolicExpression::Variable(v) => match v.entry {
                Entry::Main { offset } => self.leaf([OP_MAIN, offset as u32, v.index as u32]),
                Entry::Public => self.leaf([OP_PUBLIC, v.index as u32, 0]),
                // the hot path has no preprocessed / permutation / challenge columns
                // (uni-stark/src/prover.rs:46 passes preprocessed_width = 0)
                other => panic!("unsupported symbolic variable on the prover hot path: {other:?}"),
            },
            SymbolicExpression::IsFirstRow => self.leaf([OP_IS_FIRST_ROW, 0, 0]),
            SymbolicExpression::IsLastRow => self.leaf([OP_IS_LAST_ROW, 0, 0]),
            SymbolicExpression::IsTransition => self.leaf([OP_IS_TRANSITION, 0, 0]),
            SymbolicExpression::Constant(c) => self.leaf([OP_CONST, c.as_canonical_u32(), 0]),
            SymbolicExpression::Add { x, y, .. } => {
                let (a, b) = (self.rc(x), self.rc(y));
                self.push([OP_ADD, a, b])
            }
            SymbolicExpression::Sub { x, y, .. } => {
                let (a, b) = (self.rc(x), self.rc(y));
                self.push([OP_SUB, a, b])
            }
            SymbolicExpression::Neg { x, .. } => {
                let a = self.rc(x);
                self.push([OP_NEG, a, 0])
            }
            SymbolicExpression::Mul { x, y, .. } => {
                let (a, b) = (self.rc(x), self.rc(y));
                self.push([OP_MUL, a, b])
            }
        }
    }
}

/// The tape of `air` for `num_public_values` public inputs.  `get_log_quotient_degree`
/// (symbolic_builder.rs:15-32) is recomputed by the library from the tape with the same degree
/// rules (symbolic_expression.rs:41-61; `ts_air_info`).
pub fn serialize_constraints<F, A>(air: &A, num_public_values: usize) -> Vec<u32>
where
    F: PrimeField32,
    A: Air<SymbolicAirBuilder<F>>,
{
    let constraints: Vec<SymbolicExpression<F>> = get_symbolic_constraints(air, 0, num_public_values);
    let mut tb = TapeBuilder::<F> { nodes: Vec::new(), by_ptr: HashMap::new(), by_leaf: HashMap::new() };
    let roots: Vec<u32> = constraints.iter().map(|c| tb.expr(c)).collect();
    let mut tape = Vec::with_capacity(6 + 3 * tb.nodes.len() + roots.len());
    tape.extend_from_slice(&[
        TAPE_MAGIC,
        1,
        air.width() as u32,
        num_public_values as u32,
        tb.nodes.len() as u32,
        roots.len() as u32,
    ]);
    for n in &tb.nodes {
        tape.extend_from_slice(n);
    }
    tape.extend_from_slice(&roots);
    tape
}
