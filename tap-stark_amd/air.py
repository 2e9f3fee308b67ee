"""Symbolic AIR capture -> constraint tape.

Host-side mirror of the reference's symbolic constraint capture, which is how an
``Air::eval`` body (user code) reaches the prover:

* ``SymbolicVariable`` / ``Entry``    -- reference uni-stark/src/symbolic_variable.rs:9-38
* ``SymbolicExpression``              -- reference uni-stark/src/symbolic_expression.rs:12-61
  (degree rules at :41-61, :137 add, :182 sub, :227 mul)
* ``SymbolicAirBuilder``              -- reference uni-stark/src/symbolic_builder.rs:68-148
* ``get_symbolic_constraints`` / ``get_max_constraint_degree`` / ``get_log_quotient_degree``
                                      -- reference uni-stark/src/symbolic_builder.rs:15-64
* ``FilteredAirBuilder`` (``when_first_row`` ...) -- p3-air semantics, SURVEY.md App. A.7

The DAG is serialised into the "tape" the C ABI takes (``include/tapstark.h``, TS_OP_*):
``[magic, version, width, n_public, n_nodes, n_constraints, nodes(op,a,b)..., constraint ids...]``.
Python is only the capture front-end; the tape is evaluated by HIP kernels.
"""
from __future__ import annotations

import numpy as np

P = 0x78000001  # reference basic/src/field/mod.rs:45

TAPE_MAGIC = 0x54415354
OP_CONST, OP_MAIN, OP_PUBLIC, OP_IS_FIRST, OP_IS_LAST, OP_IS_TRANSITION = 0, 1, 2, 3, 4, 5
OP_ADD, OP_SUB, OP_NEG, OP_MUL = 6, 7, 8, 9


class SymbolicExpression:
    """Node of the constraint DAG (hash-consed per builder so shared sub-expressions are
    emitted once, like the reference's ``Rc`` sharing)."""

    __slots__ = ("b", "id", "degree_multiple")

    def __init__(self, b: "SymbolicAirBuilder", node_id: int, degree_multiple: int):
        self.b = b
        self.id = node_id
        self.degree_multiple = degree_multiple

    def _lift(self, other) -> "SymbolicExpression":
        if isinstance(other, SymbolicExpression):
            return other
        return self.b.constant(int(other))

    def __add__(self, o):
        o = self._lift(o)
        return self.b._node(OP_ADD, self.id, o.id, max(self.degree_multiple, o.degree_multiple))

    __radd__ = lambda self, o: self._lift(o).__add__(self)

    def __sub__(self, o):
        o = self._lift(o)
        return self.b._node(OP_SUB, self.id, o.id, max(self.degree_multiple, o.degree_multiple))

    def __rsub__(self, o):
        return self._lift(o).__sub__(self)

    def __neg__(self):
        return self.b._node(OP_NEG, self.id, 0, self.degree_multiple)

    def __mul__(self, o):
        o = self._lift(o)
        return self.b._node(OP_MUL, self.id, o.id, self.degree_multiple + o.degree_multiple)

    __rmul__ = lambda self, o: self._lift(o).__mul__(self)


class _Row:
    def __init__(self, exprs):
        self._e = exprs

    def __getitem__(self, i):
        return self._e[i]

    def __len__(self):
        return len(self._e)

    def __iter__(self):
        return iter(self._e)


class _MainWindow:
    """``builder.main()``: two-row window; ``row_slice(0)`` = local, ``row_slice(1)`` = next."""

    def __init__(self, rows):
        self._rows = rows

    def row_slice(self, offset: int) -> _Row:
        return self._rows[offset]


class FilteredAirBuilder:
    """p3-air ``FilteredAirBuilder`` (App. A.7): ``when(c).assert_zero(x)`` => ``assert_zero(c*x)``."""

    def __init__(self, inner, condition):
        self.inner = inner
        self.condition = condition

    def assert_zero(self, x):
        self.inner.assert_zero(self.condition * x)

    def assert_eq(self, x, y):
        self.assert_zero(self.inner._lift(x) - y)

    def assert_one(self, x):
        self.assert_zero(self.inner._lift(x) - 1)

    def when(self, c):
        return FilteredAirBuilder(self.inner, self.condition * c)


class SymbolicAirBuilder:
    """reference uni-stark/src/symbolic_builder.rs:68-148."""

    def __init__(self, width: int, num_public_values: int):
        self.width = width
        self.num_public_values = num_public_values
        self.nodes: list[tuple[int, int, int]] = []
        self._degs: list[int] = []
        self._cse: dict[tuple[int, int, int], int] = {}
        self.constraints: list[int] = []
        self._main = _MainWindow(
            [
                _Row([self._node(OP_MAIN, off, c, 1) for c in range(width)])
                for off in (0, 1)
            ]
        )
        self._public = [self._node(OP_PUBLIC, i, 0, 0) for i in range(num_public_values)]

    # -- DAG -----------------------------------------------------------------
    def _node(self, op, a, b, deg) -> SymbolicExpression:
        key = (op, a, b)
        nid = self._cse.get(key)
        if nid is None:
            nid = len(self.nodes)
            self.nodes.append(key)
            self._degs.append(deg)
            self._cse[key] = nid
        return SymbolicExpression(self, nid, self._degs[nid])

    def _lift(self, x) -> SymbolicExpression:
        return x if isinstance(x, SymbolicExpression) else self.constant(int(x))

    def constant(self, v: int) -> SymbolicExpression:
        return self._node(OP_CONST, v % P, 0, 0)

    # -- AirBuilder surface (symbolic_builder.rs:110-139) -----------------------
    def main(self) -> _MainWindow:
        return self._main

    def public_values(self):
        return self._public

    def is_first_row(self):
        return self._node(OP_IS_FIRST, 0, 0, 1)  # symbolic_expression.rs:45

    def is_last_row(self):
        return self._node(OP_IS_LAST, 0, 0, 1)  # :46

    def is_transition(self):
        return self.is_transition_window(2)

    def is_transition_window(self, size: int):
        if size != 2:
            raise ValueError("uni-stark only supports a window size of 2")  # :131
        return self._node(OP_IS_TRANSITION, 0, 0, 0)  # symbolic_expression.rs:47

    def assert_zero(self, x):
        self.constraints.append(self._lift(x).id)  # symbolic_builder.rs:136-138

    def assert_eq(self, x, y):
        self.assert_zero(self._lift(x) - y)

    def assert_one(self, x):
        self.assert_zero(self._lift(x) - 1)

    def when(self, c):
        return FilteredAirBuilder(self, self._lift(c))

    def when_first_row(self):
        return self.when(self.is_first_row())

    def when_last_row(self):
        return self.when(self.is_last_row())

    def when_transition(self):
        return self.when(self.is_transition())

    # -- serialisation -----------------------------------------------------------
    def max_constraint_degree(self) -> int:
        return max((self._degs[c] for c in self.constraints), default=0)

    def tape(self) -> np.ndarray:
        words = [TAPE_MAGIC, 1, self.width, self.num_public_values, len(self.nodes),
                 len(self.constraints)]
        for op, a, b in self.nodes:
            words += [op, a, b]
        words += self.constraints
        return np.asarray(words, dtype=np.uint32)


class BaseAir:
    """p3-air ``BaseAir``: subclasses give ``width()`` and ``eval(builder)``."""

    def width(self) -> int:  # pragma: no cover - interface
        raise NotImplementedError

    def eval(self, builder) -> None:  # pragma: no cover - interface
        raise NotImplementedError


def get_symbolic_constraints(air: BaseAir, num_public_values: int) -> SymbolicAirBuilder:
    """reference uni-stark/src/symbolic_builder.rs:52-64 (returns the builder holding them)."""
    b = SymbolicAirBuilder(air.width(), num_public_values)
    air.eval(b)
    return b


def get_max_constraint_degree(air: BaseAir, num_public_values: int) -> int:
    return get_symbolic_constraints(air, num_public_values).max_constraint_degree()


def log2_ceil(n: int) -> int:
    return max(0, (n - 1).bit_length())


def get_log_quotient_degree(air: BaseAir, num_public_values: int) -> int:
    """reference uni-stark/src/symbolic_builder.rs:15-32."""
    d = max(get_max_constraint_degree(air, num_public_values), 2)
    return log2_ceil(d - 1)


def air_tape(air: BaseAir, num_public_values: int) -> np.ndarray:
    return get_symbolic_constraints(air, num_public_values).tape()
