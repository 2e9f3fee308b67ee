"""The AIRs named by BASELINE.json's configs, with their trace generators.

* ``FibonacciAir`` / ``generate_fibonacci_trace`` restate the reference's only live AIR,
  reference uni-stark/tests/fib_air.rs:21-78.
* ``SynthMulAir`` and ``SynthExtAir`` are BUILD-DEFINED synthetic stand-ins (SURVEY.md F5,
  section 8(d)): the reference has no "synthetic random AIR" nor a RISC0-recursion AIR.
  ``SynthMulAir``'s shape follows the commented-out reference uni-stark/tests/mul_air.rs:29-116.
"""
from __future__ import annotations

import numpy as np

from .air import BaseAir, P

SPLITMIX_SEED = 0x7A957A12


class FibonacciAir(BaseAir):
    """reference uni-stark/tests/fib_air.rs:21-57. Public values = [a, b, x]."""

    NUM_FIBONACCI_COLS = 2

    def width(self) -> int:
        return self.NUM_FIBONACCI_COLS

    def eval(self, builder) -> None:
        main = builder.main()
        pis = builder.public_values()
        a, b, x = pis[0], pis[1], pis[2]
        local, nxt = main.row_slice(0), main.row_slice(1)
        left, right = 0, 1

        when_first_row = builder.when_first_row()
        when_first_row.assert_eq(local[left], a)
        when_first_row.assert_eq(local[right], b)

        when_transition = builder.when_transition()
        # a' <- b
        when_transition.assert_eq(local[right], nxt[left])
        # b' <- a + b
        when_transition.assert_eq(local[left] + local[right], nxt[right])

        builder.when_last_row().assert_eq(local[right], x)


def generate_fibonacci_trace(a: int, b: int, n: int) -> np.ndarray:
    """reference uni-stark/tests/fib_air.rs:59-78 -> row-major (n, 2) canonical u32."""
    assert n & (n - 1) == 0
    t = np.zeros((n, 2), dtype=np.uint32)
    l, r = a % P, b % P
    for i in range(n):
        t[i, 0], t[i, 1] = l, r
        l, r = r, (l + r) % P
    return t


def fibonacci_public_values(trace: np.ndarray) -> np.ndarray:
    """fib_air.rs:133-139: [0, 1, trace.values[len-1]] for the (0,1) start."""
    return np.array([trace[0, 0], trace[0, 1], trace[-1, 1]], dtype=np.uint32)


def splitmix64_stream(seed: int, count: int) -> np.ndarray:
    """SplitMix64, value = next() mod p."""
    mask = (1 << 64) - 1
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(mask)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z % np.uint64(P)).astype(np.uint32)


class SynthMulAir(BaseAir):
    """Build-defined "SynthMulAir-w": ``reps`` triples (a, b, c) + free columns up to ``w``.

    Per triple (shape from the commented reference mul_air.rs:29-116):
      * ``a*a*b - c = 0``                      (degree 3 => quotient_degree 2)
      * first row:  ``a*a + 1 = b``
      * transition: ``a + reps = a'``
    """

    def __init__(self, width: int = 64):
        self._w = width
        self.reps = width // 3

    def width(self) -> int:
        return self._w

    def eval(self, builder) -> None:
        main = builder.main()
        local, nxt = main.row_slice(0), main.row_slice(1)
        for i in range(self.reps):
            a, b, c = local[3 * i], local[3 * i + 1], local[3 * i + 2]
            builder.assert_zero(a * a * b - c)
            builder.when_first_row().assert_eq(a * a + 1, b)
            builder.when_transition().assert_eq(a + self.reps, nxt[3 * i])


def generate_synth_mul_trace(n: int, width: int = 64, seed: int = SPLITMIX_SEED) -> np.ndarray:
    """Valid trace for ``SynthMulAir(width)``: a = reps*row + k, b random (row 0: a*a+1),
    c = a*a*b, remaining columns random."""
    reps = width // 3
    free = width - 3 * reps
    rnd = splitmix64_stream(seed, n * (reps + free)).reshape(n, reps + free).astype(np.uint64)
    t = np.zeros((n, width), dtype=np.uint64)
    rows = np.arange(n, dtype=np.uint64)
    p = np.uint64(P)
    for k in range(reps):
        a = (np.uint64(reps) * rows + np.uint64(k)) % p
        b = rnd[:, k].copy()
        b[0] = (a[0] * a[0] + np.uint64(1)) % p
        c = (((a * a) % p) * b) % p
        t[:, 3 * k], t[:, 3 * k + 1], t[:, 3 * k + 2] = a, b, c
    for f in range(free):
        t[:, 3 * reps + f] = rnd[:, reps + f]
    return t.astype(np.uint32)


class HighDegreeAir(BaseAir):
    """Build-defined two-column AIR whose one constraint has degree ``degree``: ``a^(degree-1) * b = 0``
    with b = 0 (so quotient_degree = 2^ceil(log2(degree - 1)): 32 for degree 33) plus the transition
    ``a + 1 = a'``.  Exercises quotient degrees above 16 (``log_quotient_degree <= log_blowup``)."""

    def __init__(self, degree: int = 33):
        self.degree = degree

    def width(self) -> int:
        return 2

    def eval(self, builder) -> None:
        main = builder.main()
        local, nxt = main.row_slice(0), main.row_slice(1)
        acc = local[0]
        for _ in range(self.degree - 2):
            acc = acc * local[0]
        builder.assert_zero(acc * local[1])
        builder.when_transition().assert_eq(local[0] + 1, nxt[0])


def generate_high_degree_trace(n: int) -> np.ndarray:
    t = np.zeros((n, 2), dtype=np.uint32)
    t[:, 0] = (np.arange(n, dtype=np.uint64) + np.uint64(5)) % np.uint64(P)
    return t


class SynthExtAir(BaseAir):
    """Build-defined "SynthExt-w" stand-in for the RISC0-recursion-style config (SURVEY.md
    section 8(d) config 5): columns are ``groups`` triples of EF4 elements (x, y, z: 12 base
    columns per group) plus base columns; constraints ``x*y - z = 0`` in F[X]/(X^4-11)
    (4 base constraints of degree 2 per group) and a base transition on the last column.
    """

    W = 11

    def __init__(self, width: int = 163):
        self._w = width
        self.groups = (width - 1) // 12

    def width(self) -> int:
        return self._w

    def eval(self, builder) -> None:
        main = builder.main()
        local, nxt = main.row_slice(0), main.row_slice(1)
        for g in range(self.groups):
            base = 12 * g
            x = [local[base + i] for i in range(4)]
            y = [local[base + 4 + i] for i in range(4)]
            z = [local[base + 8 + i] for i in range(4)]
            for k in range(4):
                acc = None
                for i in range(4):
                    for j in range(4):
                        if (i + j) % 4 != k:
                            continue
                        term = x[i] * y[j]
                        if i + j >= 4:
                            term = term * self.W
                        acc = term if acc is None else acc + term
                builder.assert_zero(acc - z[k])
        last = self._w - 1
        builder.when_transition().assert_eq(local[last] + 1, nxt[last])


def generate_synth_ext_trace(n: int, width: int = 163, seed: int = SPLITMIX_SEED) -> np.ndarray:
    groups = (width - 1) // 12
    p = np.uint64(P)
    rnd = splitmix64_stream(seed, n * width).reshape(n, width).astype(np.uint64)
    t = rnd.copy()
    for g in range(groups):
        base = 12 * g
        x = [t[:, base + i] for i in range(4)]
        y = [t[:, base + 4 + i] for i in range(4)]
        for k in range(4):
            acc = np.zeros(n, dtype=np.uint64)
            for i in range(4):
                for j in range(4):
                    if (i + j) % 4 != k:
                        continue
                    term = (x[i] * y[j]) % p
                    if i + j >= 4:
                        term = (term * np.uint64(SynthExtAir.W)) % p
                    acc = (acc + term) % p
            t[:, base + 8 + k] = acc
    t[:, width - 1] = (np.uint64(7) + np.arange(n, dtype=np.uint64)) % p
    return t.astype(np.uint32)


# ---------------------------------------------------------------------------- random AIRs
# The reference's prove() is generic over `Air` (uni-stark/src/prover.rs:25-39): whatever an
# `Air::eval` body writes against the builder reaches the quotient evaluation through
# get_symbolic_constraints (symbolic_builder.rs:52-64).  RandomAir is a seeded family of such
# bodies for fuzzing the two constraint compilers (csrc/air.cpp, csrc/jit.cpp) and the verifier's
# tape evaluation against the oracle: shared sub-terms, long live ranges, selectors inside and
# outside products, public values in high-degree terms, edge constants.

class _Rng:
    """SplitMix64 (own implementation, so that a seed means the same AIR on every Python)."""

    M = (1 << 64) - 1

    def __init__(self, seed: int):
        self.s = (seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & self.M

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & self.M
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.M
        return z ^ (z >> 31)

    def below(self, n: int) -> int:
        return self.next() % n

    def field(self) -> int:
        return self.next() % P


EDGE_CONSTANTS = (0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 11, 31, 1 << 27, P - (1 << 27))


class _NumExpr:
    """Value of an expression on every row of a concrete trace (numpy, canonical u64) together with
    its degree_multiple under the reference's rules (symbolic_expression.rs:41-61,137,182,227): the
    numeric twin of SymbolicExpression, used to fill the defined columns of a valid RandomAir trace."""

    __slots__ = ("b", "v", "degree_multiple")

    def __init__(self, b, v, deg):
        self.b, self.v, self.degree_multiple = b, v, deg

    def _lift(self, o):
        return o if isinstance(o, _NumExpr) else self.b.constant(int(o))

    def __add__(self, o):
        o = self._lift(o)
        return _NumExpr(self.b, (self.v + o.v) % np.uint64(P), max(self.degree_multiple, o.degree_multiple))

    __radd__ = lambda self, o: self._lift(o).__add__(self)

    def __sub__(self, o):
        o = self._lift(o)
        return _NumExpr(self.b, (self.v + np.uint64(P) - o.v) % np.uint64(P),
                        max(self.degree_multiple, o.degree_multiple))

    def __rsub__(self, o):
        return self._lift(o).__sub__(self)

    def __neg__(self):
        return _NumExpr(self.b, (np.uint64(P) - self.v) % np.uint64(P), self.degree_multiple)

    def __mul__(self, o):
        o = self._lift(o)
        return _NumExpr(self.b, (self.v * o.v) % np.uint64(P), self.degree_multiple + o.degree_multiple)

    __rmul__ = lambda self, o: self._lift(o).__mul__(self)


class NumericBuilder:
    """Builder over a concrete trace with the row semantics of check_constraints
    (uni-stark/src/check_constraints.rs:18-38: is_first = (i == 0), is_last = (i == h-1),
    is_transition = (i != h-1), next row wrapping).  `constraints` collects one value vector per
    assert_zero, so a trace can be checked in numpy; RandomAir also writes defined columns through it."""

    def __init__(self, trace: np.ndarray, public_values, define: bool = True):
        self.trace = trace  # (n, w) uint64, canonical; with define=True RandomAir fills defined columns in
        self.define = define
        self.n = trace.shape[0]
        self.pis = [int(v) for v in public_values]
        self.constraints = []
        rows = np.arange(self.n)
        self._first = (rows == 0).astype(np.uint64)
        self._last = (rows == self.n - 1).astype(np.uint64)

    def _col(self, off, c):
        col = self.trace[:, c]
        return _NumExpr(self, np.roll(col, -1) if off else col.copy(), 1)

    def main(self):
        b = self

        class _W:
            def row_slice(self, off):
                class _R:
                    def __getitem__(self, c):
                        return b._col(off, c)

                    def __len__(self):
                        return b.trace.shape[1]

                return _R()

        return _W()

    def public_values(self):
        return [_NumExpr(self, np.full(self.n, v, dtype=np.uint64), 0) for v in self.pis]

    def constant(self, v):
        return _NumExpr(self, np.full(self.n, v % P, dtype=np.uint64), 0)

    def _lift(self, x):
        return x if isinstance(x, _NumExpr) else self.constant(int(x))

    def is_first_row(self):
        return _NumExpr(self, self._first.copy(), 1)

    def is_last_row(self):
        return _NumExpr(self, self._last.copy(), 1)

    def is_transition(self):
        return _NumExpr(self, np.uint64(1) - self._last, 0)

    def is_transition_window(self, size):
        assert size == 2
        return self.is_transition()

    def assert_zero(self, x):
        self.constraints.append(self._lift(x).v)

    def assert_eq(self, x, y):
        self.assert_zero(self._lift(x) - y)

    def assert_one(self, x):
        self.assert_zero(self._lift(x) - 1)

    def when(self, c):
        from .air import FilteredAirBuilder

        return FilteredAirBuilder(self, self._lift(c))

    def when_first_row(self):
        return self.when(self.is_first_row())

    def when_last_row(self):
        return self.when(self.is_last_row())

    def when_transition(self):
        return self.when(self.is_transition())

    def first_violation(self) -> int:
        """-1, or row * 65536 + constraint index of the first failure (check_constraints order)."""
        if not self.constraints:
            return -1
        bad = np.stack(self.constraints, axis=1) != 0  # (n, K)
        rows = np.flatnonzero(bad.any(axis=1))
        if len(rows) == 0:
            return -1
        r = int(rows[0])
        return r * 65536 + int(np.flatnonzero(bad[r])[0])


class RandomAir(BaseAir):
    """Seeded random AIR: ``n_constraints`` constraints of degree <= ``max_degree`` over ``width``
    columns and ``n_public`` public values.

    ``valid=False``  free-form expression DAGs (any leaf anywhere).  No trace satisfies them; the
                     quotient VALUES on a random trace are still a pure function of (tape, LDE,
                     alpha), which is what the compilers are compared on.
    ``valid=True``   structured: ``n_state`` recurrence columns (first/transition/last constraints,
                     public values = their first and last rows), free random columns, and columns
                     DEFINED as random polynomials of earlier columns (local and next row); further
                     constraints are valid ones times arbitrary expressions and multiples of
                     ``is_last * is_transition`` (= Z_H).  Selectors appear only as factors of a
                     vanishing expression, so ``generate_random_air_trace`` gives a trace the
                     reference verifier accepts.  Needs ``n_public >= 2 * n_state``.
    """

    def __init__(self, seed: int, width: int, n_constraints: int, max_degree: int, n_public: int = 3,
                 valid: bool = False, share_pct: int = 35, max_depth: int = 6):
        assert width >= 1 and n_constraints >= 1 and max_degree >= 1
        if valid:
            max_degree = max(max_degree, 2)  # is_first_row * (column - public value) is already degree 2
        self.seed, self._w, self.n_constraints, self.max_degree = seed, width, n_constraints, max_degree
        self.valid, self.share_pct, self.max_depth = valid, share_pct, max_depth
        if valid:
            r = _Rng(seed ^ 0x5EED)
            self.n_state = min(width, 1 + r.below(3))
            self.n_free = min(width - self.n_state, r.below(1 + max(1, width // 3)))
            n_public = max(n_public, 2 * self.n_state)
        self.n_public = n_public

    def width(self) -> int:
        return self._w

    # -- expression generator ---------------------------------------------------------------------
    def _setup(self, builder):
        self._rng = _Rng(self.seed)
        self._b = builder
        main = builder.main()
        self._local, self._next = main.row_slice(0), main.row_slice(1)
        self._pis = builder.public_values()
        self._pool = [[] for _ in range(self.max_degree + 1)]  # sub-terms by degree, oldest first

    def _constant(self):
        r = self._rng
        v = EDGE_CONSTANTS[r.below(len(EDGE_CONSTANTS))] if r.below(2) else r.field()
        return self._b.constant(v)

    def _leaf(self, d: int, max_col: int, selectors: bool):
        r, k = self._rng, self._rng.below(100)
        if d >= 1 and k < 70:
            if selectors and k < 12:
                return self._b.is_first_row() if k < 6 else self._b.is_last_row()
            c = r.below(max_col)
            return (self._next if r.below(3) == 0 else self._local)[c]
        if selectors is True and k < 78:
            return self._b.is_transition()
        if self._pis and k < 90:
            return self._pis[r.below(len(self._pis))]
        return self._constant()

    def _gen(self, d: int, depth: int, max_col: int, selectors: bool):
        """Expression of degree_multiple <= d over columns < max_col.  selectors: False none, True all,
        "rows" only is_first_row / is_last_row -- is_transition counts as degree 0
        (symbolic_expression.rs:47) although it is x - omega^-1, so a valid AIR may multiply it in only
        fewer than max_degree times or the quotient outgrows its qd * n coefficients."""
        r = self._rng
        k = r.below(100)
        if depth <= 0 or k < 18:
            return self._leaf(d, max_col, selectors)
        if k < 18 + self.share_pct:
            dd = r.below(d + 1)
            for deg in range(dd, -1, -1):
                pool = self._pool[deg]
                if pool:
                    # a third of the picks take one of the oldest entries: long live ranges
                    idx = r.below(min(len(pool), 4)) if r.below(3) == 0 else r.below(len(pool))
                    return pool[idx]
            return self._leaf(d, max_col, selectors)
        if k < 80 and d >= 1:
            d1 = r.below(d + 1)
            e = self._gen(d1, depth - 1, max_col, selectors) * self._gen(d - d1, depth - 1, max_col, selectors)
        elif k < 88:
            e = -self._gen(d, depth - 1, max_col, selectors)
        elif k < 94:
            e = self._gen(d, depth - 1, max_col, selectors) - self._gen(r.below(d + 1), depth - 1, max_col, selectors)
        else:
            e = self._gen(d, depth - 1, max_col, selectors) + self._gen(r.below(d + 1), depth - 1, max_col, selectors)
        if e.degree_multiple <= self.max_degree:
            self._pool[e.degree_multiple].append(e)
        return e

    def _full_degree_product(self, max_col: int):
        """Exactly max_degree main variables multiplied together."""
        r = self._rng
        e = self._local[r.below(max_col)]
        for _ in range(self.max_degree - 1):
            e = e * (self._next if r.below(4) == 0 else self._local)[r.below(max_col)]
        return e

    # -- eval ---------------------------------------------------------------------------------------
    def eval(self, builder) -> None:
        self._setup(builder)
        if self.valid:
            self._eval_valid(builder)
        else:
            self._eval_free(builder)

    def _eval_free(self, b) -> None:
        r, D, w = self._rng, self.max_degree, self._w
        for k in range(self.n_constraints):
            if k == self.n_constraints - 1:  # pins max_constraint_degree to D
                b.assert_zero(self._full_degree_product(w) - self._gen(D, 3, w, True))
                break
            d = 1 + r.below(D)
            form = r.below(8)
            g = lambda dd: self._gen(max(dd, 0), self.max_depth, w, True)
            if form == 0:
                b.when_first_row().assert_eq(g(d - 1), g(d - 1))
            elif form == 1:
                b.when_last_row().assert_zero(g(d - 1))
            elif form == 2:
                b.when_transition().assert_eq(g(d), g(d))
            elif form == 3 and d >= 2:
                b.when_first_row().when(g(1)).assert_zero(g(d - 2))
            elif form == 4:
                b.assert_one(g(d))
            elif form == 5:
                b.when(g(d // 2)).assert_eq(g(d - d // 2), r.field())
            else:
                b.assert_zero(g(d))

    def _state_step(self, c: int, local):
        """next value of state column c as an expression of the local row (degree <= 2)."""
        kind, a, k = self._state_kinds[c]
        if kind == 0 or self.max_degree < 2:
            return local[c] * a + k
        if kind == 1:
            return local[c] * local[c] + k
        return local[c] * local[(c + 1) % self.n_state] + a

    def _eval_valid(self, b) -> None:
        r, D, w, s, f = self._rng, self.max_degree, self._w, self.n_state, self.n_free
        local, nxt, pis = self._local, self._next, self._pis
        self._state_kinds = [(r.below(3), 1 + r.field() % (P - 1), r.field()) for _ in range(s)]
        zeros = []  # (expression vanishing on every row, its degree)
        n_emitted = 0

        def emit(fn):
            nonlocal n_emitted
            fn()
            n_emitted += 1

        for c in range(s):
            emit(lambda: b.when_first_row().assert_eq(local[c], pis[c]))
            step = self._state_step(c, local)
            emit(lambda: b.when_transition().assert_eq(nxt[c], step))
            emit(lambda: b.when_last_row().assert_eq(local[c], pis[s + c]))
            z = b.is_transition() * (nxt[c] - step)
            zeros.append((z, z.degree_multiple))
        for j in range(s + f, w):
            filt = r.below(5)
            budget = D - (1 if filt in (1, 2) else 0)
            if budget < 1:
                filt, budget = 0, D
            e = self._gen(budget, self.max_depth, j, False) if j > 0 else self._constant()
            self._define(b, j, e)
            z = e - local[j]
            if filt == 1:
                emit(lambda: b.when_first_row().assert_zero(z))
            elif filt == 2:
                emit(lambda: b.when_last_row().assert_eq(e, local[j]))
            elif filt == 3:
                emit(lambda: b.when_transition().assert_zero(z))
            if filt != 3 or r.below(2):
                emit(lambda: b.assert_zero(z))
            zeros.append((z, max(z.degree_multiple, 1)))
        zh = b.is_last_row() * b.is_transition()  # = Z_H(x): vanishes on the whole trace domain
        zeros.append((zh, 1))
        if self.max_degree >= 2:
            zeros.append((b.is_first_row() * b.is_last_row(), 2))  # vanishes on H for n >= 2
        pinned = False
        while n_emitted < self.n_constraints or not pinned:
            last = n_emitted >= self.n_constraints - 1
            z, dz = zeros[r.below(len(zeros))]
            if last or not pinned and r.below(8) == 0:
                # a term of full degree that still vanishes: Z_H * (D-1 variables)
                if D >= 2:
                    e = zh
                    for _ in range(D - 1):
                        e = e * (nxt if r.below(4) == 0 else local)[r.below(w)]
                    emit(lambda: b.assert_zero(e + z * self._gen(max(D - dz, 0), 3, w, "rows")
                                               if dz <= D else e))
                else:
                    emit(lambda: b.assert_zero(zh))
                pinned = True
                continue
            if dz > D:
                continue
            g = self._gen(D - dz, self.max_depth, w, "rows")
            emit(lambda: b.assert_zero(z * g if r.below(2) else g * z))

    def _define(self, b, j: int, e) -> None:
        if isinstance(b, NumericBuilder) and b.define:
            b.trace[:, j] = e.v


def generate_random_air_trace(air: RandomAir, n: int, seed: int | None = None):
    """(trace, public_values) satisfying ``air`` (``valid=True``) on n >= 2 rows."""
    assert air.valid and n >= 2 and n & (n - 1) == 0
    r = _Rng((air.seed if seed is None else seed) ^ 0x7ACE)
    w, s, f = air.width(), air.n_state, air.n_free
    t = np.zeros((n, w), dtype=np.uint64)
    if f:
        t[:, s:s + f] = splitmix64_stream(r.next() & 0xFFFFFFFF, n * f).reshape(n, f)
    pis = [r.field() for _ in range(air.n_public)]
    # state recurrences, row by row (Python ints).  _state_kinds is a function of the seed: take it
    # from a dry symbolic run.
    from .air import SymbolicAirBuilder

    air.eval(SymbolicAirBuilder(w, air.n_public))
    kinds = list(air._state_kinds)
    cur = [pis[c] for c in range(s)]
    for i in range(n):
        t[i, :s] = cur
        nx = []
        for c in range(s):
            kind, a, k = kinds[c]
            if kind == 0 or air.max_degree < 2:
                nx.append((cur[c] * a + k) % P)
            elif kind == 1:
                nx.append((cur[c] * cur[c] + k) % P)
            else:
                nx.append((cur[c] * cur[(c + 1) % s] + a) % P)
        cur = nx
    for c in range(s):
        pis[s + c] = int(t[n - 1, c])
    nb = NumericBuilder(t, pis)
    air.eval(nb)  # fills the defined columns in order
    return t.astype(np.uint32), np.asarray(pis, dtype=np.uint32), nb


def random_air_case(seed: int):
    """The fuzz campaign's case table: seed -> (air, log_n).  Degrees 1..9 (quotient degree 1, 2, 4,
    8), widths 1..200, 1..3000 constraints; every third case is a valid-trace AIR."""
    r = _Rng(seed ^ 0xF022)
    D = (1, 2, 2, 3, 3, 3, 4, 5, 5, 6, 7, 8, 9, 9)[r.below(14)]
    size = r.below(100)
    if size < 60:
        width, nc = 1 + r.below(12), 1 + r.below(24)
    elif size < 90:
        width, nc = 1 + r.below(64), 1 + r.below(200)
    elif size < 98:
        width, nc = 1 + r.below(200), 200 + r.below(800)
    else:
        width, nc = 100 + r.below(101), 1000 + r.below(2001)
    valid = seed % 3 == 0
    air = RandomAir(seed, width, nc, D, n_public=r.below(5), valid=valid,
                    share_pct=(10, 35, 60)[r.below(3)], max_depth=3 + r.below(5))
    log_n = 1 + r.below(6) if width * nc < 20000 else 1 + r.below(3)
    return air, log_n
