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
