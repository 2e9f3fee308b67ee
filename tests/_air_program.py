"""Test-side interpreter of the product's lowered register program (``ts_air_program``,
tap-stark_amd/csrc/air.cpp) in numpy: what the on-device interpreter (csrc/quotient.hip k_quotient)
does per row, vectorised over m rows.  TEST INFRASTRUCTURE: lets the CPU suite check the
tape -> register program lowering (scheduling, register reuse, constant table) against the oracle's
direct DAG evaluation without a GPU."""
import numpy as np

P = 0x78000001
D_LOAD, D_CONST, D_SEL, D_ADD, D_SUB, D_NEG, D_MUL, D_ASSERT = range(8)


def run_program(prog: dict, local: np.ndarray, nxt: np.ndarray, pis, sels: np.ndarray, n_constraints: int):
    """(m, n_constraints) constraint values; local/nxt (m, w) canonical, sels (m, 3)."""
    m = local.shape[0]
    p = np.uint64(P)
    consts = [int(pis[pi]) if pi != 0xFFFFFFFF else int(v) for v, pi in zip(prog["consts"], prog["const_public"])]
    regs = np.zeros((prog["n_regs"], m), dtype=np.uint64)
    written = np.zeros(prog["n_regs"], dtype=bool)
    out = np.zeros((m, n_constraints), dtype=np.uint32)
    seen = np.zeros(n_constraints, dtype=bool)
    rows = (local.astype(np.uint64), nxt.astype(np.uint64))
    sels = sels.astype(np.uint64)
    for op, dst, a, b in prog["code"].tolist():
        if op == D_LOAD:
            v = rows[a][:, b]
        elif op == D_CONST:
            v = np.full(m, consts[a], dtype=np.uint64)
        elif op == D_SEL:
            v = sels[:, a]
        elif op == D_ASSERT:
            assert written[a] and not seen[b]
            seen[b] = True
            out[:, b] = regs[a]
            continue
        else:
            assert written[a] and (op == D_NEG or written[b]), "read of a register never written"
            if op == D_ADD:
                v = (regs[a] + regs[b]) % p
            elif op == D_SUB:
                v = (regs[a] + p - regs[b]) % p
            elif op == D_NEG:
                v = (p - regs[a]) % p
            elif op == D_MUL:
                v = (regs[a] * regs[b]) % p
            else:
                raise AssertionError(f"unknown op {op}")
        regs[dst] = v
        written[dst] = True
    assert seen.all(), "a constraint was never asserted"
    return out
