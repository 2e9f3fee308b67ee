"""tap-stark hot path, MI355X-native: host-side mirror of the reference's uni-stark/fri
prover interface over the C-ABI HIP library (include/tapstark.h)."""
from . import air, airs  # noqa: F401
from .air import (BaseAir, SymbolicAirBuilder, air_tape, get_log_quotient_degree,  # noqa: F401
                  get_max_constraint_degree, get_symbolic_constraints)
from .stark import (BfChallenger, Blake3Mmcs, CompiledAir, Context, DeviceMatrix, FriConfig, PcsData, PinnedHostMatrix,  # noqa: F401
                    Proof, StarkConfig, TwoAdicFriPcs, VerificationError, check_constraints,
                    default_context, prove, prove_sharded, prove_stream, verify)
