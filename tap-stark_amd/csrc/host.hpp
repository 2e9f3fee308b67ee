// Host-side C++ mirror of the reference interface for the prover hot path: StarkConfig /
// TwoAdicFriPcs / FriConfig / BfChallenger / prove(), sitting on the HIP kernels.  The names and
// argument meaning follow the reference (uni-stark/src/prover.rs, fri/src/two_adic_pcs.rs,
// fri/src/prover.rs, basic/src/challenger/mod.rs); device objects replace host matrices.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <functional>
#include <memory>
#include <vector>

#include "air.hpp"
#include "bb.hpp"
#include "chal_dev.hpp"
#include "context.hpp"
#include "kernels.hpp"

namespace ts {

// ------------------------------------------------------------------ challenger
// reference basic/src/challenger/mod.rs:67-84 BfChallenger<F, U32, P, 16>
class BfChallenger {
public:
    enum Permutation { Blake3Permutation = 0, TestReversePermutation = 1 };
    BfChallenger(int permutation, bool sample_ext)
        : permutation_(permutation), sample_ext_(sample_ext) {}

    void observe(uint32_t word);                   // mod.rs:183-194
    void observe_commitment(const uint32_t d[8]);  // mod.rs:197-223
    uint32_t sample_base();                        // mod.rs:269-279
    Ef sample_ext();                               // mod.rs:282-304 (canonical EF4)
    Ef sample();                                   // F = EF4 or BabyBear (embedded)
    uint64_t sample_bits(unsigned bits);           // mod.rs:341-348
    bool check_witness(unsigned bits, uint32_t witness);  // mod.rs:108-114
    uint32_t grind(unsigned bits);                 // mod.rs:95-105 (throws TS_ERR_INVARIANT)
    void export_state(uint32_t out[34]) const;
    // hand the transcript to the device for the FRI commit phase, and take it back
    void export_dev(DevChallenger& d) const;
    void import_dev(const DevChallenger& d);

private:
    void duplexing();  // mod.rs:151-174
    uint32_t pop();
    uint32_t state_[16] = {0};
    uint32_t in_[8] = {0};
    int n_in_ = 0;
    uint32_t out_[8] = {0};
    int n_out_ = 0;
    int permutation_;
    bool sample_ext_;
};

// ------------------------------------------------------------------ config
// reference fri/src/config.rs:11-16
struct FriConfig {
    uint32_t log_blowup = 1;
    uint32_t num_queries = 1;
    uint32_t proof_of_work_bits = 0;
    uint32_t blowup() const { return 1u << log_blowup; }
};

// ------------------------------------------------------------------ device matrices
// RowMajorMatrix<Val> handed to the library.  ROW_MAJOR: natural rows, as uploaded.
// COL_MAJOR_BITREV: produced on the device (quotient chunks), column-major, bit-reversed rows.
struct DeviceMatrix {
    enum Layout { ROW_MAJOR, COL_MAJOR_BITREV };
    DevBuf<uint32_t> buf;
    uint64_t height = 0;
    uint32_t width = 0;
    Layout layout = ROW_MAJOR;
};

// Pcs::ProverData / BFMmcs::ProverData: the committed LDEs (column-major, bit-reversed rows) and
// the whole Merkle tree, all resident in HBM.
struct PcsData {
    std::vector<DevBuf<uint32_t>> lde_storage;
    std::vector<ColMat> ldes;
    unsigned log_height = 0;
    DevBuf<uint32_t> tree;  // merkle_total_digests(log_height) x 8 words
    DevBuf<const uint32_t*> col_table;  // one base pointer per column of the concatenated row
    bool col_table_uploaded = false;    // mmcs_commit uploads it only for the table-addressed leaf kernels
    uint32_t root[8] = {0};
    LeafMats leaf_mats() const;
};

unsigned log2_strict(uint64_t n);  // throws TS_ERR_INVALID unless n is a power of two

// BFMmcs::commit (basic/src/mmcs/bf_mmcs.rs:22-35) over data.ldes, already resident column-major;
// data.log_height = log2 of the tallest matrix.  Fills data.tree, data.col_table, data.root.
void mmcs_commit(Context& ctx, PcsData& data);

// ------------------------------------------------------------------ PCS (TwoAdicFriPcs)
class TwoAdicFriPcs {
public:
    TwoAdicFriPcs(Context& ctx, FriConfig fri) : ctx_(ctx), fri_(fri) {}
    const FriConfig& fri() const { return fri_; }
    Context& ctx() const { return ctx_; }

    // two_adic_pcs.rs:227-245.  Consumes the matrices.  build_tree = false stops after the LDE
    // (a caller with another MMCS -- the taptree one -- commits to data->ldes itself).
    std::unique_ptr<PcsData> commit(std::vector<DeviceMatrix>& evals,
                                    const std::vector<uint32_t>& domain_shifts, bool build_tree = true);

    // two_adic_pcs.rs:247-258 + uni-stark prover.rs:122-194,78-80
    std::vector<DeviceMatrix> quotient_chunks(const PcsData& trace_data, const AirProgram& air,
                                              const std::vector<uint32_t>& public_values, Ef alpha);

    // two_adic_pcs.rs:312-389 for the prove() shape; returns the FRI input (N EF4, device)
    DevBuf<Ef> open_reduce(const PcsData& trace_data, const PcsData& quotient_data, Ef zeta,
                           Ef batch_alpha, std::vector<Ef>& opened_values);

    // The same two steps on a slab of the LDE: global rows [row0, row0 + rows) = whole cosets
    // beta0, beta0+1, ... (bit-reversed coset order), as held by one rank of the sharded prover.
    struct Slab {
        uint64_t row0 = 0, rows = 0;  // rows = 0: the whole LDE
        uint32_t beta0 = 0;
    };
    // domain_shift: the shift s of the quotient domain s * H_{n qd} the slab's first rows are the
    // evaluations on (31 for the reference's own domain; a rank's own cosets otherwise)
    std::vector<DeviceMatrix> quotient_chunks_slab(const ColMat& lde_slab, unsigned log_n, const Slab& slab,
                                                   const AirProgram& air,
                                                   const std::vector<uint32_t>& public_values, Ef alpha,
                                                   uint32_t domain_shift = GENERATOR);
    DevBuf<Ef> open_reduce_slab(const PcsData& trace_data, const PcsData& quotient_data, unsigned log_N,
                                const Slab& slab, Ef zeta, Ef batch_alpha, std::vector<Ef>& opened_values);

    // two_adic_pcs.rs:260-419 for any rounds x matrices x points: samples the batch challenge,
    // computes the opened values ((round, matrix, point, column) order) and returns the FriProof
    // (TSPF v1 words from the commit-phase round count on).
    struct OpenRound {
        const PcsData* data = nullptr;
        std::vector<std::vector<Ef>> points;  // per matrix, canonical EF4
    };
    std::vector<uint32_t> open(const std::vector<OpenRound>& rounds, BfChallenger& challenger,
                               std::vector<Ef>& opened_values);

    // fri/src/prover.rs:19-141 bf_prove over the reduced openings (strictly descending heights),
    // with the input openings of two_adic_pcs.rs:399-414; appends the FriProof to `pf`
    // pass_through: no committed batches; the input proof of a query is the literal reduced
    // openings [(log_height, value)] as in fri/tests/fri.rs:109-118
    void fri_prove(std::vector<DevBuf<Ef>>& inputs, const std::vector<unsigned>& log_lens,
                   BfChallenger& challenger, const std::vector<const PcsData*>& input_rounds,
                   std::vector<uint32_t>& pf, bool pass_through = false);

    // BFMmcs::open_batch
    void open_batch(const PcsData& d, uint64_t index, std::vector<uint32_t>& rows,
                    std::vector<uint32_t>& path);

private:
    Context& ctx_;
    FriConfig fri_;
};

// ------------------------------------------------------------------ prove
// uni-stark/src/prover.rs:25-119.  Returns the proof in TSPF v1 words.
std::vector<uint32_t> prove(TwoAdicFriPcs& pcs, const AirProgram& air, BfChallenger& challenger,
                            DeviceMatrix trace, const std::vector<uint32_t>& public_values);

// ------------------------------------------------------------------ prove, one proof over G GPUs
// Collectives the sharded prover needs, supplied by the host (torch.distributed over RCCL in
// tap-stark_amd/dist.py).  Buffers are device memory; the call is ordered on `stream`: the callee
// either enqueues on it or synchronises it, and on return later work on `stream` sees the result.
struct Comm {
    int rank = 0, world = 1;
    std::function<void(const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream)> all_gather;
    std::function<void(void* buf, size_t bytes, int root, hipStream_t stream)> broadcast;
};
struct ShardOptions {
    // FRI rounds stay sharded while a rank's slab holds at least 2^min_local_log values; the
    // remaining rounds run replicated on the gathered vector
    unsigned min_local_log = 12;
    // true: every rank already holds the WHOLE trace (e.g. generated on each device by
    // ts_trace_*): the one bulk exchange, the all-gather of the trace rows, is skipped
    bool trace_replicated = false;
    // true: every rank evaluates the quotient on its OWN cosets and derives its slab of the chunk
    // LDEs from that (sharded.cpp "local quotient"): no rank waits for the owner of the quotient
    // domain, no chunk broadcast.  Needs 2^log_blowup / G >= quotient degree (else the broadcast
    // path runs).  The same proof as ts_prove for EVERY trace: for one that violates its constraints
    // (which a release build of the reference proves without complaint, prover.rs:40-41) constraints /
    // Z_H is not a polynomial, the mixed chunks are not low-degree, FRI's final polynomial is not
    // constant (fri/src/prover.rs:129-134) -- and that sends every rank back through the broadcast
    // path (sharded.cpp; counted in Context::local_quotient_fallbacks).
    bool local_quotient = false;
};
// rank g owns the bit-reversed LDE rows [g N/G, (g+1) N/G) (whole cosets,
// G <= 2^log_blowup) of every committed matrix and the matching Merkle sub-trees, FRI slabs and
// queries.  `trace_rows`: natural rows [g n/G, (g+1) n/G) of the trace (or the whole trace with
// opt.trace_replicated).  Every rank passes a
// challenger in the same state and gets the whole proof, bit-identical to prove()'s.
std::vector<uint32_t> prove_sharded(TwoAdicFriPcs& pcs, const Comm& comm, const AirProgram& air,
                                    BfChallenger& challenger, DeviceMatrix trace_rows,
                                    const std::vector<uint32_t>& public_values, const ShardOptions& opt);

// ------------------------------------------------------------------ prove / verify over taptrees
// The reference's own configuration: `TapTreeMmcs` (basic/src/mmcs/taptree_mmcs.rs:24-119) as the
// MMCS of the PCS and of FRI (uni-stark/tests/fib_air.rs:117-131): every commitment is num_queries
// taptrees, the challenger observes all their roots, query q opens in tree q.  `TapLocks` is the flat
// table of lock scripts in commit order -- trace: Q (1 + w); quotient chunks: Q (1 + 4 qd); each of
// the log2(n) FRI rounds: Q (1 + 2) -- i.e. the order in which the reference's bit-commitment
// manager hands them out (tcs/mod.rs:251-260).  Proof = TSPF v2.
struct TapLocks {
    const uint8_t* bytes = nullptr;
    const uint64_t* offsets = nullptr;  // n_scripts + 1 entries
    size_t n_scripts = 0;
};
// comm != nullptr: the commitments of ONE proof split by tree over comm->world ranks, each holding the
// whole trace (tap_prover.cpp); every rank returns the whole proof, identical to the one-GPU one
std::vector<uint32_t> prove_tap(TwoAdicFriPcs& pcs, const AirProgram& air, BfChallenger& challenger,
                                DeviceMatrix trace, const std::vector<uint32_t>& public_values,
                                const TapLocks& locks, const Comm* comm = nullptr);
int verify_tap(const FriConfig& fri, const AirProgram& air, BfChallenger& challenger,
               const uint32_t* proof, size_t n_words, const std::vector<uint32_t>& public_values,
               const TapLocks& locks);
// verify_batch on words (taptree.cpp): leaf rebuilt from `locks[first .. first + n_evals]`, the index
// and the opened values; digests as 8 words = their bytes read little-endian
bool tap_verify_words(const TapLocks& locks, size_t first, uint32_t n_evals, uint32_t u32_size,
                      uint64_t index, const uint32_t* values, const uint32_t* path_words, size_t depth,
                      const uint32_t root_words[8]);

// ------------------------------------------------------------------ verify (host only)
// uni-stark/src/verifier.rs:19-161.  0 = accept; 1 InvalidProofShape, 2 InvalidOpeningArgument
// (FRI proof shape), 3 InvalidPowWitness, 4 input MMCS error, 5 commit-phase MMCS error,
// 6 FinalPolyMismatch, 7 OodEvaluationMismatch, 8 folded evaluation mismatch, 9 malformed buffer.
int verify(const FriConfig& fri, const AirProgram& air, BfChallenger& challenger,
           const uint32_t* proof, size_t n_words, const std::vector<uint32_t>& public_values);

// Pcs::verify (fri/src/two_adic_pcs.rs:421-534) for any rounds x matrices x points; same codes.
struct PcsMatClaim {
    unsigned log_height;                  // log2 of the LDE height = log_degree + log_blowup
    uint32_t width;
    std::vector<Ef> points;               // canonical EF4
    std::vector<std::vector<Ef>> values;  // values[p][column], the claimed p_column(points[p])
};
struct PcsRoundClaim {
    const uint32_t* root = nullptr;  // the round's commitment
    std::vector<PcsMatClaim> mats;
};
int pcs_verify(const FriConfig& fri, BfChallenger& challenger, const std::vector<PcsRoundClaim>& rounds,
               const uint32_t* fri_proof, size_t n_words);
// FRI alone (fri/src/verifier.rs:20-165) on a proof whose input proofs are the literal reduced
// openings (fri/tests/fri.rs:126-140)
int fri_verify_pass_through(const FriConfig& fri, BfChallenger& challenger, const uint32_t* fri_proof,
                            size_t n_words);

// ------------------------------------------------------------------ wire format (wire.cpp)
bool tspf_to_postcard(const uint32_t* words, size_t n_words, std::vector<uint8_t>& out);
bool postcard_to_tspf(const uint8_t* bytes, size_t n_bytes, std::vector<uint32_t>& out, int want_version = 0);

}  // namespace ts
