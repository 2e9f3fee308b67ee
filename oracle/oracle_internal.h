/* oracle/oracle_internal.h -- TEST INFRASTRUCTURE ONLY. Private declarations shared by the
 * oracle's translation units. */
#ifndef TS_ORACLE_INTERNAL_H
#define TS_ORACLE_INTERNAL_H
#include "oracle.h"

const uint32_t* ts_or_mmcs_matrix(const ts_or_mmcs_data* d, int i);
size_t ts_or_mmcs_height(const ts_or_mmcs_data* d, int i);
size_t ts_or_mmcs_width(const ts_or_mmcs_data* d, int i);
int ts_or_mmcs_n_mats(const ts_or_mmcs_data* d);

typedef struct {
    uint32_t width, n_public, n_nodes, n_constraints;
    const uint32_t* nodes;       /* n_nodes x 3 */
    const uint32_t* constraints; /* n_constraints */
} ts_or_tape;

int ts_or_tape_parse(const uint32_t* tape, size_t n_words, ts_or_tape* t);

/* evaluate every node over the base field; vals = scratch of n_nodes words */
void ts_or_tape_eval_base(const ts_or_tape* t, const uint32_t* local, const uint32_t* next,
                          const uint32_t* pis, uint32_t is_first, uint32_t is_last,
                          uint32_t is_transition, uint32_t* vals);
/* same over EF4 (verifier: uni-stark/src/folder.rs:24-32) */
void ts_or_tape_eval_ext(const ts_or_tape* t, const ef4* local, const ef4* next,
                         const uint32_t* pis, ef4 is_first, ef4 is_last, ef4 is_transition,
                         ef4* vals);

/* growable u32 buffer for proof assembly */
typedef struct {
    uint32_t* w;
    size_t len, cap;
    int overflow;
} ts_or_wbuf;

#endif
