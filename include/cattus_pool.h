/*
 * cattus_pool.h -- C ABI of libcattus_pool.so: pooling a self-play round's records over RCCL (xGMI) without Python.
 *
 * Self-play shards over one process per GPU (cattus_sp_config.first_game / game_stride, include/cattus_selfplay.h); the
 * reference pools a round by having every worker write its files into one directory (training/self-play/src/self_play.rs:60,273)
 * and tallies {player1_wins, player2_wins, draws} in one process (self_play.rs:65-69).  For a host in any language -- the Rust
 * self-player the binding in integration/rust targets (training/self-play/src/self_play_cmd.rs:55-153) -- this library does the
 * same across processes with two collectives: the fixed-size records are gathered on rank 0 (counts all-gathered first, then one
 * grouped send / receive of the zero-padded payloads), the counters are all-reduced.  The Python path (cattus_amd/dist.py over
 * torch.distributed) does the same and is what bench.py and scripts/selfplay_multi_gpu.py use.
 *
 * Every entry point returns 0 or a negative status; cattus_pool_last_error() gives the calling thread's last message.
 * Nothing here is on the evaluation path.
 *
 * Failure path: no entry point hangs.  The communicator is non-blocking (ncclCommInitRankConfig, blocking = 0) and every wait
 * inside an entry point -- for RCCL to finish what a call started, for the collective's kernels to leave the stream -- is a poll
 * with a deadline (cattus_pool_set_timeout; 600 s by default).  When a peer has died or never calls, the deadline passes:
 * the communicator is aborted (ncclCommAbort, which also ends its kernels on the device), the entry point returns
 * CATTUS_POOL_E_TIMEOUT, and the handle is dead: every later call on it returns CATTUS_POOL_E_STATE, cattus_pool_destroy frees it.
 * The caller's output arguments are written only by a call that returns 0.  What a host does then is the supervisor's recipe
 * (cattus_amd/supervisor.py): the shard's records are on disk already, the dead rank's games are re-queued, a new pool is made.
 * The reference leaves a dead worker undetected (training/self-play/src/self_play.rs:128, its own TODO).
 */
#ifndef CATTUS_POOL_H
#define CATTUS_POOL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CATTUS_POOL_ID_BYTES 128 /* == NCCL_UNIQUE_ID_BYTES */
#define CATTUS_POOL_E_STATE (-5)   /* the pool's communicator was aborted by an earlier failure */
#define CATTUS_POOL_E_TIMEOUT (-6) /* a wait passed its deadline: a peer is gone or never called; the communicator is aborted */

typedef struct cattus_pool cattus_pool;

/* Rank 0 makes the job's id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by whatever channel the host has
 * (a file in the round's directory, the launcher's environment, a socket). */
int cattus_pool_unique_id(uint8_t id[CATTUS_POOL_ID_BYTES]);

/* One communicator per process: rank `rank` of `world` on HIP device `device` (ncclCommInitRank; collective: every rank calls). */
int cattus_pool_create(const uint8_t id[CATTUS_POOL_ID_BYTES], int rank, int world, int device, cattus_pool** out);
void cattus_pool_destroy(cattus_pool* p);
/* Deadline, in seconds, of every wait inside the entry points of this pool (>= 0; 600 by default; cattus_pool_create itself
 * waits at most the default for the other ranks). */
int cattus_pool_set_timeout(cattus_pool* p, double seconds);
/* Diagnostic, for tests of the failure path: a collective that does not complete (the pool's stream is held while an all-reduce
 * waits behind it -- with one rank RCCL leaves nothing unmatched) returns what a collective returns when a peer never calls:
 * CATTUS_POOL_E_TIMEOUT after the pool's timeout, the communicator aborted, the stream drained. */
int cattus_pool_debug_stalled_collective(cattus_pool* p);

/* Collective.  bytes [n_local][record_bytes] and meta [n_local][3] = (game_idx, pos_idx, dir) as cattus_sp_result_records
 * returns them.  On rank 0: *all_bytes / *all_meta receive malloc'ed arrays of *n_total records sorted by (game, ply) -- free
 * them with cattus_pool_free --; on the other ranks they are set to NULL and *n_total to the job's total.  record_bytes must
 * be the same on every rank (a rank without records passes its game's record size all the same). */
int cattus_pool_records(cattus_pool* p, const uint8_t* bytes, const uint32_t* meta, uint64_t n_local, uint32_t record_bytes,
                        uint8_t** all_bytes, uint32_t** all_meta, uint64_t* n_total);

/* Collective: counters[i] = sum over ranks of counters[i] (win counters, positions, node_evals, cache hits / misses). */
int cattus_pool_reduce_counters(cattus_pool* p, uint64_t* counters, uint32_t n);

void cattus_pool_free(void* ptr);
const char* cattus_pool_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
