/*
 * skw_dist.h — C ABI of the ONE exchange step of the sharded Oneshot path (libskw_dist.so): an all-gather of fixed-size int32 token buffers over RCCL (xGMI).
 *
 * BASELINE.json configs[2] / SURVEY.md section 8(e): whole clips are sharded over the GPUs of a node (clip c -> rank c mod N, weights replicated, no data-path collective) and the
 * transcripts are gathered so that every rank holds every clip's tokens.  bench.py and the tests do that gather through torch.distributed (backend "nccl" = RCCL); this header is
 * the same collective for a host that is not Python — the reference's Rust server (crates/engine drives every plugin instance of a pipeline from ONE process:
 * /root/reference/crates/engine/src/oneshot.rs) binds these six calls, INTEGRATION.md section H shows the `extern "C"` block.  The reference itself has no counterpart: its
 * Oneshot batch path is one CPU node per request.
 *
 *   one process, N local devices (the server's shape)     skw_dist_create_local(devices, n)          -> ncclCommInitAll; a call drives all N ranks inside one RCCL group
 *   one process per GPU (bench.py's / torchrun's shape)   skw_dist_unique_id + skw_dist_create_rank  -> ncclGetUniqueId / ncclCommInitRank; the host carries the 128 id bytes to the ranks
 *
 * A row is SKW_DIST_ROW int32: [n_tokens, n_segments, token ids padded with -1] — streamkit_amd/dist.py pack_tokens.  Buffers are host memory; the library stages them through
 * device buffers of its own (a rank's gather is 64 x 226 x 4 B = 58 KB: the staging copies are noise beside the collective's latency).  Plain pointers and sizes only.
 */
#ifndef SKW_DIST_H
#define SKW_DIST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define SKW_DIST_ROW 226
#define SKW_DIST_ID_BYTES 128
typedef struct skw_dist skw_dist;

/* one process, n local devices: rank i of the group lives on devices[i] */
skw_dist* skw_dist_create_local(const int* devices, int n_devices, char* err, size_t errlen);
/* one process per rank: rank 0 fills `id`, the host hands the same bytes to every rank (file, socket, MPI ...), each rank then joins */
int       skw_dist_unique_id(unsigned char id[SKW_DIST_ID_BYTES], char* err, size_t errlen);
skw_dist* skw_dist_create_rank(const unsigned char id[SKW_DIST_ID_BYTES], int rank, int world, int device, char* err, size_t errlen);
int       skw_dist_world(const skw_dist*);         /* ranks in the group */
int       skw_dist_n_local(const skw_dist*);       /* ranks this handle drives: n_devices, or 1 */
/* Every rank contributes rows_per_rank rows.  send[i] / recv[i]: host buffers of local rank i — rows_per_rank x SKW_DIST_ROW and world x rows_per_rank x SKW_DIST_ROW int32;
 * afterwards recv[i] holds rank 0's rows, then rank 1's, ...  Blocking; 0 on success.  Every rank of the group must call it with the same rows_per_rank. */
int       skw_dist_all_gather_tokens(skw_dist*, const int32_t* const* send, int rows_per_rank, int32_t* const* recv);
const char* skw_dist_last_error(const skw_dist*);
void      skw_dist_free(skw_dist*);
#ifdef __cplusplus
}
#endif
#endif
