/* kmercamel_amd -- C ABI of the MI355X-native BIGREEDY hot path (libkmercamel_amd.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types, no exceptions.  Every entry point names
 * the reference interface it stands in for (file:line relative to the KmerCamel v2.3.0 tree).  The C++ headers in
 * kmercamel_amd/src/ (global.h, simplitigs.h, parser.h) re-create the reference's template signatures on top of
 * these calls; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * All functions return 0 on success or a negative KC_E_* code; kc_last_error() gives the message of the last
 * failure on that context.  All functions fail with KC_E_NO_DEVICE when no HIP device is present -- there is no
 * CPU fallback behind this ABI.
 *
 * Keys: a k-mer is 2 bits/nucleotide (A=0,C=1,G=2,T=3), first nucleotide most significant (kmers.h:15-44), held in
 * W = 1/2/4 little-endian 64-bit words for k<32 / k<64 / k<=127 (main.cpp:309-315).
 */
#ifndef KMERCAMEL_AMD_H
#define KMERCAMEL_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define KC_OK 0
#define KC_E_NO_DEVICE (-1)   /* no HIP device / HIP runtime error */
#define KC_E_EMPTY (-2)       /* "input cannot be empty" (global.h:219-221) */
#define KC_E_INVALID_CHAR (-3)/* non-ACGT where the reference asserts / throws (simplitigs.h:82, masks.h:75-77) */
#define KC_E_ARG (-4)         /* bad argument (k out of 1..127, null pointer, ...) */
#define KC_E_CAPACITY (-5)    /* hash set capacity exhausted (kh_put's ret=-1, khash.h:314-321) */
#define KC_E_IO (-6)          /* "couldn't open file" (parser.h:95-97) */
#define KC_E_NO_KMERS (-7)    /* input holds no k-mers (main.cpp:155-158) */
#define KC_E_TRANSPORT (-8)   /* an exchange of the hash-sharded path failed (RCCL error, transport callback, peer copy) */
#define KC_E_UNSUPPORTED_SHARDED (-9) /* a valid request that the hash-sharded path does not serve (every rank returns it together, nothing has been
                                         exchanged): run it through kc_compute on one GPU -- the CLI does that by itself */

typedef struct kc_ctx kc_ctx;
typedef struct kc_set kc_set;
typedef struct kc_simplitigs kc_simplitigs;
typedef struct kc_result kc_result;

/* ---- context: one per GPU (one host thread / process per GPU); owns a private HIP stream and a memory pool ---- */
int kc_ctx_create(int device, kc_ctx **out);
void kc_ctx_destroy(kc_ctx *ctx);
const char *kc_last_error(const kc_ctx *ctx);
/* the private HIP stream (hipStream_t) the kernels are launched on */
void *kc_ctx_stream(kc_ctx *ctx);
int kc_ctx_synchronize(kc_ctx *ctx);
/* stats[4]: peak bytes held by the context's device memory pool, bytes obtained from hipMalloc, hipMalloc calls,
 * microseconds spent inside them */
int kc_ctx_stats(kc_ctx *ctx, uint64_t *stats);

/* ---- k-mer set: replaces kh_S64/S128/S256 + kmer_dictNN_t (khash_utils.h:28-95) ---- */
/* wrapper.kh_init_set() (khash_utils.h:43). capacity_hint = expected number of k-mer occurrences to be added (0 = grow on demand). */
int kc_set_create(kc_ctx *ctx, int k, int complements, uint64_t capacity_hint, kc_set **out);
void kc_set_destroy(kc_set *set); /* wrapper.kh_destroy_set (khash_utils.h:55) */
/* AddKMers (parser.h:22-49): all k-mers of one record; non-ACGT bytes break k-mers; `case_sensitive` keeps a k-mer
 * only if its first base is upper case.  Separate records by any non-ACGT byte (e.g. '\n') to add many at once. */
int kc_set_add_sequence(kc_set *set, const char *seq, uint64_t len, int case_sensitive);
/* same, sequence bytes already resident in device memory */
int kc_set_add_sequence_device(kc_set *set, const void *d_seq, uint64_t len, int case_sensitive);
/* -z support (ReadKMersFiltered / AddKMersWithFrequencies, parser.h:51-85,122-141): call kc_set_enable_counting on an empty set,
 * add sequences as usual (every k-mer occurrence is counted), then kc_set_filter_min_frequency keeps the k-mers seen
 * at least `min_frequency` (1..255) times and turns counting off again. */
int kc_set_enable_counting(kc_set *set);
int kc_set_filter_min_frequency(kc_set *set, int min_frequency);
/* kh_size (khash.h:543) */
int kc_set_size(kc_set *set, uint64_t *n);
/* kMersToVec (khash_utils.h:131-141): all keys, arbitrary order, W words each; `out` holds kc_set_size()*W words */
int kc_set_export(kc_set *set, uint64_t *out_words);
/* kh_get != kh_end for n keys of W words each (khash.h:232-246); exact-key membership (no RC lookup) */
int kc_set_contains(kc_set *set, const uint64_t *key_words, uint64_t n, uint8_t *out);

/* kh_del (khash.h:353-359) for n exact keys (W words each): *n_erased (may be NULL) = how many were present.  The table has
 * no tombstones (the simplitig phase consumes k-mers through port links instead); erasing rebuilds it without the keys,
 * O(table) per call -- it exists for the element-wise compatibility API (eraseKMer, next_simplitig, khash_utils.h:105-118). */
int kc_set_erase(kc_set *set, const uint64_t *key_words, uint64_t n, uint64_t *n_erased);

/* ---- simplitigs: replaces std::vector<simplitig_t> (simplitigs.h:11) as a packed, device-resident collection ---- */
/* get_simplitigs (simplitigs.h:191-205): greedy simplitigs of the set by de Bruijn neighbour lookup. The reference
 * consumes the set; here the set stays intact (it is reused for the -M mask). Order/orientation follow the
 * canonical rule, not the khash bucket order. */
int kc_simplitigs_build(kc_set *set, kc_simplitigs **out);
/* simplitigs_from_fasta / simplitig_from_string (simplitigs.h:78-103): n records, ACGT only (else KC_E_INVALID_CHAR),
 * record i = seqs[offsets[i] .. offsets[i+1]) , record order kept. */
int kc_simplitigs_from_strings(kc_ctx *ctx, int k, const char *seqs, const uint64_t *offsets, uint64_t n, kc_simplitigs **out);
void kc_simplitigs_destroy(kc_simplitigs *s);
int kc_simplitigs_count(kc_simplitigs *s, uint64_t *n, uint64_t *total_bases);
/* ASCII export: `seqs` holds total_bases bytes, `offsets` n+1 entries */
int kc_simplitigs_export(kc_simplitigs *s, char *seqs, uint64_t *offsets);
/* simplitigs_to_kmer_vec + PartialPreSort (simplitigs.h:53-65, global_sparse.h:14-35): the sparse branch's vertex
 * list as 1-k-mer simplitigs */
int kc_simplitigs_to_sparse(kc_simplitigs *s, kc_simplitigs **out);

/* ---- global greedy ---- */
/* OverlapHamiltonianPath (global.h:43-133) / OverlapHamiltonianPathSparse (global_sparse.h:43-132).
 * edge_from: V = n*(1+complements) entries (UINT64_MAX = none), overlaps: V bytes (0xFF = none); host buffers. */
int kc_overlap_path(kc_simplitigs *s, int complements, int lower_bound, uint64_t *edge_from, uint8_t *overlaps);
/* SuperstringFromPath (global.h:149-210). `set` (may be NULL) enables the max-one mask output (the -M stream,
 * global.h:185-195); it must hold exactly the k-mers of `s` (fill_kmers, simplitigs.h:41-51).
 * *out / *mask_out are malloc'd NUL-terminated strings (free with kc_free); *length = returned size_t. */
int kc_superstring_from_path(kc_simplitigs *s, const uint64_t *edge_from, const uint8_t *overlaps, int complements,
                             kc_set *set, char **out, char **mask_out, uint64_t *length);
/* Global (global.h:217-226) = path + superstring, device-resident in between. want_mask builds the k-mer set of `s`
 * internally when `set` is NULL. */
int kc_global(kc_simplitigs *s, int complements, kc_set *set, int want_mask, char **out, char **mask_out, uint64_t *length);
/* LowerBoundLength (lower_bound.h:7-21) */
int kc_lower_bound(kc_simplitigs *s, int complements, uint64_t *length);
void kc_free(void *p);

/* ---- fused device-resident pipeline: the greedy branch of kmercamel<>() (main.cpp:122-212) ---- */
#define KC_FLAG_COMPLEMENTS 1u       /* bidirectional (absence = -u) */
#define KC_FLAG_SIMPLITIGS_INPUT 2u  /* -S: records are simplitigs */
#define KC_FLAG_WANT_MASK 4u         /* -M: also produce the max-one mask */
#define KC_FLAG_INPUT_ON_DEVICE 8u   /* seq points to device memory */
#define KC_FLAG_MIN_FREQUENCY(z) (((uint32_t)(z) & 0xffu) << 8) /* -z INT (1..255; 0/1 = no filter; not with -S) */
/* seq: record sequences separated by one '\n' (any non-ACGT byte separates k-mers).  -S: the records are the simplitigs -- either
 * packed back to back with `offsets` (host, n+1), or, with KC_FLAG_INPUT_ON_DEVICE and offsets == NULL, separated by '\n' as
 * kc_read_fasta_device leaves them (simplitigs_from_fasta, simplitigs.h:78-103: one record per simplitig).  The result stays on the
 * device until copied out. */
int kc_compute(kc_ctx *ctx, const void *seq, uint64_t len, const uint64_t *offsets, uint64_t n_records, int k, uint32_t flags,
               kc_result **out);
/* stats[16]: 0 n_kmers, 1 n_simplitigs, 2 length, 3 sparse_branch, 4 n_vertices, 5 matching_rounds,
 *            6.. device time per phase in microseconds: 6 pack+insert, 7 simplitigs, 8 overlap path, 9 emit(+mask),
 *            10 neighbour-probe kernel time (us, HIP events around the launch of kc_probe_block_kernel -- the dominant
 *            kernel; kc_choose_dense_kernel for k < 18), 11 k-mer insert kernel time (us, summed over its launches), 12 table capacity,
 *            13 overlap repair rounds, 14 total device us, 15 reserved */
int kc_result_stats(kc_result *r, uint64_t *stats);
/* the same followed by more: stats[16] cycles the simplitig rule had to cut (every k-mer of such a cycle has both ports
 * matched; the reference's delete-as-you-go walk cannot close a cycle, so this bounds how many more simplitigs than the
 * reference there can be), [17] k-mers held by this rank, [18] neighbour queries this rank sent (sharded runs); n <= 32 */
int kc_result_stats_ext(kc_result *r, uint64_t *stats, int n);
/* device pointers to the output characters (length bytes each; mask NULL unless requested) */
int kc_result_device_ptrs(kc_result *r, void **d_out, void **d_mask);
/* copy to host buffers of `length` bytes (mask may be NULL) */
int kc_result_copy(kc_result *r, char *out, char *mask_out);
void kc_result_destroy(kc_result *r);

/* ---- hash-sharded multi-GPU path (the reference is one thread of one process, main.cpp:122-212, so these
 * have no reference counterpart): one process (or thread) per GPU, every rank holds a kc_ctx and a kc_comm.  The k-mer
 * set is partitioned by table-block range (block = hash of the canonical minimizer), the ranks exchange k-mer buckets,
 * neighbour queries and chain fragments, and together emit ONE masked superstring, byte-identical to kc_compute's on
 * the union of the inputs; rank r ends up with the r-th contiguous slice of it. ---- */
typedef struct kc_comm kc_comm;
/* single rank (world = 1): the sharded code path without a transport */
int kc_comm_create_self(kc_ctx *ctx, kc_comm **out);
/* Transport supplied by the caller on HOST buffers (tests: torch.distributed/gloo; any MPI).  all-to-all(v): `send` holds
 * the bytes for rank 0, 1, ... back to back (send_bytes[p] each), `recv` receives rank 0's, 1's, ... bytes back to back.
 * all-gather: `out` = every rank's `bytes_per_rank` bytes in rank order.  Return 0 on success. */
typedef int (*kc_alltoallv_host_fn)(void *user, const void *send, const uint64_t *send_bytes, void *recv, const uint64_t *recv_bytes);
typedef int (*kc_allgather_host_fn)(void *user, const void *in, void *out, uint64_t bytes_per_rank);
int kc_comm_create_callbacks(kc_ctx *ctx, int rank, int world, kc_alltoallv_host_fn alltoallv, kc_allgather_host_fn allgather, void *user,
                             kc_comm **out);
/* RCCL over xGMI: rank 0 calls kc_rccl_unique_id (128 bytes), hands the id to the other ranks by any means (bench.py:
 * torch.distributed broadcast), then every rank calls kc_comm_create_rccl.  librccl.so is loaded on first use. */
int kc_rccl_unique_id(void *id128);
int kc_comm_create_rccl(kc_ctx *ctx, int rank, int world, const void *id128, kc_comm **out);
/* Ranks that are threads of ONE process (kmercamel compute --gpus N): create the group once, then one communicator per rank
 * thread; exchanges are device-to-device copies out of the peers' buffers (peer DMA between GPUs, plain copies when ranks
 * share a GPU).  Destroy the communicators first, then the group. */
typedef struct kc_thread_group kc_thread_group;
int kc_thread_group_create(int world, kc_thread_group **out);
void kc_thread_group_destroy(kc_thread_group *group);
int kc_comm_create_threads(kc_ctx *ctx, kc_thread_group *group, int rank, kc_comm **out);
void kc_comm_destroy(kc_comm *comm);
/* stats[8]: 0 bytes sent, 1 bytes received, 2 collective calls, 3 microseconds spent inside them (host clock), 4 exchanges in which this
 * rank's own share went through ncclSend / ncclRecv (KC_RCCL_SELF_SENDRECV=1: the RCCL code path with one rank) */
int kc_comm_stats(kc_comm *comm, uint64_t *stats);
/* The partition itself, host side only (no GPU needed; the multi-rank CPU tests exercise it): geometry of the table for
 * `total_occurrences` k-mer occurrences over `world` ranks -- out6 = {table blocks, log2 slots per block, blocks per rank,
 * first global slot of `rank`, slots per rank, minimizer length} -- and the owner rank of n canonical k-mers (W words each). */
int kc_shard_geometry(int k, uint64_t total_occurrences, int world, int rank, uint64_t *out6);
int kc_shard_owner_of(int k, uint64_t total_occurrences, int world, const uint64_t *keys, uint64_t n, uint8_t *owner);
/* The greedy branch of kmercamel<>() over all ranks' records: `seq`/`len` are THIS rank's records (joined by '\n', may be
 * empty), collective call.  Supported flags: KC_FLAG_COMPLEMENTS, KC_FLAG_INPUT_ON_DEVICE, KC_FLAG_WANT_MASK, KC_FLAG_MIN_FREQUENCY
 * (-S input: kc_compute_sharded_simplitigs).  What the hash-sharded flow does not serve itself -- k < 18 (plain hashing: no
 * minimizer blocks to shard by), the sparse branch (main.cpp:175-181, found out by all ranks together after phase 1), -z with
 * overflowing table blocks, -S with -M -- stays inside the collective: the ranks' records go to rank 0, which runs kc_compute, and
 * every rank receives its slice [L r / G, L (r + 1) / G) of the one result (no speed-up for these, but the same contract).  With
 * KC_SHARD_NO_ROOT_FALLBACK=1 in the environment such a request comes back as KC_E_UNSUPPORTED_SHARDED from ALL ranks together, before
 * anything is exchanged, and the caller runs kc_compute on one GPU itself (the CLI does).  The result holds this rank's slice; stats
 * as kc_result_stats with global counts. */
int kc_compute_sharded(kc_ctx *ctx, kc_comm *comm, const void *seq, uint64_t len, int k, uint32_t flags, kc_result **out);
/* -S over all ranks (simplitigs_from_fasta, simplitigs.h:89-103; main.cpp:150-160): THIS rank's simplitig records one after the other
 * (no separators) with their offsets[0..n_records], as kc_compute takes them; the ranks' records in rank order are the simplitig
 * list.  Flags: KC_FLAG_COMPLEMENTS, KC_FLAG_INPUT_ON_DEVICE (-M with -S goes through the single-GPU kc_compute).  Collective. */
int kc_compute_sharded_simplitigs(kc_ctx *ctx, kc_comm *comm, const void *seq, uint64_t len, const uint64_t *offsets, uint64_t n_records, int k,
                                  uint32_t flags, kc_result **out);
/* position and size of this rank's slice inside the superstring (offset 0 / whole length for kc_compute results);
 * kc_result_device_ptrs / kc_result_copy address the slice */
int kc_result_slice(kc_result *r, uint64_t *offset, uint64_t *slice_len);

/* ---- streamed I/O of the CLI path (ReadKMers' kseq loop, parser.h:88-118 + kseq.h:183-234; the output stream of
 * main.cpp:164-210): a reader thread feeds pinned chunks, the device tokenises the FASTA (header lines and line breaks cut
 * out, records joined by '\n') while the next chunk is read; the result leaves through pinned chunks, copy overlapping write. */
/* path: plain / gzip file, or "-" for stdin.  *d_seq: device buffer with the joined record sequences (free with
 * kc_device_free), ready for kc_compute(..., KC_FLAG_INPUT_ON_DEVICE).  *is_fastq = 1 (and nothing returned) when the input
 * is FASTQ: use the host parser.  KC_E_IO when the file cannot be opened (parser.h:95-97). */
int kc_read_fasta_device(kc_ctx *ctx, const char *path, void **d_seq, uint64_t *len, int *is_fastq);
/* the same for the bytes [offset, offset + nbytes) of a plain regular file: the share of the input that one rank of the multi-GPU
 * `compute` reads for itself (the reference reads one stream in one thread, parser.h:88-118); the range should begin at a header line */
int kc_read_fasta_device_range(kc_ctx *ctx, const char *path, uint64_t offset, uint64_t nbytes, void **d_seq, uint64_t *len, int *is_fastq);
void kc_device_free(kc_ctx *ctx, void *d_ptr);
/* write the result's characters (which = 0) or its -M mask (which = 1) to a file descriptor */
int kc_result_write_fd(kc_result *r, int fd, int which);
/* the same into host memory (e.g. a shared mapping of the output file whose pages fallocate() has provided): pinned chunks, each
 * moved on by several threads while the next one arrives from the device.  dst must hold the slice (kc_result_slice). */
int kc_result_write_mem(kc_result *r, void *dst, int which);

/* ---- next row 1: maskopt -t max-one / min-one (masks.h:40-78, 239-261) ---- */
/* ms: one masked superstring (ACGTacgt, else KC_E_INVALID_CHAR after output like the reference). out: len bytes. */
int kc_maskopt_ones(kc_ctx *ctx, const char *ms, uint64_t len, int k, int complements, int minimize, char *out);
/* the same with the masked superstring on the device (e.g. from kc_read_fasta_device: header and line breaks already cut out) and the
 * result left there, to leave through kc_result_write_fd / kc_result_write_mem / kc_result_copy -- `maskopt` on multi-gigabyte files
 * (masks.h:239-261 reads and writes through iostreams).  KC_E_ARG when the sequence holds a record separator ('\n': more than one
 * FASTA record, masks.h:258); KC_E_INVALID_CHAR is returned WITH a result (masks.h:75-77 prints the string, then reports). */
int kc_maskopt_device(kc_ctx *ctx, const void *d_ms, uint64_t len, int k, int complements, int minimize, kc_result **out);

/* number of HIP devices visible to the process (0 when there is none) */
int kc_device_count(void);
/* library/ABI version string */
const char *kc_version(void);

#ifdef __cplusplus
}
#endif
#endif
