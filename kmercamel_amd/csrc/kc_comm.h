// Rank-to-rank exchange layer of the hash-sharded path (one process per GPU).  Not part of the ABI.
//
// The sharded pipeline (kc_shard.hip) needs exactly two collectives, neither of them a reduction over the data path:
//   * all-to-all(v) of device buffers (k-mer buckets, neighbour queries and their answers, chain-fragment pointers,
//     packed bases on their way to the rank that emits them): every rank sends a different byte range to every peer;
//   * all-gather of a few host words per rank (counts, flags).
// Back ends: RCCL (grouped ncclSend/ncclRecv on the context's stream -- over xGMI every peer pair has its own link, so
// the 7 sends of a rank run concurrently), host callbacks (the caller moves host bytes: torch.distributed/gloo in the
// tests, MPI elsewhere; device buffers are staged through pinned memory), and the trivial single-rank one.
#pragma once
#include "kc_internal.h"

#include <mutex>
#include <condition_variable>
// Ranks that are THREADS of one process (the CLI's --gpus): a rank posts where its outgoing bytes lie on its device, and after a
// barrier every rank copies what is meant for it straight out of its peers' buffers (device-to-device: peer DMA over xGMI when
// the ranks sit on different GPUs, a plain copy when they share one).
struct kc_thread_group {
    int world = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    u64 generation = 0;
    bool failed = false;
    struct Post { const void *d_send; const u64 *send_off, *send_cnt; int device; const u64 *words; } post[64];
    // false: a rank has failed on its own and left (abort()): nobody waits for it, every later exchange fails with KC_E_TRANSPORT
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const u64 gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen || failed; });
        return !failed;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        cv.notify_all();
    }
};

struct kc_comm {
    kc_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    int kind = 0;   // 0 self, 1 host callbacks, 2 RCCL, 3 threads of one process
    kc_thread_group *group = nullptr;
    // host callbacks
    kc_alltoallv_host_fn a2a_host = nullptr;
    kc_allgather_host_fn ag_host = nullptr;
    void *user = nullptr;
    u8 *stage_send = nullptr, *stage_recv = nullptr;   // pinned
    size_t stage_send_bytes = 0, stage_recv_bytes = 0;
    // RCCL
    void *nccl_comm = nullptr;
    u64 *d_small = nullptr;   // device scratch for the host-word all-gather
    // test knobs, read from the environment ONCE when the communicator is created (kcc_read_test_knobs), never on the data path:
    // KC_RCCL_SELF_SENDRECV=1: with RCCL, a rank's own share goes through grouped ncclSend / ncclRecv to itself too (exercises that
    // code on a box with one GPU); KC_TEST_FAIL_RANK=r: rank r fails on its own after the build (tests the release of its peers)
    bool test_self_sendrecv = false;
    int test_fail_rank = -1;
    // statistics
    u64 bytes_sent = 0, bytes_received = 0, calls = 0, self_sendrecv_calls = 0;
    double seconds = 0;
};

// All counts / offsets in BYTES, arrays of `world` entries (host).  Send ranges may overlap (an all-gather is an
// all-to-all whose send ranges coincide).  Enqueued on ctx->stream; returns after the data has arrived for the host
// back ends, asynchronously for RCCL (stream order).
int kcc_alltoallv(kc_comm *c, const void *d_send, const u64 *send_off, const u64 *send_cnt, void *d_recv, const u64 *recv_off, const u64 *recv_cnt);
// out[world * words] <- every rank's in[words] (host, synchronous)
int kcc_allgather_words(kc_comm *c, const u64 *in, u64 *out, int words);
// convenience: element-wise sum / max over ranks of a few host words
int kcc_allreduce_sum(kc_comm *c, u64 *vals, int words);
int kcc_allreduce_max(kc_comm *c, u64 *vals, int words);
// After a collective entry point (kc_compute_sharded*) returned `rc` on this rank.  An error that the ranks did NOT take together
// (the agreed ones end in "not part of the hash-sharded path" / "no k-mers" / "cannot be empty" / "(on some rank)") means the peers
// are still inside the flow and will wait for this rank in their next exchange: the thread transport wakes them with
// KC_E_TRANSPORT (kc_thread_group::abort).  RCCL and caller-supplied transports have no such channel -- as with any NCCL program
// the job is lost then and the launcher's watchdog has to end it; the failure points that can be CHECKED (capacity of the spill
// lists and recording pools) are therefore agreed among the ranks before anybody returns.
void kcc_after_collective_call(kc_comm *c, int rc);
