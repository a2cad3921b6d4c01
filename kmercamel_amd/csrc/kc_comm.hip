// Exchange layer of the hash-sharded path: see kc_comm.h.  RCCL is reached through dlopen, so that single-GPU users
// (the CLI measures process wall time) never load it.
#include "kc_comm.h"
#include <chrono>
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- RCCL entry points: the prototypes and types come from <rccl/rccl.h> (the library itself is dlopen'ed, not linked)
typedef ncclUniqueId NcclId;
static_assert(sizeof(NcclId) == 128, "kc_rccl_unique_id hands out 128 bytes");
typedef decltype(&ncclGetUniqueId) fn_get_id;
typedef decltype(&ncclCommInitRank) fn_init_rank;
typedef decltype(&ncclCommDestroy) fn_destroy;
typedef decltype(&ncclSend) fn_send;
typedef decltype(&ncclRecv) fn_recv;
typedef decltype(&ncclGroupStart) fn_group;
typedef decltype(&ncclAllGather) fn_allgather;
typedef decltype(&ncclGetErrorString) fn_errstr;
struct Rccl {
    void *lib = nullptr;
    fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_destroy destroy = nullptr;
    fn_send send = nullptr; fn_recv recv = nullptr; fn_group group_start = nullptr, group_end = nullptr;
    fn_allgather allgather = nullptr; fn_errstr errstr = nullptr;
    bool load(std::string &err) {
        if (lib) return true;
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
        get_id = (fn_get_id)dlsym(lib, "ncclGetUniqueId"); init_rank = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
        destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy"); send = (fn_send)dlsym(lib, "ncclSend"); recv = (fn_recv)dlsym(lib, "ncclRecv");
        group_start = (fn_group)dlsym(lib, "ncclGroupStart"); group_end = (fn_group)dlsym(lib, "ncclGroupEnd");
        allgather = (fn_allgather)dlsym(lib, "ncclAllGather"); errstr = (fn_errstr)dlsym(lib, "ncclGetErrorString");
        if (!get_id || !init_rank || !destroy || !send || !recv || !group_start || !group_end || !allgather) { err = "librccl.so lacks an expected symbol"; return false; }
        return true;
    }
};
Rccl g_rccl;
const ncclDataType_t kNcclUint8 = ncclUint8;

int nccl_fail(kc_ctx *ctx, const char *what, int rc) {
    return kc_fail(ctx, KC_E_TRANSPORT, std::string(what) + ": RCCL error " + std::to_string(rc) + (g_rccl.errstr ? std::string(" (") + g_rccl.errstr((ncclResult_t)rc) + ")" : ""));
}

int ensure_stage(kc_comm *c, size_t send_bytes, size_t recv_bytes) {
    kc_ctx *ctx = c->ctx;
    if (send_bytes > c->stage_send_bytes) {
        if (c->stage_send) (void)hipHostFree(c->stage_send);
        c->stage_send = nullptr; c->stage_send_bytes = 0;
        size_t want = send_bytes + send_bytes / 4 + 4096;
        KC_HIP(ctx, hipHostMalloc((void **)&c->stage_send, want, hipHostMallocDefault));
        c->stage_send_bytes = want;
    }
    if (recv_bytes > c->stage_recv_bytes) {
        if (c->stage_recv) (void)hipHostFree(c->stage_recv);
        c->stage_recv = nullptr; c->stage_recv_bytes = 0;
        size_t want = recv_bytes + recv_bytes / 4 + 4096;
        KC_HIP(ctx, hipHostMalloc((void **)&c->stage_recv, want, hipHostMallocDefault));
        c->stage_recv_bytes = want;
    }
    return KC_OK;
}

void read_test_knobs(kc_comm *c) {
    c->test_self_sendrecv = getenv("KC_RCCL_SELF_SENDRECV") != nullptr;
    if (const char *e = getenv("KC_TEST_FAIL_RANK")) c->test_fail_rank = atoi(e);
}

}  // namespace

int kcc_alltoallv(kc_comm *c, const void *d_send, const u64 *send_off, const u64 *send_cnt, void *d_recv, const u64 *recv_off, const u64 *recv_cnt) {
    kc_ctx *ctx = c->ctx;
    const double t0 = now_s();
    const int G = c->world;
    u64 tot_s = 0, tot_r = 0;
    for (int p = 0; p < G; ++p) { tot_s += send_cnt[p]; tot_r += recv_cnt[p]; }
    c->bytes_sent += tot_s - send_cnt[c->rank]; c->bytes_received += tot_r - recv_cnt[c->rank]; ++c->calls;
    int rc = KC_OK;
    if (c->kind == 0) {
        if (send_cnt[0] != recv_cnt[0]) return kc_fail(ctx, KC_E_ARG, "self exchange with unequal counts");
        if (send_cnt[0]) KC_HIP(ctx, hipMemcpyAsync((u8 *)d_recv + recv_off[0], (const u8 *)d_send + send_off[0], send_cnt[0], hipMemcpyDeviceToDevice, ctx->stream));
    } else if (c->kind == 1) {
        // device -> pinned (packed in rank order) -> caller's transport -> pinned -> device
        KC_TRY(ensure_stage(c, tot_s, tot_r));
        u64 o = 0;
        for (int p = 0; p < G; ++p) {
            if (send_cnt[p]) KC_HIP(ctx, hipMemcpyAsync(c->stage_send + o, (const u8 *)d_send + send_off[p], send_cnt[p], hipMemcpyDeviceToHost, ctx->stream));
            o += send_cnt[p];
        }
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (c->a2a_host(c->user, c->stage_send, send_cnt, c->stage_recv, recv_cnt) != 0) return kc_fail(ctx, KC_E_TRANSPORT, "the all-to-all callback failed");
        o = 0;
        for (int p = 0; p < G; ++p) {
            if (recv_cnt[p]) KC_HIP(ctx, hipMemcpyAsync((u8 *)d_recv + recv_off[p], c->stage_recv + o, recv_cnt[p], hipMemcpyHostToDevice, ctx->stream));
            o += recv_cnt[p];
        }
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else if (c->kind == 3) {
        kc_thread_group *g = c->group;
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));   // what this rank sends is complete before anybody reads it
        g->post[c->rank] = kc_thread_group::Post{d_send, send_off, send_cnt, ctx->device, nullptr};
        if (!g->barrier()) return kc_fail_agreed(ctx, KC_E_TRANSPORT, "thread transport: another rank has failed and left the exchange");
        for (int p = 0; p < G && rc == KC_OK; ++p) {
            const kc_thread_group::Post &q = g->post[p];
            const u64 n = q.send_cnt[c->rank];
            if (n != recv_cnt[p]) { rc = kc_fail(ctx, KC_E_ARG, "thread transport: send and receive counts disagree"); break; }
            if (!n) continue;
            const u8 *src = (const u8 *)q.d_send + q.send_off[c->rank];
            hipError_t e = q.device == ctx->device ? hipMemcpyAsync((u8 *)d_recv + recv_off[p], src, n, hipMemcpyDeviceToDevice, ctx->stream)
                                                   : hipMemcpyPeerAsync((u8 *)d_recv + recv_off[p], ctx->device, src, q.device, n, ctx->stream);
            if (e != hipSuccess) rc = kc_fail(ctx, KC_E_TRANSPORT, std::string("thread transport copy: ") + hipGetErrorString(e));
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == KC_OK) rc = kc_fail(ctx, KC_E_TRANSPORT, "thread transport: copy failed");
        if (!g->barrier() && rc == KC_OK) rc = kc_fail_agreed(ctx, KC_E_TRANSPORT, "thread transport: another rank has failed and left the exchange");   // (else: the senders may reuse their buffers)
    } else {
        // the part a rank keeps for itself is a device-to-device copy; the rest goes out as one group of sends and receives.
        // kc_comm::test_self_sendrecv (KC_RCCL_SELF_SENDRECV=1 when the communicator was created; tests): the rank's own part goes through ncclSend / ncclRecv to itself as well, so that the
        // grouped send / receive code -- prototypes, datatype, zero counts, stream order -- runs on a box with a single GPU.
        const bool self_sr = c->test_self_sendrecv;
        if (send_cnt[c->rank] != recv_cnt[c->rank]) return kc_fail(ctx, KC_E_ARG, "all-to-all: a rank's send and receive counts for itself differ");
        if (send_cnt[c->rank] && !self_sr)
            KC_HIP(ctx, hipMemcpyAsync((u8 *)d_recv + recv_off[c->rank], (const u8 *)d_send + send_off[c->rank], send_cnt[c->rank], hipMemcpyDeviceToDevice, ctx->stream));
        int e = g_rccl.group_start();
        if (e) return nccl_fail(ctx, "ncclGroupStart", e);
        for (int p = 0; p < G && !e; ++p) {
            if (p == c->rank && !self_sr) continue;
            if (send_cnt[p]) e = g_rccl.send((const u8 *)d_send + send_off[p], send_cnt[p], kNcclUint8, p, (ncclComm_t)c->nccl_comm, ctx->stream);
            if (!e && recv_cnt[p]) e = g_rccl.recv((u8 *)d_recv + recv_off[p], recv_cnt[p], kNcclUint8, p, (ncclComm_t)c->nccl_comm, ctx->stream);
        }
        int e2 = g_rccl.group_end();
        if (self_sr) ++c->self_sendrecv_calls;
        if (e || e2) rc = nccl_fail(ctx, "grouped ncclSend/ncclRecv", e ? e : e2);
    }
    c->seconds += now_s() - t0;
    return rc;
}

int kcc_allgather_words(kc_comm *c, const u64 *in, u64 *out, int words) {
    kc_ctx *ctx = c->ctx;
    const double t0 = now_s();
    ++c->calls;
    int rc = KC_OK;
    if (c->kind == 0) memcpy(out, in, (size_t)words * 8);
    else if (c->kind == 1) {
        if (c->ag_host(c->user, in, out, (u64)words * 8) != 0) rc = kc_fail(ctx, KC_E_TRANSPORT, "the all-gather callback failed");
    } else if (c->kind == 3) {
        kc_thread_group *g = c->group;
        g->post[c->rank].words = in;
        if (!g->barrier()) return kc_fail_agreed(ctx, KC_E_TRANSPORT, "thread transport: another rank has failed and left the exchange");
        for (int p = 0; p < c->world; ++p) memcpy(out + (size_t)p * words, g->post[p].words, (size_t)words * 8);
        if (!g->barrier()) rc = kc_fail_agreed(ctx, KC_E_TRANSPORT, "thread transport: another rank has failed and left the exchange");
    } else {
        // through a 64-word device window per rank; longer vectors go in pieces
        std::vector<u64> piece((size_t)64 * c->world);
        for (int w0 = 0; w0 < words; w0 += 64) {
            const int nw = words - w0 < 64 ? words - w0 : 64;
            u64 *d_in = c->d_small, *d_out = c->d_small + 64;
            KC_HIP(ctx, hipMemcpyAsync(d_in, in + w0, (size_t)nw * 8, hipMemcpyHostToDevice, ctx->stream));
            int e = g_rccl.allgather(d_in, d_out, (size_t)nw * 8, kNcclUint8, (ncclComm_t)c->nccl_comm, ctx->stream);
            if (e) return nccl_fail(ctx, "ncclAllGather", e);
            KC_HIP(ctx, hipMemcpyAsync(piece.data(), d_out, (size_t)nw * 8 * c->world, hipMemcpyDeviceToHost, ctx->stream));
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (int p = 0; p < c->world; ++p) memcpy(out + (size_t)p * words + w0, piece.data() + (size_t)p * nw, (size_t)nw * 8);
        }
    }
    c->seconds += now_s() - t0;
    return rc;
}

void kcc_after_collective_call(kc_comm *c, int rc) {
    if (rc == KC_OK || !c || c->kind != 3 || !c->group) return;
    if (c->ctx->err_agreed) return;   // every rank is returning this error at this point (kc_fail_agreed): nobody is left waiting
    c->group->abort();
}

int kcc_allreduce_sum(kc_comm *c, u64 *vals, int words) {
    std::vector<u64> all((size_t)words * c->world);
    KC_TRY(kcc_allgather_words(c, vals, all.data(), words));
    for (int w = 0; w < words; ++w) { u64 s = 0; for (int p = 0; p < c->world; ++p) s += all[(size_t)p * words + w]; vals[w] = s; }
    return KC_OK;
}
int kcc_allreduce_max(kc_comm *c, u64 *vals, int words) {
    std::vector<u64> all((size_t)words * c->world);
    KC_TRY(kcc_allgather_words(c, vals, all.data(), words));
    for (int w = 0; w < words; ++w) { u64 s = 0; for (int p = 0; p < c->world; ++p) s = all[(size_t)p * words + w] > s ? all[(size_t)p * words + w] : s; vals[w] = s; }
    return KC_OK;
}

extern "C" int kc_comm_create_self(kc_ctx *ctx, kc_comm **out) {
    if (!ctx || !out) return KC_E_ARG;
    kc_comm *c = new kc_comm();
    c->ctx = ctx; c->rank = 0; c->world = 1; c->kind = 0;
    read_test_knobs(c);
    *out = c;
    return KC_OK;
}

extern "C" int kc_comm_create_callbacks(kc_ctx *ctx, int rank, int world, kc_alltoallv_host_fn alltoallv, kc_allgather_host_fn allgather, void *user,
                                        kc_comm **out) {
    if (!ctx || !out || !alltoallv || !allgather || world < 1 || rank < 0 || rank >= world) return KC_E_ARG;
    if (world > 64) return kc_fail(ctx, KC_E_ARG, "at most 64 ranks");
    kc_comm *c = new kc_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->kind = 1;
    read_test_knobs(c);
    c->a2a_host = alltoallv; c->ag_host = allgather; c->user = user;
    *out = c;
    return KC_OK;
}

extern "C" int kc_thread_group_create(int world, kc_thread_group **out) {
    if (!out || world < 1 || world > 64) return KC_E_ARG;
    kc_thread_group *g = new kc_thread_group();
    g->world = world;
    *out = g;
    return KC_OK;
}
extern "C" void kc_thread_group_destroy(kc_thread_group *g) { delete g; }
extern "C" int kc_comm_create_threads(kc_ctx *ctx, kc_thread_group *group, int rank, kc_comm **out) {
    if (!ctx || !group || !out || rank < 0 || rank >= group->world) return KC_E_ARG;
    kc_comm *c = new kc_comm();
    c->ctx = ctx; c->rank = rank; c->world = group->world; c->kind = 3; c->group = group;
    read_test_knobs(c);
    // ranks on different GPUs read each other's buffers: peer access both ways (already enabled is fine)
    KC_HIP(ctx, hipSetDevice(ctx->device));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess)
        for (int d = 0; d < ndev; ++d)
            if (d != ctx->device) { int can = 0; if (hipDeviceCanAccessPeer(&can, ctx->device, d) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d, 0); (void)hipGetLastError(); }
    *out = c;
    return KC_OK;
}

extern "C" int kc_rccl_unique_id(void *id128) {
    if (!id128) return KC_E_ARG;
    std::string err;
    if (!g_rccl.load(err)) return KC_E_NO_DEVICE;
    NcclId id;
    if (g_rccl.get_id(&id) != 0) return KC_E_NO_DEVICE;
    memcpy(id128, id.internal, 128);
    return KC_OK;
}

extern "C" int kc_comm_create_rccl(kc_ctx *ctx, int rank, int world, const void *id128, kc_comm **out) {
    if (!ctx || !out || !id128 || world < 1 || rank < 0 || rank >= world) return KC_E_ARG;
    if (world > 64) return kc_fail(ctx, KC_E_ARG, "at most 64 ranks");
    std::string err;
    if (!g_rccl.load(err)) return kc_fail(ctx, KC_E_NO_DEVICE, err);
    KC_HIP(ctx, hipSetDevice(ctx->device));
    NcclId id;
    memcpy(id.internal, id128, 128);
    ncclComm_t comm = nullptr;
    int e = g_rccl.init_rank(&comm, world, id, rank);
    if (e) return nccl_fail(ctx, "ncclCommInitRank", e);
    kc_comm *c = new kc_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->kind = 2; c->nccl_comm = comm;
    read_test_knobs(c);
    if (kc_alloc(ctx, &c->d_small, 64 + 64 * 64) != KC_OK) { (void)g_rccl.destroy(comm); delete c; return KC_E_NO_DEVICE; }
    *out = c;
    return KC_OK;
}

extern "C" void kc_comm_destroy(kc_comm *c) {
    if (!c) return;
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->stage_send) (void)hipHostFree(c->stage_send);
    if (c->stage_recv) (void)hipHostFree(c->stage_recv);
    if (c->d_small) kc_release(c->ctx, c->d_small);
    if (c->nccl_comm) (void)g_rccl.destroy((ncclComm_t)c->nccl_comm);
    delete c;
}

extern "C" int kc_comm_stats(kc_comm *c, uint64_t *stats) {
    if (!c || !stats) return KC_E_ARG;
    for (int i = 0; i < 8; ++i) stats[i] = 0;
    stats[0] = c->bytes_sent; stats[1] = c->bytes_received; stats[2] = c->calls; stats[3] = (u64)(c->seconds * 1e6);
    stats[4] = c->self_sendrecv_calls;   // exchanges whose own part went through ncclSend / ncclRecv (KC_RCCL_SELF_SENDRECV, tests)
    return KC_OK;
}
