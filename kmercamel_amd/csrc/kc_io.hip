// Streamed host <-> device I/O of the CLI path: the reference reads records through kseq's 16 KiB buffer
// (kseq.h:183-234, parser.h:106-118) and writes the superstring through an ostream; here
//   * kc_read_fasta_device: a reader thread fills pinned chunks from the file (plain, gz or stdin, OpenFile parser.h:88-101),
//     every chunk is copied to the device and TOKENISED THERE -- header lines and line breaks are cut out, records are
//     separated by one '\n' -- straight into the joined sequence kc_compute takes; reading, copying and tokenising overlap;
//   * kc_result_write_fd: the result leaves through pinned chunks, the device-to-host copy of chunk i+1 overlapping the
//     write() of chunk i.
// FASTA tokeniser semantics (kseq.h:183-224): a line that starts with '>' or '@' is a header; every other line is sequence;
// the line break (and a '\r' before it) is dropped, so k-mers span the lines of a record but never two records.  FASTQ
// input (a line starting with '+') is reported back and left to the host parser.
#include "kc_internal.h"
#include "kc_scan_dev.h"
#include <zlib.h>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <atomic>
#include <vector>
#include <functional>
#include <algorithm>

namespace {

// What a run of bytes does to the tokeniser: does it contain a line start, is the last line start a header, and how many
// bytes does it emit if it begins inside a header line (c1) or not (c0).  Runs compose associatively.
struct TokFn { u32 has, hdr; u64 c0, c1; };
__device__ __forceinline__ TokFn tok_compose(const TokFn &a, const TokFn &b) {   // a, then b
    TokFn r;
    r.has = a.has | b.has;
    r.hdr = b.has ? b.hdr : a.hdr;
    r.c0 = a.c0 + ((a.has ? a.hdr : 0u) ? b.c1 : b.c0);
    r.c1 = a.c1 + ((a.has ? a.hdr : 1u) ? b.c1 : b.c0);
    return r;
}
__device__ __forceinline__ TokFn tok_shfl_up(const TokFn &v, int d) {
    TokFn r;
    r.has = __shfl_up(v.has, d, 64); r.hdr = __shfl_up(v.hdr, d, 64); r.c0 = __shfl_up(v.c0, d, 64); r.c1 = __shfl_up(v.c1, d, 64);
    return r;
}
// inclusive scan of TokFn over a 256-thread workgroup; *excl = composition of all earlier threads (identity for thread 0), *total = all
__device__ __forceinline__ void tok_block_scan(const TokFn &mine, TokFn *excl, TokFn *total) {
    __shared__ TokFn wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    TokFn inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const TokFn o = tok_shfl_up(inc, d);
        if (lane >= d) inc = tok_compose(o, inc);
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    TokFn pre{0, 0, 0, 0}, tot{0, 0, 0, 0};
    bool first = true;
    for (int i = 0; i < 4; ++i) {
        if (i < wv) pre = tok_compose(pre, wsum[i]);
        tot = first ? wsum[i] : tok_compose(tot, wsum[i]);
        first = false;
    }
    __syncthreads();
    // exclusive within the wave
    TokFn prev = tok_shfl_up(inc, 1);
    if (lane == 0) prev = TokFn{0, 0, 0, 0};
    *excl = tok_compose(pre, prev);
    *total = tot;
}

// masks of one thread's 64 raw bytes (raw[-1] is the byte before the chunk, '\n' at the start of the input)
struct TokMasks { u64 ls, hs, nl, crnl, plus; };
__device__ __forceinline__ TokMasks tok_masks(const u8 *__restrict__ raw, u64 base, u64 n) {
    TokMasks m{0, 0, 0, 0, 0};
    u8 prev = raw[(long long)base - 1];
    for (int i = 0; i < 64; ++i) {
        if (base + i >= n) break;
        const u8 c = raw[base + i];
        const bool ls = prev == '\n';
        if (ls) {
            m.ls |= (u64)1 << i;
            if (c == '>' || c == '@') m.hs |= (u64)1 << i;
            if (c == '+') m.plus |= (u64)1 << i;
        }
        if (c == '\n') m.nl |= (u64)1 << i;
        if (c == '\r' && base + i + 1 < n && raw[base + i + 1] == '\n') m.crnl |= (u64)1 << i;
        prev = c;
    }
    return m;
}
// header bits of the 64 bytes given the state at their first byte; emitted = sequence bytes + one separator per header start
__device__ __forceinline__ u64 tok_header_mask(const TokMasks &m, u32 in_hdr) {
    u64 h = 0, ls = m.ls;
    u32 state = in_hdr;
    int pos = 0;
    while (true) {
        const int nxt = ls ? __builtin_ctzll(ls) : 64;
        if (state && nxt > pos) h |= (nxt == 64 ? ~(u64)0 : (((u64)1 << nxt) - 1)) & ~(((u64)1 << pos) - 1);
        if (nxt == 64) break;
        state = (m.hs >> nxt) & 1;
        pos = nxt;
        ls &= ls - 1;
    }
    return h;
}
__device__ __forceinline__ TokFn tok_fn(const TokMasks &m, u64 valid) {
    TokFn f;
    f.has = m.ls != 0;
    f.hdr = f.has ? (u32)((m.hs >> (63 - __builtin_clzll(m.ls))) & 1) : 0;
    const u64 drop = m.nl | m.crnl;
    const u64 h0 = tok_header_mask(m, 0), h1 = tok_header_mask(m, 1);
    f.c0 = (u64)__builtin_popcountll(((~h0 & ~drop) | m.hs) & valid);
    f.c1 = (u64)__builtin_popcountll(((~h1 & ~drop) | m.hs) & valid);
    return f;
}
__global__ void __launch_bounds__(256) tok_summary_kernel(const u8 *__restrict__ raw, u64 n, TokFn *blocks, u64 *flags) {
    const u64 base = ((u64)blockIdx.x * 256 + threadIdx.x) * 64;
    TokFn f{0, 0, 0, 0};
    if (base < n) {
        const TokMasks m = tok_masks(raw, base, n);
        const u64 valid = n - base >= 64 ? ~(u64)0 : (((u64)1 << (n - base)) - 1);
        f = tok_fn(m, valid);
        if (m.plus) flags[0] = 1;   // a '+' line: FASTQ
    }
    TokFn ex, tot;
    tok_block_scan(f, &ex, &tot);
    if (threadIdx.x == 0) blocks[blockIdx.x] = tot;
}
// one workgroup over the workgroup summaries of a chunk (a few thousand): state and output offset at every workgroup's start;
// carry = {in_header, out_offset} across chunks
__global__ void __launch_bounds__(256) tok_chain_kernel(const TokFn *__restrict__ blocks, u64 nb, u64 *carry, u64 *bstate, u64 *boff) {
    const u64 per = (nb + 255) / 256, b0 = (u64)threadIdx.x * per, b1 = b0 + per < nb ? b0 + per : nb;
    TokFn f{0, 0, 0, 0};
    for (u64 b = b0; b < b1; ++b) f = tok_compose(f, blocks[b]);
    TokFn ex, tot;
    tok_block_scan(f, &ex, &tot);
    const u32 st0 = (u32)carry[0];
    const u64 off0 = carry[1];
    u32 st = ex.has ? ex.hdr : st0;
    u64 off = off0 + (st0 ? ex.c1 : ex.c0);
    for (u64 b = b0; b < b1; ++b) {
        bstate[b] = st; boff[b] = off;
        const TokFn g = blocks[b];
        off += st ? g.c1 : g.c0;
        st = g.has ? g.hdr : st;
    }
    __syncthreads();
    if (threadIdx.x == 0) { carry[0] = tot.has ? tot.hdr : st0; carry[1] = off0 + (st0 ? tot.c1 : tot.c0); }
}
__global__ void __launch_bounds__(256) tok_write_kernel(const u8 *__restrict__ raw, u64 n, const u64 *__restrict__ bstate, const u64 *__restrict__ boff, u8 *out, u64 out_cap,
                                                        u64 *flags) {
    const u64 base = ((u64)blockIdx.x * 256 + threadIdx.x) * 64;
    TokFn f{0, 0, 0, 0};
    TokMasks m{0, 0, 0, 0, 0};
    u64 valid = 0;
    if (base < n) {
        m = tok_masks(raw, base, n);
        valid = n - base >= 64 ? ~(u64)0 : (((u64)1 << (n - base)) - 1);
        f = tok_fn(m, valid);
    }
    TokFn ex, tot;
    tok_block_scan(f, &ex, &tot);
    if (base >= n) return;
    const u32 st0 = (u32)bstate[blockIdx.x];
    const u32 st = ex.has ? ex.hdr : st0;
    u64 pos = boff[blockIdx.x] + (st0 ? ex.c1 : ex.c0);
    const u64 h = tok_header_mask(m, st);
    u64 emit = ((~h & ~(m.nl | m.crnl)) | m.hs) & valid;
    if (pos + (u64)__builtin_popcountll(emit) > out_cap) { flags[1] = 1; return; }
    while (emit) {
        const int i = __builtin_ctzll(emit);
        emit &= emit - 1;
        out[pos++] = ((m.hs >> i) & 1) ? (u8)'\n' : raw[base + i];
    }
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool io_trace() { static const bool t = getenv("KC_CLI_TRACE") != nullptr; return t; }

// copy threads per pinned chunk (reads from the page cache into it, copies out of it): KC_IO_THREADS, default 8 (measured on the
// 256-core MI355X box, 2.6 GB from tmpfs through 32 MB pinned chunks to the device: 4 threads 34 GB/s, 8 threads 44 GB/s, 16 threads 45 GB/s
// against 57 GB/s for the bare H2D copy; tools/io_probe.hip)
int kc_io_threads() {
    static const int t = [] { const char *e = getenv("KC_IO_THREADS"); const int v = e ? atoi(e) : 8; return v >= 1 && v <= 64 ? v : 8; }();
    return getenv("KC_IO_SERIAL") ? 1 : t;
}

// A few threads that stay for the whole transfer: run(f) executes f(0..T-1), the caller taking part as thread 0, and returns when all
// are through (a chunk is 1-2 ms of copying: starting threads per chunk showed in the totals).
struct Workers {
    int T;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const std::function<void(int)> *job = nullptr;
    u64 gen = 0;
    int pending = 0;
    bool quit = false;
    explicit Workers(int t) : T(t < 1 ? 1 : t) {
        for (int i = 1; i < T; ++i) th.emplace_back([this, i] {
            u64 seen = 0;
            for (;;) {
                const std::function<void(int)> *j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_go.wait(lk, [&] { return quit || gen != seen; });
                    if (quit) return;
                    seen = gen; j = job;
                }
                (*j)(i);
                { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cv_done.notify_one(); }
            }
        });
    }
    void run(const std::function<void(int)> &f) {
        if (T == 1) { f(0); return; }
        { std::lock_guard<std::mutex> lk(mu); job = &f; pending = T - 1; ++gen; }
        cv_go.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~Workers() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
};

// ---- reader: plain files by pread(), gzip (and anything zlib passes through) by gzread, "-" = stdin
struct Reader {
    int fd = -1;
    gzFile gz = nullptr;
    off_t pos = 0, size = -1;   // regular files: chunks are read by a few threads at once (pread on disjoint ranges)
    bool open(const std::string &path, std::string &err) {
        if (path == "-") { gz = gzdopen(0, "r"); return gz != nullptr; }
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) { err = "couldn't open file " + path; return false; }
        unsigned char magic[2] = {0, 0};
        const ssize_t got = pread(fd, magic, 2, 0);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) { gz = gzdopen(fd, "r"); fd = -1; if (gz) gzbuffer(gz, 1 << 20); return gz != nullptr; }
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) size = st.st_size;
        return true;
    }
    bool parallel() const { return !gz && size >= 0; }
    long read_some(u8 *dst, size_t cap, Workers *wk) {
        if (gz) return gzread(gz, dst, (unsigned)cap);
        if (size >= 0) {
            // page-cache / tmpfs reads are memory copies in the kernel: one thread moves ~5 GB/s, eight move the chunk at ~44 GB/s
            const size_t want = (size_t)std::min<off_t>((off_t)cap, size > pos ? size - pos : 0);
            const int T = (wk && want >= ((size_t)8 << 20)) ? wk->T : 1;
            std::atomic<bool> bad{false};
            const std::function<void(int)> part = [&](int t) {
                if (t >= T) return;
                size_t lo = want * (size_t)t / (size_t)T, hi = want * (size_t)(t + 1) / (size_t)T;
                while (lo < hi) {
                    const ssize_t r = pread(fd, dst + lo, hi - lo, pos + (off_t)lo);
                    if (r <= 0) { bad = true; return; }   // (a file that shrinks under us is an error)
                    lo += (size_t)r;
                }
            };
            if (T > 1) wk->run(part); else part(0);
            if (bad) return -1;
            pos += (off_t)want;
            return (long)want;
        }
        size_t have = 0;
        while (have < cap) {
            const ssize_t r = ::read(fd, dst + have, cap - have);
            if (r < 0) return -1;
            if (r == 0) break;
            have += (size_t)r;
        }
        return (long)have;
    }
    void close() { if (gz) gzclose(gz); else if (fd >= 0) ::close(fd); gz = nullptr; fd = -1; }
};

int ensure_copy_stream(kc_ctx *ctx) {
    if (!ctx->copy_stream) KC_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    return KC_OK;
}

}  // namespace

// ---- pinned chunks of the context
void *kc_pin_acquire(kc_ctx *ctx, size_t bytes) {
    for (auto &b : ctx->pinned) if (!b.busy && b.bytes >= bytes) { b.busy = true; return b.p; }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    ctx->pinned.push_back(kc_ctx::Pin{p, bytes, true});
    return p;
}
void kc_pin_release(kc_ctx *ctx, void *p) {
    for (auto &b : ctx->pinned) if (b.p == p) b.busy = false;
}
void kc_pin_free_all(kc_ctx *ctx) {
    for (auto &b : ctx->pinned) (void)hipHostFree(b.p);
    ctx->pinned.clear();
}

// *d_seq (kc_alloc'd; release with kc_device_free) = the record sequences joined by '\n', *len bytes.
// *is_fastq = 1: the input is FASTQ (or could not be handled here) -- nothing is returned, use the host parser.
// Pipeline per 32 MB chunk: [reader thread + its helpers] pread into a pinned chunk -> [copy stream] H2D -> [context stream] the three
// tokeniser kernels; the copy of chunk i+1 runs under the kernels of chunk i, the read of chunk i+2 under both.
static int read_fasta_device_impl(kc_ctx *ctx, const char *path, u64 range_off, u64 range_len, bool ranged, void **d_seq, uint64_t *len, int *is_fastq) {
    if (!ctx || !path || !d_seq || !len || !is_fastq) return KC_E_ARG;
    *d_seq = nullptr; *len = 0; *is_fastq = 0;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    const double t_begin = now_s();
    Reader rd;
    std::string err;
    if (!rd.open(path, err)) return kc_fail(ctx, KC_E_IO, err.empty() ? std::string("couldn't open file ") + path : err);
    if (ranged) {
        // a byte range of a plain regular file (a rank's share of the input: kc_read_fasta_device_range)
        if (!rd.parallel()) { rd.close(); return kc_fail(ctx, KC_E_ARG, "a byte range can only be read from a plain regular file"); }
        const off_t end = (off_t)(range_off + range_len) < rd.size ? (off_t)(range_off + range_len) : rd.size;
        rd.pos = (off_t)range_off < end ? (off_t)range_off : end;
        rd.size = end;
    }
    // (sized BEFORE the reader thread exists: that thread advances rd.pos, and the main thread must not look at rd.pos / rd.size again)
    const u64 range_bytes = ranged ? (u64)(rd.size - rd.pos) : 0;
    const size_t CH = (size_t)32 << 20;
    const int NB = 3;
    u8 *hbuf[NB] = {nullptr, nullptr, nullptr};
    hipEvent_t done[NB], copied[NB];
    bool have_ev[NB] = {false, false, false};
    u8 *d_raw = nullptr, *out = nullptr;
    TokFn *blocks = nullptr;
    u64 *aux = nullptr;   // [0..1] carry, [2..3] flags, then bstate / boff
    const u64 nbmax = (CH + 16383) / 16384 + 1;
    int rc = ensure_copy_stream(ctx);
    // (only the FIRST pinned chunk before the reader starts: a fresh process pays ~5 ms per 32 MB of hipHostMalloc, and the reader can fill
    // chunk 0 while chunks 1 and 2 -- and the device buffers -- are being obtained; round 6)
    auto get_chunk = [&](int i) -> int {
        u8 *p = (u8 *)kc_pin_acquire(ctx, CH + 64);
        if (!p) return KC_E_NO_DEVICE;   // (no message from here: the helper thread below must not write the context's error string)
        hbuf[i] = p;
        if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&copied[i], hipEventDisableTiming) != hipSuccess) return KC_E_NO_DEVICE;
        have_ev[i] = true;
        return KC_OK;
    };
    if (rc == KC_OK && get_chunk(0) != KC_OK) rc = kc_fail(ctx, KC_E_NO_DEVICE, "pinned buffer allocation failed");
    // reader thread: fills free pinned buffers, hands them over in order (started before the device buffers are allocated: the first
    // chunks are read while hipMalloc and the first kernel launch -- which loads the code object -- run)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<int, long>> ready;   // (buffer, bytes); bytes <= 0: end / error
    bool free_buf[NB] = {true, false, false};   // (chunks 1 and 2 become free once they exist, below)
    bool stop = false;
    std::thread reader;
    if (rc == KC_OK) {
        reader = std::thread([&] {
            Workers wk(rd.parallel() ? kc_io_threads() : 1);
            u8 held = 0; bool have_held = false;   // a '\r' that ended a chunk waits for the next one ("\r\n" must stay together)
            for (int b = 0;; b = (b + 1) % NB) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return free_buf[b] || stop; });
                    if (stop) return;
                    free_buf[b] = false;
                }
                size_t off = 0;
                if (have_held) { hbuf[b][0] = held; off = 1; have_held = false; }
                long got = rd.read_some(hbuf[b] + off, CH - off, &wk);
                long total = got < 0 ? -1 : (long)off + got;
                if (got > 0 && (size_t)got == CH - off && hbuf[b][total - 1] == '\r') { held = '\r'; have_held = true; --total; }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ready.emplace_back(b, total);
                }
                cv.notify_all();
                if (got <= 0 || (size_t)got < CH - off) {
                    // the last chunk: followed by the end marker unless the chunk itself says "end" (0 bytes) or "error" (< 0).  (A chunk
                    // that holds only the '\r' kept back from a full chunk before EOF has 1 byte and got == 0: it needs the marker too.)
                    if (total > 0) { std::lock_guard<std::mutex> lk(mu); ready.emplace_back(-1, 0); cv.notify_all(); }
                    return;
                }
            }
        });
    }
    // (chunks 1 and 2 while the reader fills chunk 0.  Pinning them from a helper thread, so that this one could go on to the device buffers,
    // was measured and dropped: the two threads' HIP calls serialise, the input arrived 20 - 40 ms later)
    for (int i = 1; i < NB && rc == KC_OK; ++i) {
        if (get_chunk(i) != KC_OK) { rc = kc_fail(ctx, KC_E_NO_DEVICE, "pinned buffer allocation failed"); break; }
        { std::lock_guard<std::mutex> lk(mu); free_buf[i] = true; }
        cv.notify_all();
    }
    if (rc == KC_OK) rc = kc_alloc(ctx, &d_raw, NB * (CH + 256));
    if (rc == KC_OK) rc = kc_alloc(ctx, &blocks, nbmax);
    if (rc == KC_OK) rc = kc_alloc(ctx, &aux, 4 + 2 * nbmax);
    u64 out_cap = 0;
    {   // plain files: the joined sequence is never longer than the file; streams: grow as needed
        struct stat st;
        if (rc == KC_OK && ranged) out_cap = range_bytes + 64;
        else if (rc == KC_OK && !rd.gz && fstat(rd.fd, &st) == 0 && st.st_size > 0) out_cap = (u64)st.st_size + 64;
        else out_cap = (u64)256 << 20;
    }
    if (rc == KC_OK) rc = kc_alloc(ctx, &out, out_cap);
    if (rc == KC_OK && hipMemsetAsync(aux, 0, 32, ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
    const double t_setup = now_s();
    double t_first = 0;
    u64 total_in = 0;
    bool first = true, fastq = false;
    u8 last_byte = (u8)'\n';        // the byte before the next chunk ('\n': the input starts at a line start)
    std::deque<int> inflight;       // buffers whose copy / kernels are still running, oldest first
    auto give_back = [&](int b) { { std::lock_guard<std::mutex> lk(mu); free_buf[b] = true; } cv.notify_all(); };
    // (the launches sit in a lambda: KC_LAUNCH returns from its caller on a refused grid, and this function has a reader thread to stop first)
    auto tokenise = [&](const u8 *dr, u64 n, u64 nb) -> int {
        KC_LAUNCH(tok_summary_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dr, n, blocks, aux + 2);
        KC_LAUNCH(tok_chain_kernel, dim3(1), dim3(256), 0, ctx->stream, (const TokFn *)blocks, nb, aux, aux + 4, aux + 4 + nbmax);
        KC_LAUNCH(tok_write_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dr, n, (const u64 *)(aux + 4), (const u64 *)(aux + 4 + nbmax), out, out_cap, aux + 2);
        return KC_OK;
    };
    while (rc == KC_OK) {
        while (!inflight.empty() && hipEventQuery(done[inflight.front()]) == hipSuccess) { give_back(inflight.front()); inflight.pop_front(); }
        std::pair<int, long> item;
        {
            std::unique_lock<std::mutex> lk(mu);
            if (ready.empty() && !inflight.empty()) {   // the reader may be waiting for the oldest buffer: finish that one first
                lk.unlock();
                (void)hipEventSynchronize(done[inflight.front()]);
                give_back(inflight.front());
                inflight.pop_front();
                continue;
            }
            cv.wait(lk, [&] { return !ready.empty(); });
            item = ready.front();
            ready.pop_front();
        }
        if (item.second < 0) { rc = kc_fail(ctx, KC_E_IO, std::string("read error on ") + path); break; }
        if (item.first < 0 || item.second == 0) {
            if (item.first >= 0) give_back(item.first);
            break;
        }
        const int b = item.first;
        u64 n = (u64)item.second;
        const u8 *src = hbuf[b];
        if (first) {
            // everything before the first header character is skipped (kseq.h:190-193); an '@' header means FASTQ-style input
            u64 p = 0;
            while (p < n && src[p] != '>' && src[p] != '@') ++p;
            if (p == n) { give_back(b); continue; }
            if (src[p] == '@') { fastq = true; break; }
            src += p; n -= p;
            first = false;
        }
        // grow the output of a stream of unknown length
        if (total_in + n + 64 > out_cap) {
            u64 hcarry[2] = {0, 0};
            if (hipMemcpyAsync(hcarry, aux, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = KC_E_NO_DEVICE; break; }
            u8 *bigger = nullptr;
            const u64 ncap = out_cap * 2 + n;
            rc = kc_alloc(ctx, &bigger, ncap);
            if (rc != KC_OK) break;
            if (hipMemcpyAsync(bigger, out, hcarry[1], hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = KC_E_NO_DEVICE; break; }
            kc_release(ctx, out);
            out = bigger; out_cap = ncap;
        }
        u8 *dr = d_raw + (size_t)b * (CH + 256) + 128;   // dr[-1] = the byte before this chunk
        u8 *stash = hbuf[b] + CH + 2;                    // (pinned, and not refilled before this chunk is through)
        stash[0] = last_byte;
        // (d_raw's piece b is free: buffer b came back from the reader only after done[b], the end of its previous kernels)
        if (hipMemcpyAsync(dr - 1, stash, 1, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess || hipMemcpyAsync(dr, src, n, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess ||
            hipEventRecord(copied[b], ctx->copy_stream) != hipSuccess || hipStreamWaitEvent(ctx->stream, copied[b], 0) != hipSuccess) {
            rc = kc_fail(ctx, KC_E_NO_DEVICE, "H2D copy failed");
            break;
        }
        last_byte = src[n - 1];
        const u64 nb = (n + 16383) / 16384;
        rc = tokenise((const u8 *)dr, n, nb);
        if (rc != KC_OK) break;
        if (hipGetLastError() != hipSuccess || hipEventRecord(done[b], ctx->stream) != hipSuccess) { rc = kc_fail(ctx, KC_E_NO_DEVICE, "tokeniser launch failed"); break; }
        if (total_in == 0 && io_trace()) { (void)hipEventSynchronize(done[b]); t_first = now_s(); }
        total_in += n;
        inflight.push_back(b);
    }
    u64 h[4] = {0, 0, 0, 0};
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->stream);
    { std::lock_guard<std::mutex> lk(mu); stop = true; for (int i = 0; i < NB; ++i) free_buf[i] = true; }
    cv.notify_all();
    if (reader.joinable()) reader.join();
    if (rc == KC_OK && !fastq) {
        if (hipMemcpyAsync(h, aux, 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
        if (rc == KC_OK && h[3]) rc = kc_fail(ctx, KC_E_NO_DEVICE, "FASTA tokeniser: output buffer too small");
        if (rc == KC_OK && h[2]) fastq = true;
    }
    rd.close();
    for (int i = 0; i < NB; ++i) { if (hbuf[i]) kc_pin_release(ctx, hbuf[i]); if (have_ev[i]) { (void)hipEventDestroy(done[i]); (void)hipEventDestroy(copied[i]); } }
    kc_release(ctx, d_raw); kc_release(ctx, blocks); kc_release(ctx, aux);
    if (io_trace())
        fprintf(stderr, "[io] read: setup (pinned chunks, device buffers) %.1f ms, first chunk tokenised after %.1f ms more, %.2f GB in %.1f ms in all\n", (t_setup - t_begin) * 1e3,
                t_first > 0 ? (t_first - t_setup) * 1e3 : 0.0, total_in / 1e9, (now_s() - t_begin) * 1e3);
    if (rc != KC_OK || fastq) { kc_release(ctx, out); *is_fastq = fastq ? 1 : 0; return rc; }
    *d_seq = out; *len = h[1];
    return KC_OK;
}

extern "C" int kc_read_fasta_device(kc_ctx *ctx, const char *path, void **d_seq, uint64_t *len, int *is_fastq) {
    return read_fasta_device_impl(ctx, path, 0, 0, false, d_seq, len, is_fastq);
}
// The bytes [offset, offset + nbytes) of a plain regular file (clipped to its end): the share of the input one rank of `compute --gpus N`
// reads for itself, all ranks at once.  The range should begin at a header line -- what lies before the first '>' in it is skipped, as at
// the start of a file (kseq.h:190-193).
extern "C" int kc_read_fasta_device_range(kc_ctx *ctx, const char *path, uint64_t offset, uint64_t nbytes, void **d_seq, uint64_t *len, int *is_fastq) {
    return read_fasta_device_impl(ctx, path, offset, nbytes, true, d_seq, len, is_fastq);
}

extern "C" void kc_device_free(kc_ctx *ctx, void *d_ptr) {
    if (!ctx || !d_ptr) return;
    (void)hipStreamSynchronize(ctx->stream);
    kc_release(ctx, d_ptr);
}

namespace {
// result -> host, chunk by chunk through the context's pinned chunks: the device-to-host copy of the next chunks runs while `sink`
// disposes of the current one (sink(chunk, bytes, offset in the string) -> false: stop with KC_E_IO)
int drain_result(kc_ctx *ctx, const char *src, u64 total, const std::function<bool(const char *, u64, u64)> &sink) {
    const size_t CH = (size_t)32 << 20;
    const int NB = 3;
    char *hb[NB] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[NB];
    bool have_ev[NB] = {false, false, false};
    int rc = KC_OK;
    for (int i = 0; i < NB && rc == KC_OK; ++i) {
        hb[i] = (char *)kc_pin_acquire(ctx, CH + 64);
        if (!hb[i]) rc = kc_fail(ctx, KC_E_NO_DEVICE, "pinned buffer allocation failed");
        if (rc == KC_OK && hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) rc = KC_E_NO_DEVICE;
        else if (rc == KC_OK) have_ev[i] = true;
    }
    u64 sent = 0, done = 0;
    u64 len_in[NB] = {0, 0, 0};
    u64 nsent = 0, ndone = 0;   // chunks
    auto send_next = [&]() -> bool {
        const int s = (int)(nsent % NB);
        len_in[s] = total - sent < CH ? total - sent : CH;
        if (hipMemcpyAsync(hb[s], src + sent, len_in[s], hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipEventRecord(ev[s], ctx->stream) != hipSuccess) return false;
        sent += len_in[s]; ++nsent;
        return true;
    };
    while (rc == KC_OK && done < total) {
        while (sent < total && nsent - ndone < (u64)NB) if (!send_next()) { rc = KC_E_NO_DEVICE; break; }
        if (rc != KC_OK) break;
        const int s = (int)(ndone % NB);
        if (hipEventSynchronize(ev[s]) != hipSuccess) { rc = KC_E_NO_DEVICE; break; }
        if (!sink(hb[s], len_in[s], done)) { rc = kc_fail(ctx, KC_E_IO, "write failed"); break; }
        done += len_in[s]; ++ndone;
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < NB; ++i) { if (hb[i]) kc_pin_release(ctx, hb[i]); if (have_ev[i]) (void)hipEventDestroy(ev[i]); }
    return rc;
}
}  // namespace

// result -> file descriptor.  (Measured on the MI355X box, 2.5 GB to tmpfs: one sequential write() stream moves ~5 GB/s with page
// allocation and ~7 GB/s into allocated pages, and more writers do not help -- buffered writes to one file queue on the inode lock.  The
// CLI therefore goes through kc_result_write_mem and a mapping whenever the output is a regular file; this is the path for pipes, stdout
// and everything unusual.)
extern "C" int kc_result_write_fd(kc_result *r, int fd, int which) {
    if (!r || fd < 0) return KC_E_ARG;
    kc_ctx *ctx = r->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    const char *src = which ? r->d_mask : r->d_out;
    if (!src) return kc_fail(ctx, KC_E_ARG, "the result holds no such string");
    return drain_result(ctx, src, r->slice_len, [&](const char *buf, u64 n, u64) {
        u64 off = 0;
        while (off < n) {
            const ssize_t w = ::write(fd, buf + off, n - off);
            if (w < 0) return false;
            off += (u64)w;
        }
        return true;
    });
}

// Destination = host memory: every chunk is copied on by eight threads while the next ones arrive.  For a pageable destination this
// beats one big hipMemcpy (which stages through a single thread's copies), and it is what the CLI uses to fill the shared mapping of
// its output file -- stores into a mapping take no inode lock, so they scale where write() does not (tmpfs, fallocate'd but untouched
// pages: 27 GB/s with eight threads against 7 GB/s for write(); tools/io_probe2.cpp).
extern "C" int kc_result_write_mem(kc_result *r, void *dst_v, int which) {
    if (!r || !dst_v) return KC_E_ARG;
    kc_ctx *ctx = r->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    const char *src = which ? r->d_mask : r->d_out;
    if (!src) return kc_fail(ctx, KC_E_ARG, "the result holds no such string");
    char *dst = static_cast<char *>(dst_v);
    Workers wk(r->slice_len >= ((u64)8 << 20) ? kc_io_threads() : 1);
    const double t0 = now_s();
    const int rc = drain_result(ctx, src, r->slice_len, [&](const char *buf, u64 n, u64 at) {
        const int T = n >= ((u64)4 << 20) ? wk.T : 1;
        const std::function<void(int)> part = [&](int t) {
            if (t >= T) return;
            const u64 lo = n * (u64)t / (u64)T, hi = n * (u64)(t + 1) / (u64)T;
            memcpy(dst + at + lo, buf + lo, hi - lo);
        };
        if (T > 1) wk.run(part); else part(0);
        return true;
    });
    if (io_trace()) fprintf(stderr, "[io] result -> host memory: %.2f GB in %.1f ms\n", r->slice_len / 1e9, (now_s() - t0) * 1e3);
    return rc;
}
