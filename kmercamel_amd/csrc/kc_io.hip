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

// ---- reader: plain files by read(), gzip (and anything zlib passes through) by gzread, "-" = stdin
// copy threads per pinned chunk (reads from the page cache into it, copies out of it): KC_IO_THREADS, default 4
static int kc_io_threads() {
    static const int t = [] { const char *e = getenv("KC_IO_THREADS"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 64 ? v : 4; }();
    return t;
}
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
    long read_some(u8 *dst, size_t cap) {
        if (gz) return gzread(gz, dst, (unsigned)cap);
        if (size >= 0 && !getenv("KC_IO_SERIAL")) {
            // page-cache / tmpfs reads are memory copies in the kernel: one thread moves ~5 GB/s, four move the chunk in a quarter of the time
            const size_t want = (size_t)std::min<off_t>((off_t)cap, size > pos ? size - pos : 0);
            const int T = want >= ((size_t)8 << 20) ? kc_io_threads() : 1;
            std::atomic<bool> bad{false};
            auto part = [&](int t) {
                size_t lo = want * (size_t)t / (size_t)T, hi = want * (size_t)(t + 1) / (size_t)T;
                while (lo < hi) {
                    const ssize_t r = pread(fd, dst + lo, hi - lo, pos + (off_t)lo);
                    if (r <= 0) { bad = true; return; }   // (a file that shrinks under us is an error)
                    lo += (size_t)r;
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t) th.emplace_back(part, t);
            part(0);
            for (auto &x : th) x.join();
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

}  // namespace

// *d_seq (kc_alloc'd; release with kc_device_free) = the record sequences joined by '\n', *len bytes.
// *is_fastq = 1: the input is FASTQ (or could not be handled here) -- nothing is returned, use the host parser.
extern "C" int kc_read_fasta_device(kc_ctx *ctx, const char *path, void **d_seq, uint64_t *len, int *is_fastq) {
    if (!ctx || !path || !d_seq || !len || !is_fastq) return KC_E_ARG;
    *d_seq = nullptr; *len = 0; *is_fastq = 0;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    Reader rd;
    std::string err;
    if (!rd.open(path, err)) return kc_fail(ctx, KC_E_IO, err.empty() ? std::string("couldn't open file ") + path : err);
    const size_t CH = (size_t)32 << 20;
    const int NB = 3;
    u8 *hbuf[NB] = {nullptr, nullptr, nullptr};
    hipEvent_t done[NB];
    u8 *d_raw = nullptr, *out = nullptr;
    TokFn *blocks = nullptr;
    u64 *aux = nullptr;   // [0..1] carry, [2..3] flags, then bstate / boff
    const u64 nbmax = (CH + 16383) / 16384 + 1;
    int rc = KC_OK;
    for (int i = 0; i < NB && rc == KC_OK; ++i) {
        if (hipHostMalloc((void **)&hbuf[i], CH + 64, hipHostMallocDefault) != hipSuccess) rc = kc_fail(ctx, KC_E_NO_DEVICE, "pinned buffer allocation failed");
        if (rc == KC_OK && hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) rc = KC_E_NO_DEVICE;
    }
    if (rc == KC_OK) rc = kc_alloc(ctx, &d_raw, NB * (CH + 256));
    if (rc == KC_OK) rc = kc_alloc(ctx, &blocks, nbmax);
    if (rc == KC_OK) rc = kc_alloc(ctx, &aux, 4 + 2 * nbmax);
    u64 out_cap = 0;
    {   // plain files: the joined sequence is never longer than the file; streams: grow as needed
        struct stat st;
        if (rc == KC_OK && !rd.gz && fstat(rd.fd, &st) == 0 && st.st_size > 0) out_cap = (u64)st.st_size + 64;
        else out_cap = (u64)256 << 20;
    }
    if (rc == KC_OK) rc = kc_alloc(ctx, &out, out_cap);
    if (rc == KC_OK && hipMemsetAsync(aux, 0, 32, ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
    // reader thread: fills free pinned buffers, hands them over in order
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<int, long>> ready;   // (buffer, bytes); bytes <= 0: end / error
    bool free_buf[NB] = {true, true, true};
    bool stop = false;
    std::thread reader;
    if (rc == KC_OK) {
        reader = std::thread([&] {
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
                long got = rd.read_some(hbuf[b] + off, CH - off);
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
    u64 total_in = 0;
    bool first = true, fastq = false;
    u8 last_byte = (u8)'\n';        // the byte before the next chunk ('\n': the input starts at a line start)
    std::deque<int> inflight;       // buffers whose copy / kernels are still running, oldest first
    auto give_back = [&](int b) { { std::lock_guard<std::mutex> lk(mu); free_buf[b] = true; } cv.notify_all(); };
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
        if (hipMemcpyAsync(dr - 1, stash, 1, hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipMemcpyAsync(dr, src, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            rc = kc_fail(ctx, KC_E_NO_DEVICE, "H2D copy failed");
            break;
        }
        last_byte = src[n - 1];
        const u64 nb = (n + 16383) / 16384;
        KC_LAUNCH(tok_summary_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const u8 *)dr, n, blocks, aux + 2);
        KC_LAUNCH(tok_chain_kernel, dim3(1), dim3(256), 0, ctx->stream, (const TokFn *)blocks, nb, aux, aux + 4, aux + 4 + nbmax);
        KC_LAUNCH(tok_write_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const u8 *)dr, n, (const u64 *)(aux + 4), (const u64 *)(aux + 4 + nbmax), out, out_cap,
                           aux + 2);
        if (hipGetLastError() != hipSuccess || hipEventRecord(done[b], ctx->stream) != hipSuccess) { rc = kc_fail(ctx, KC_E_NO_DEVICE, "tokeniser launch failed"); break; }
        total_in += n;
        inflight.push_back(b);
    }
    u64 h[4] = {0, 0, 0, 0};
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
    for (int i = 0; i < NB; ++i) { if (hbuf[i]) (void)hipHostFree(hbuf[i]); (void)hipEventDestroy(done[i]); }
    kc_release(ctx, d_raw); kc_release(ctx, blocks); kc_release(ctx, aux);
    if (rc != KC_OK || fastq) { kc_release(ctx, out); *is_fastq = fastq ? 1 : 0; return rc; }
    *d_seq = out; *len = h[1];
    return KC_OK;
}

extern "C" void kc_device_free(kc_ctx *ctx, void *d_ptr) {
    if (!ctx || !d_ptr) return;
    (void)hipStreamSynchronize(ctx->stream);
    kc_release(ctx, d_ptr);
}

// result -> file descriptor through two pinned chunks: the device-to-host copy of piece i+1 overlaps the write() of piece i.
// (Measured on the MI355X box, 2.5 GB to tmpfs: this sequential stream 0.42 s; four threads with pwrite 0.68 s -- buffered
// writes to one file queue on the inode lock; eight threads copying into a shared mapping 1.2 s -- the page faults are slower
// than write()'s page allocation.  The single memcpy-bound writer is the fastest of the three.)
extern "C" int kc_result_write_fd(kc_result *r, int fd, int which) {
    if (!r || fd < 0) return KC_E_ARG;
    kc_ctx *ctx = r->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    const char *src = which ? r->d_mask : r->d_out;
    if (!src) return kc_fail(ctx, KC_E_ARG, "the result holds no such string");
    const u64 total = r->slice_len;
    const size_t CH = (size_t)32 << 20;
    char *hb[2] = {nullptr, nullptr};
    hipEvent_t ev[2];
    int rc = KC_OK;
    for (int i = 0; i < 2 && rc == KC_OK; ++i) {
        if (hipHostMalloc((void **)&hb[i], CH, hipHostMallocDefault) != hipSuccess) rc = kc_fail(ctx, KC_E_NO_DEVICE, "pinned buffer allocation failed");
        if (rc == KC_OK && hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) rc = KC_E_NO_DEVICE;
    }
    // A regular file whose blocks are already allocated up to the end of this write (the CLI preallocates the output while the
    // GPU computes) takes its chunks through four pwrite()s at once: into allocated pages a write is a plain memory copy and
    // scales with threads; where pages still have to be allocated, parallel writers only contend (measured: slower than one).
    off_t fpos = -1;
    {
        struct stat st;
        if (!getenv("KC_IO_SERIAL") && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && !(fcntl(fd, F_GETFL) & O_APPEND)) {
            const off_t cur = lseek(fd, 0, SEEK_CUR);
            if (cur >= 0 && (u64)st.st_blocks * 512 >= (u64)cur + total) fpos = cur;
        }
    }
    u64 sent = 0, written = 0;
    int slot = 0;
    u64 len_in[2] = {0, 0};
    if (rc == KC_OK && total) {
        len_in[0] = total < CH ? total : CH;
        if (hipMemcpyAsync(hb[0], src, len_in[0], hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipEventRecord(ev[0], ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
        sent = len_in[0];
    }
    while (rc == KC_OK && written < total) {
        const int nxt = slot ^ 1;
        if (sent < total) {
            len_in[nxt] = total - sent < CH ? total - sent : CH;
            if (hipMemcpyAsync(hb[nxt], src + sent, len_in[nxt], hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipEventRecord(ev[nxt], ctx->stream) != hipSuccess) { rc = KC_E_NO_DEVICE; break; }
            sent += len_in[nxt];
        }
        if (hipEventSynchronize(ev[slot]) != hipSuccess) { rc = KC_E_NO_DEVICE; break; }
        if (fpos >= 0) {
            const int T = len_in[slot] >= ((u64)8 << 20) ? 4 : 1;
            std::atomic<bool> bad{false};
            const char *buf = hb[slot];
            const u64 n = len_in[slot];
            const off_t at = fpos + (off_t)written;
            auto part = [&](int t) {
                u64 lo = n * (u64)t / (u64)T, hi = n * (u64)(t + 1) / (u64)T;
                while (lo < hi) {
                    const ssize_t w = pwrite(fd, buf + lo, hi - lo, at + (off_t)lo);
                    if (w <= 0) { bad = true; return; }
                    lo += (u64)w;
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t) th.emplace_back(part, t);
            part(0);
            for (auto &x : th) x.join();
            if (bad) { rc = kc_fail(ctx, KC_E_IO, "write failed"); break; }
        } else {
            u64 off = 0;
            while (off < len_in[slot]) {
                const ssize_t w = ::write(fd, hb[slot] + off, len_in[slot] - off);
                if (w < 0) { rc = kc_fail(ctx, KC_E_IO, "write failed"); break; }
                off += (u64)w;
            }
        }
        written += len_in[slot];
        slot = nxt;
    }
    if (fpos >= 0 && rc == KC_OK) (void)lseek(fd, fpos + (off_t)total, SEEK_SET);   // the caller goes on writing behind the string
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 2; ++i) { if (hb[i]) (void)hipHostFree(hb[i]); (void)hipEventDestroy(ev[i]); }
    return rc;
}

// Same pipeline, destination = host memory: chunk i is copied on by four threads while chunk i+1 arrives.  For a pageable
// destination this beats one big hipMemcpy (which stages through a single thread's copies), and it is what the CLI uses to
// fill the (prefaulted) shared mapping of its output file -- stores into a mapping take no inode lock, so they scale.
extern "C" int kc_result_write_mem(kc_result *r, void *dst_v, int which) {
    if (!r || !dst_v) return KC_E_ARG;
    kc_ctx *ctx = r->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    const char *src = which ? r->d_mask : r->d_out;
    if (!src) return kc_fail(ctx, KC_E_ARG, "the result holds no such string");
    char *dst = static_cast<char *>(dst_v);
    const u64 total = r->slice_len;
    const size_t CH = (size_t)32 << 20;
    char *hb[2] = {nullptr, nullptr};
    hipEvent_t ev[2];
    int rc = KC_OK;
    for (int i = 0; i < 2 && rc == KC_OK; ++i) {
        if (hipHostMalloc((void **)&hb[i], CH, hipHostMallocDefault) != hipSuccess) rc = kc_fail(ctx, KC_E_NO_DEVICE, "pinned buffer allocation failed");
        if (rc == KC_OK && hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) rc = KC_E_NO_DEVICE;
    }
    u64 sent = 0, done = 0;
    int slot = 0;
    u64 len_in[2] = {0, 0};
    if (rc == KC_OK && total) {
        len_in[0] = total < CH ? total : CH;
        if (hipMemcpyAsync(hb[0], src, len_in[0], hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipEventRecord(ev[0], ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
        sent = len_in[0];
    }
    while (rc == KC_OK && done < total) {
        const int nxt = slot ^ 1;
        if (sent < total) {
            len_in[nxt] = total - sent < CH ? total - sent : CH;
            if (hipMemcpyAsync(hb[nxt], src + sent, len_in[nxt], hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipEventRecord(ev[nxt], ctx->stream) != hipSuccess) { rc = KC_E_NO_DEVICE; break; }
            sent += len_in[nxt];
        }
        if (hipEventSynchronize(ev[slot]) != hipSuccess) { rc = KC_E_NO_DEVICE; break; }
        const u64 n = len_in[slot];
        const int T = (n >= ((u64)8 << 20) && !getenv("KC_IO_SERIAL")) ? kc_io_threads() : 1;
        const char *buf = hb[slot];
        char *at = dst + done;
        auto part = [&](int t) { const u64 lo = n * (u64)t / (u64)T, hi = n * (u64)(t + 1) / (u64)T; memcpy(at + lo, buf + lo, hi - lo); };
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(part, t);
        part(0);
        for (auto &x : th) x.join();
        done += n;
        slot = nxt;
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 2; ++i) { if (hb[i]) (void)hipHostFree(hb[i]); (void)hipEventDestroy(ev[i]); }
    return rc;
}
