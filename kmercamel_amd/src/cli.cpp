// kmercamel (MI355X build): the `compute` (greedy, incl. -S / -M / -z), `maskopt` (max-one / min-one), `lowerbound` and the
// four format-converter subcommands of the reference CLI (main.cpp:214-691) with the same options, validation messages,
// log lines and output format, running the hot path on the GPU through the C ABI.  `compute` uses the fused
// device-resident pipeline (kc_compute); the converters are host-side byte shuffles (conversions.h); everything the
// hot-path scope leaves out (other algorithms, min-run masks) is rejected explicitly.
#include <algorithm>
#include <chrono>
#include <thread>
#include <sys/stat.h>
#include <sys/mman.h>
#include <sstream>
#include <cstring>
#include <cerrno>
#include <ctime>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include <unistd.h>
#include <fcntl.h>
#include <getopt.h>
#include <atomic>
#include <functional>
#include <condition_variable>
#include <mutex>
#include "global.h"
#include "parser.h"
#include "conversions.h"
#include "kc_runtime.h"

static const char *KC_CLI_VERSION = "v2.3.0-mi355x";
static const int MAX_K = 127;   // main.cpp:93

namespace {

// subs: letters c(ompute) m(askopt) l(owerbound) s(ms2mssep) j(mssep2ms) r(ms2spss) p(spss2ms)
struct Opt { const char *flag, *arg, *help; const char *subs; };
const Opt OPTIONS[] = {
    {"-k", "INT ", "k-mer size [required; up to 127]", "cmlrp"},
    {"-a", "STR ", "the algorithm to be run [greedy (default); the MI355X build implements greedy only]", "c"},
    {"-t", "STR ", "the target mask type to be run [max-one (default), min-one]", "m"},
    {"-o", "FILE", "output for the (min-one) masked superstring; if not specified, printed to stdout", "cmjrp"},
    {"--gpus", "INT", "number of GPUs (ranks) the k-mer set is hash-sharded over [1 (default); also KMERCAMEL_GPUS]", "c"},
    {"-M", "FILE", "if given, print also ms with mask maximizing ones (only with greedy)", "c"},
    {"-S", "    ", "optimize for the input being pre-computed simplitigs or unitigs (only with greedy)", "cl"},
    {"-u", "    ", "treat k-mer and its reverse complement as distinct", "cml"},
    {"-z", "INT ", "minimum frequency to represent a k-mer; default 1", "cl"},
    {"-m", "FILE", "input file with mask", "j"},
    {"-s", "FILE", "input file with superstring", "j"},
    {"-m", "FILE", "output file with mask", "s"},
    {"-s", "FILE", "output file with superstring", "s"},
    {"-h", "    ", "print help", "cmlsjrp"},
};

int usage() {
    std::cerr << "\nProgram: KmerCamel - a tool for computation of masked superstrings (MI355X-native greedy path).\n"
              << "Version: " << KC_CLI_VERSION << "\n\nUsage:   kmercamel <command> [options]\n\nCommand:\n"
              << "    compute    - Compute an approximately shortest masked superstring from input FASTA.\n"
              << "    maskopt    - Optimize a given masked superstring (max-one / min-one).\n"
              << "    ms2mssep   - Split a masked superstring into a superstring and a mask.\n"
              << "    mssep2ms   - Join a masked superstring from a superstring and a mask.\n"
              << "    ms2spss    - Compute rSPSS from a masked superstring (may be of much larger total length).\n"
              << "    spss2ms    - Compute masked superstring corresponding to (r)SPSS.\n"
              << "    lowerbound - Compute the lower bound on masked superstring size of a k-mer set.\n\n";
    return 1;
}

int usage_subcommand(const std::string &sub) {
    char tag = sub == "compute" ? 'c' : sub == "maskopt" ? 'm' : sub == "ms2mssep" ? 's' : sub == "mssep2ms" ? 'j' : sub == "ms2spss" ? 'r'
             : sub == "spss2ms" ? 'p' : 'l';
    const char *positional = (tag == 'c' || tag == 'l' || tag == 'p') ? " <fasta>" : (tag == 'j' ? "" : " <ms>");   // main.cpp:45-48
    std::cerr << "\nUsage:   kmercamel " << sub << " [options]" << positional << "\n\nOptions:\n";
    for (const Opt &o : OPTIONS)
        if (std::string(o.subs).find(tag) != std::string::npos) std::cerr << "  " << o.flag << " " << o.arg << " - " << o.help << "\n";
    std::cerr << std::endl;
    return 1;
}

std::string normalize_compute_algorithm(const std::string &a) {   // main.cpp:97-106
    if (a == "global") return "greedy";
    if (a == "local" || a == "localgreedy") return "local-greedy";
    if (a == "globalAC" || a == "greedyAC") return "greedy-ac";
    if (a == "localAC" || a == "localgreedyAC") return "local-greedy-ac";
    return a;
}
std::string normalize_maskopt_algorithm(const std::string &a) {   // main.cpp:109-115
    if (a == "maxone") return "max-one";
    if (a == "minone") return "min-one";
    if (a == "minrun") return "min-run";
    if (a == "approxminrun") return "approx-min-run";
    return a;
}

struct Records {
    std::string joined;               // sequences joined by '\n' (compute) or concatenated (-S)
    std::vector<uint64_t> offsets;    // -S only
    uint64_t n = 0;
};
Records load_records(const std::string &path, bool simplitigs) {
    Records r;
    const std::string buf = kcamd::slurp_fast(path);
    if (kcamd::join_fasta(buf, simplitigs, r.joined, r.offsets, r.n)) return r;   // FASTA: one pass, no per-record strings
    r = Records();
    auto recs = kcamd::parse_records(buf);
    r.n = recs.size();
    size_t total = 0;
    for (auto &x : recs) total += x.seq.size() + 1;
    r.joined.reserve(total);
    for (auto &x : recs) {
        if (simplitigs) r.offsets.push_back(r.joined.size());
        r.joined += x.seq;
        if (!simplitigs) r.joined.push_back('\n');
    }
    if (simplitigs) r.offsets.push_back(r.joined.size());
    return r;
}

// greedy branch of kmercamel<>() (main.cpp:122-212)
// all of `data` to a file descriptor
static void write_all(int fd, const char *data, size_t n) {
    while (n) {
        ssize_t w = ::write(fd, data, n);
        if (w < 0) throw std::runtime_error("write failed");
        data += w; n -= (size_t)w;
    }
}

// The output file of a large run gets its pages while the device comes up and computes -- fallocate(), one call, one thread: measured on the
// MI355X box for 2.55 GB on tmpfs 125 ms, against 370 ms for faulting the pages in through a mapping with one thread and 630 ms with eight
// (page allocation for ONE file serialises on the inode; tools/io_probe2.cpp) -- and is mapped.  The result then goes straight into
// the page cache through the mapping (kc_result_write_mem: stores into a mapping take no inode lock, eight threads move each chunk at
// ~27 GB/s, first-touch faults included; a write() stream manages 7 GB/s) and the file is cut to its final length.  Anything unusual
// (not a regular file, small input, mapping refused, result longer than estimated) leaves `ok` unset: the caller takes the plain
// descriptor path.  Used by `compute` and `maskopt`.
struct MappedOutput {
    int fd = -1; char *map = nullptr; size_t cap = 0; bool ok = false, committed = false;
    std::thread th;
    void settle() { if (th.joinable()) th.join(); }
    ~MappedOutput() {
        settle();
        if (map) ::munmap(map, cap);
        if (fd >= 0) { if (!committed && ::ftruncate(fd, 0) != 0) {} ::close(fd); }
    }
    // in_path: the input file (its size bounds the output's: header + that + slack); out_path: the -o file, already created by the caller
    void prepare(const std::string &in_path, const std::string &out_path, size_t hdr_bytes, bool trace, std::chrono::steady_clock::time_point t_start) {
        if (std::getenv("KC_CLI_NOPREALLOC")) return;
        struct stat sti, sto;
        if (!(::stat(in_path.c_str(), &sti) == 0 && S_ISREG(sti.st_mode) && sti.st_size > (off_t)(64 << 20) && ::stat(out_path.c_str(), &sto) == 0 && S_ISREG(sto.st_mode))) return;
        cap = hdr_bytes + (size_t)sti.st_size + 4096;
        const int f = ::open(out_path.c_str(), O_RDWR);
        if (f < 0) return;
        if (::ftruncate(f, (off_t)cap) != 0) { ::close(f); return; }
        fd = f;   // (from here on the destructor restores the empty file unless the result is committed)
        void *m = ::mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) return;
        map = static_cast<char *>(m);
        ok = true;
        // Then the pages are mapped in by four threads with write touches that change nothing (an atomic OR of 0: the result may
        // already be arriving in the front of the file while the back is still being touched).  Touches of allocated pages
        // scale with threads where allocation does not, and a copy into mapped pages runs at the speed of the PCIe transfer
        // (measured: 2.55 GB in ~50 ms against 105-200 ms when every store of the copy first had to fault its page in).
        // MADV_POPULATE_WRITE is NOT used: unmapping pages it brought in took twice as long (90-110 ms against 43 ms).
        // (a file system without fallocate: the pages then come with these touches, as for any mapping)
        th = std::thread([this, trace, t_start] {
            const int rc = ::fallocate(fd, 0, 0, (off_t)cap) == 0 ? 0 : errno;
            if (trace) std::cerr << "[cli] output pages allocated (rc " << rc << "): " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count() << " ms" << std::endl;
            const char *te = std::getenv("KC_CLI_TOUCH_THREADS");
            const int T = te ? std::atoi(te) : 4;
            if (T < 1) return;
            const size_t PIECE = (size_t)8 << 20;
            auto part = [this, T, PIECE](int t) {
                for (size_t lo = (size_t)t * PIECE; lo < cap; lo += (size_t)T * PIECE) {
                    const size_t hi = lo + PIECE < cap ? lo + PIECE : cap;
                    for (size_t p = lo; p < hi; p += 4096) __atomic_fetch_or(map + p, (char)0, __ATOMIC_RELAXED);
                }
            };
            std::vector<std::thread> ths;
            for (int t = 1; t < T; ++t) ths.emplace_back(part, t);
            part(0);
            for (auto &x : ths) x.join();
            if (trace) std::cerr << "[cli] output pages mapped: " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count() << " ms" << std::endl;
        });
    }
    // the process is about to end abnormally while other threads may still be storing into the mapping (the touch threads, other ranks'
    // copy threads): anonymous memory takes the mapping's place FIRST -- a store past the new end of a truncated MAP_SHARED file would
    // raise SIGBUS -- then the file goes back to empty.  No destructor runs behind this (the caller leaves through _exit).
    void neutralise() {
        if (map) (void)::mmap(map, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED, -1, 0);
        if (fd >= 0 && ::ftruncate(fd, 0) != 0) {}
    }
    // the mapping cannot take this result (or is not wanted after all): the file goes back to empty, the caller writes through the descriptor
    void abandon() {
        settle();
        if (map) { ::munmap(map, cap); map = nullptr; }
        if (fd >= 0) { if (::ftruncate(fd, 0) != 0) {} ::close(fd); fd = -1; }
        ok = false;
    }
    // header line, the result's characters (which: 0 string, 1 mask), '\n'; the file is cut to that length
    void commit(kc_result *res, const std::string &hdr, uint64_t length, int which, const std::function<void(const char *)> &mark) {
        kcamd::check(kc_result_write_mem(res, map + hdr.size(), which));
        mark("result in the mapping");
        finish(hdr, length, mark);
    }
    // the characters are in place behind where the header goes (written by one caller, or slice by slice by the ranks of `--gpus N`)
    void finish(const std::string &hdr, uint64_t length, const std::function<void(const char *)> &mark) {
        memcpy(map, hdr.data(), hdr.size());
        settle();
        map[hdr.size() + length] = '\n';   // main.cpp:210
        // munmap tears the 6 10^5 page-table entries of the mapping down under the exclusive mmap lock, 40 - 100 ms for this file; the same
        // entries dropped by MADV_DONTNEED -- which takes the lock shared, so eight threads work side by side, and which leaves the data
        // of a shared file mapping where it is, in the page cache -- make the munmap behind them a formality.  (KC_CLI_NO_ZAP: A/B knob)
        if (!std::getenv("KC_CLI_NO_ZAP")) {
            const int T = 8;
            const size_t per = ((cap / T) + 4095) & ~(size_t)4095;
            std::vector<std::thread> zap;
            for (int t = 0; t < T; ++t) {
                const size_t lo = (size_t)t * per;
                if (lo >= cap) break;
                const size_t n = lo + per < cap ? per : cap - lo;
                zap.emplace_back([this, lo, n] { (void)::madvise(map + lo, n, MADV_DONTNEED); });
            }
            for (auto &z : zap) z.join();
            mark("mapping dropped");
        }
        ::munmap(map, cap); map = nullptr;
        if (::ftruncate(fd, (off_t)(hdr.size() + length + 1)) != 0) throw std::runtime_error("cannot size the output file");
        committed = true;
    }
};

// `compute --gpus N` on a large plain FASTA file: every rank thread reads ITS byte range of the file through the device tokeniser
// (kc_read_fasta_device_range: the ranges begin at header lines), the ranks compute one superstring (kc_compute_sharded on device input),
// and every rank copies its slice straight into the mapping of the fallocate'd output file -- no host copy of the input, no host copy of the
// output, nothing done by one thread for all.  Returns KC_CLI_STREAM_NA (nothing read, nothing written) when the input is not a plain
// regular file of more than 64 MB or some rank's share does not look like FASTA: the caller then takes the host-parser form below.
static const int KC_CLI_SHARDED_FALLBACK = -1000, KC_CLI_STREAM_NA = -1001;
static int run_compute_sharded_streamed(const std::string &path, int k, std::ostream *of, std::ostream *maskf, bool complements, int gpus, int min_frequency,
                                        const std::string &out_path, const std::string &mask_path, std::string *why) {
    struct stat sti;
    if (path == "-" || std::getenv("KC_CLI_NOSTREAM") || ::stat(path.c_str(), &sti) != 0 || !S_ISREG(sti.st_mode) || sti.st_size <= (off_t)(64 << 20)) return KC_CLI_STREAM_NA;
    const uint64_t fsize = (uint64_t)sti.st_size;
    // rank r reads [cut[r], cut[r + 1]): cut[r] = the first header line that starts at or after fsize * r / N
    std::vector<uint64_t> cut(gpus + 1, fsize);
    cut[0] = 0;
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return KC_CLI_STREAM_NA;
        unsigned char magic[2] = {0, 0};
        if (fread(magic, 1, 2, f) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) { fclose(f); return KC_CLI_STREAM_NA; }   // gzip: one stream
        std::vector<char> win((size_t)4 << 20);
        for (int r = 1; r < gpus; ++r) {
            uint64_t p = fsize / (uint64_t)gpus * (uint64_t)r;
            if (p < cut[r - 1]) p = cut[r - 1];
            uint64_t found = fsize;
            // (look at the byte before p too: a '>' exactly at p counts only if a line starts there)
            for (uint64_t at = p ? p - 1 : 0; at < fsize && found == fsize; at += win.size() - 1) {
                if (fseeko(f, (off_t)at, SEEK_SET) != 0) break;
                const size_t got = fread(win.data(), 1, win.size(), f);
                if (got < 2) break;
                for (size_t i = (at == 0 && p == 0) ? 0 : 1; i < got; ++i)
                    if (win[i] == '>' && (at + i == 0 || win[i - 1] == '\n') && at + i >= p) { found = at + i; break; }
            }
            cut[r] = found;
        }
        fclose(f);
    }
    const bool trace = std::getenv("KC_CLI_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (trace) std::cerr << "[cli] " << what << ": " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count() << " ms" << std::endl;
    };
    WriteLog("Started computation of a masked superstring from '" + path + "'.");
    MappedOutput om, omm;
    std::string hdr, mhdr;
    if (!out_path.empty()) {
        std::ostringstream hs;
        WriteName(path, "greedy", k, false, !complements, hs);
        hdr = hs.str();
        om.prepare(path, out_path, hdr.size(), trace, t_start);
    }
    if (maskf != nullptr && !mask_path.empty()) {
        std::ostringstream hs;
        WriteName(path, "greedy", k, true, !complements, hs);
        mhdr = hs.str();
        omm.prepare(path, mask_path, mhdr.size(), trace, t_start);
    }
    const int ndev = kc_device_count();
    if (ndev <= 0) throw std::runtime_error("kmercamel_amd: no usable MI355X / HIP device");
    kc_thread_group *group = nullptr;
    if (kc_thread_group_create(gpus, &group) != KC_OK) throw std::runtime_error("kmercamel_amd: cannot create the rank group");
    std::vector<kc_result *> res(gpus, nullptr);
    std::vector<std::string> err(gpus);
    std::vector<int> rcs(gpus, KC_OK);
    std::vector<uint64_t> st(16, 0);
    std::string out, mask;   // (only when the output cannot be mapped)
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, not_fasta = 0;
    bool out_ready = false, mapped = false, mmapped = false;
    // a rank that cannot go on ends the process (its peers would wait for it forever) -- but not with output files of the preallocated
    // size and whatever the pages hold: they go back to empty first (MappedOutput::neutralise; no destructors run behind _exit)
    auto die = [&]() {
        om.neutralise();    // (both mappings turned harmless and both files emptied before anything can fault)
        omm.neutralise();
        _exit(1);
    };
    auto rank_main = [&](int r) {
        kc_ctx *ctx = nullptr;
        kc_comm *comm = nullptr;
        int rc = kc_ctx_create(r % ndev, &ctx);
        if (rc == KC_OK) rc = kc_comm_create_threads(ctx, group, r, &comm);
        if (rc != KC_OK) { std::cerr << "kmercamel_amd: rank " << r << " cannot start (code " << rc << ")" << std::endl; die(); }   // (the others would wait for it forever)
        void *d_seq = nullptr;
        uint64_t d_len = 0;
        int fq = 0;
        rc = kc_read_fasta_device_range(ctx, path.c_str(), cut[r], cut[r + 1] - cut[r], &d_seq, &d_len, &fq);
        {   // the ranks agree on whether this is input for the streamed path before any of them enters the collective
            std::unique_lock<std::mutex> lk(mu);
            if (rc != KC_OK || fq) ++not_fasta;
            if (++arrived == gpus) cv.notify_all(); else cv.wait(lk, [&] { return arrived == gpus; });
        }
        if (not_fasta) { if (d_seq) kc_device_free(ctx, d_seq); rcs[r] = KC_CLI_STREAM_NA; kc_comm_destroy(comm); kc_ctx_destroy(ctx); return; }
        if (r == 0) mark("every rank has its share of the input on its device");
        rc = kc_compute_sharded(ctx, comm, d_seq, d_len, k, (complements ? KC_FLAG_COMPLEMENTS : 0) | (maskf ? KC_FLAG_WANT_MASK : 0) | KC_FLAG_INPUT_ON_DEVICE |
                                (min_frequency > 1 ? KC_FLAG_MIN_FREQUENCY(min_frequency) : 0), &res[r]);
        if (d_seq) kc_device_free(ctx, d_seq);
        rcs[r] = rc;
        if (rc != KC_OK) err[r] = kc_last_error(ctx);
        const bool sparse_case = rc == KC_E_UNSUPPORTED_SHARDED;   // (a request the sharded path does not serve: one GPU takes it)
        if (rc != KC_OK && rc != KC_E_NO_KMERS && !sparse_case) { std::cerr << "kmercamel_amd: rank " << r << ": " << err[r] << " (code " << rc << ")" << std::endl; die(); }
        if (rc == KC_OK) {
            if (r == 0) {
                mark("kc_compute_sharded returned");
                kc_result_stats(res[0], st.data());
                {
                    std::lock_guard<std::mutex> lk(mu);
                    mapped = om.ok && hdr.size() + st[2] + 1 <= om.cap;
                    mmapped = omm.ok && mhdr.size() + st[2] + 1 <= omm.cap;
                    if (!mapped) out.resize(st[2]);
                    if (maskf && !mmapped) mask.resize(st[2]);
                    out_ready = true;
                }
                cv.notify_all();
            } else {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return out_ready; });
            }
            uint64_t off = 0, n = 0;
            kc_result_slice(res[r], &off, &n);
            int wrc = KC_OK;
            if (n) {
                // every rank moves its slice itself: into the mapping where there is one (pinned chunks, eight copy threads per rank)
                if (mapped) wrc = kc_result_write_mem(res[r], om.map + hdr.size() + off, 0);
                else wrc = kc_result_copy(res[r], &out[off], nullptr);
                if (wrc == KC_OK && maskf) {
                    if (mmapped) wrc = kc_result_write_mem(res[r], omm.map + mhdr.size() + off, 1);
                    else { std::vector<char> tmp(n); wrc = kc_result_copy(res[r], tmp.data(), &mask[off]); }
                }
            }
            if (wrc != KC_OK) { std::cerr << "kmercamel_amd: rank " << r << " cannot copy its slice" << std::endl; die(); }
            kc_result_destroy(res[r]);
        }
        kc_comm_destroy(comm);
        kc_ctx_destroy(ctx);
    };
    std::vector<std::thread> th;
    for (int r = 1; r < gpus; ++r) th.emplace_back(rank_main, r);
    rank_main(0);
    for (auto &t : th) t.join();
    kc_thread_group_destroy(group);
    if (rcs[0] == KC_CLI_STREAM_NA) { om.abandon(); omm.abandon(); return KC_CLI_STREAM_NA; }
    if (rcs[0] == KC_E_NO_KMERS) {
        om.abandon(); omm.abandon();
        std::cerr << "Path '" << path << "' contains no k-mers. Make sure that your file is a FASTA or gzipped FASTA." << std::endl;
        return usage_subcommand("compute");
    }
    if (rcs[0] == KC_E_UNSUPPORTED_SHARDED) { om.abandon(); omm.abandon(); *why = err[0]; return KC_CLI_SHARDED_FALLBACK; }
    mark("every slice is in place");
    WriteLog("Finished collecting k-mers: " + std::to_string(st[0]) + " " + std::to_string(k) + "-mers.");
    if (!mapped) { if (om.fd >= 0) om.abandon(); WriteName(path, "greedy", k, false, !complements, *of); }
    if (maskf != nullptr && !mmapped) { if (omm.fd >= 0) omm.abandon(); WriteName(path, "greedy", k, true, !complements, *maskf); }
    WriteLog("Finished 1. part: simplitigs (" + std::to_string(st[1]) + " simplitigs).");
    WriteLog("Finished 2. part: Hamiltonian path.");
    if (mapped) om.finish(hdr, st[2], mark); else of->write(out.data(), (std::streamsize)out.size());
    if (maskf != nullptr) { if (mmapped) omm.finish(mhdr, st[2], mark); else { maskf->write(mask.data(), (std::streamsize)mask.size()); *maskf << std::endl; } }   // global.h:206-208
    WriteLog("Finished 3. part: masked superstring (l=" + std::to_string(st[2]) + ").");
    if (!mapped) *of << std::endl;
    mark("output written");
    return 0;
}

// `compute` over several GPUs (--gpus N / KMERCAMEL_GPUS; the reference is one thread, main.cpp:122-212): one rank thread per GPU (ranks
// beyond the number of devices share them round robin), each with its own context and a communicator of the in-process
// thread transport; the records are cut into N contiguous parts, kc_compute_sharded builds ONE masked superstring --
// byte-identical to the single-GPU result -- and every rank copies its slice into place.
static int run_compute_sharded(const std::string &path, int k, std::ostream *of, std::ostream *maskf, bool complements, int gpus, int min_frequency, bool assume_simplitigs,
                               std::string *why, Records *loaded) {
    WriteLog("Started computation of a masked superstring from '" + path + "'.");
    // (the records stay with the caller: if the ranks hand the input back -- sparse branch, -z overflow -- the single-GPU run must not
    // read the input a second time; stdin, FIFOs and process substitutions can be read only once)
    *loaded = load_records(path, assume_simplitigs);
    Records &recs = *loaded;
    const std::string &j = recs.joined;
    // part r = the records from the first line start at or after byte len * r / N (-S: records rec_cut[r] .. rec_cut[r + 1], with their
    // offsets counted from the part's first byte)
    std::vector<size_t> cut(gpus + 1, j.size());
    std::vector<uint64_t> rec_cut(gpus + 1, recs.n);
    std::vector<std::vector<uint64_t>> part_offsets(gpus);
    cut[0] = 0; rec_cut[0] = 0;
    if (assume_simplitigs) {
        for (int r = 1; r < gpus; ++r) { rec_cut[r] = recs.n * (uint64_t)r / (uint64_t)gpus; cut[r] = recs.offsets[rec_cut[r]]; }
        for (int r = 0; r < gpus; ++r)
            for (uint64_t i = rec_cut[r]; i <= rec_cut[r + 1]; ++i) part_offsets[r].push_back(recs.offsets[i] - recs.offsets[rec_cut[r]]);
    } else
    for (int r = 1; r < gpus; ++r) {
        size_t p = j.size() / gpus * r;
        if (p < cut[r - 1]) p = cut[r - 1];
        while (p < j.size() && p > 0 && j[p - 1] != '\n') ++p;
        cut[r] = p;
    }
    const int ndev = kc_device_count();
    if (ndev <= 0) throw std::runtime_error("kmercamel_amd: no usable MI355X / HIP device");
    kc_thread_group *group = nullptr;
    if (kc_thread_group_create(gpus, &group) != KC_OK) throw std::runtime_error("kmercamel_amd: cannot create the rank group");
    std::vector<kc_result *> res(gpus, nullptr);
    std::vector<std::string> err(gpus);
    std::vector<int> rcs(gpus, KC_OK);
    std::vector<uint64_t> st(16, 0);
    std::string out, mask;
    std::mutex out_mu;
    std::condition_variable out_cv;
    bool out_ready = false;
    auto rank_main = [&](int r) {
        kc_ctx *ctx = nullptr;
        kc_comm *comm = nullptr;
        int rc = kc_ctx_create(r % ndev, &ctx);
        if (rc == KC_OK) rc = kc_comm_create_threads(ctx, group, r, &comm);
        if (rc != KC_OK) { std::cerr << "kmercamel_amd: rank " << r << " cannot start (code " << rc << ")" << std::endl; _exit(1); }   // (the others would wait for it forever)
        if (assume_simplitigs)
            rc = kc_compute_sharded_simplitigs(ctx, comm, j.data() + cut[r], cut[r + 1] - cut[r], part_offsets[r].data(), rec_cut[r + 1] - rec_cut[r], k,
                                               complements ? KC_FLAG_COMPLEMENTS : 0, &res[r]);
        else {
            const uint64_t len = cut[r + 1] > cut[r] ? cut[r + 1] - cut[r] - (cut[r + 1] == j.size() ? 0 : 1) : 0;   // (the separator after the part is not part of it)
            rc = kc_compute_sharded(ctx, comm, j.data() + cut[r], len, k, (complements ? KC_FLAG_COMPLEMENTS : 0) | (maskf ? KC_FLAG_WANT_MASK : 0) | (min_frequency > 1 ? KC_FLAG_MIN_FREQUENCY(min_frequency) : 0), &res[r]);
        }
        rcs[r] = rc;
        if (rc != KC_OK) err[r] = kc_last_error(ctx);
        // S >= N / 5 (main.cpp:175, the sparse branch): every rank learns it from the same replicated simplitig list and returns together
        const bool sparse_case = rc == KC_E_UNSUPPORTED_SHARDED;   // (a request the sharded path does not serve: one GPU takes it)
        if (rc != KC_OK && rc != KC_E_NO_KMERS && !sparse_case) { std::cerr << "kmercamel_amd: rank " << r << ": " << err[r] << " (code " << rc << ")" << std::endl; _exit(1); }
        if (rc == KC_OK) {
            if (r == 0) {
                kc_result_stats(res[0], st.data());
                { std::lock_guard<std::mutex> lk(out_mu); out.resize(st[2]); if (maskf) mask.resize(st[2]); out_ready = true; }
                out_cv.notify_all();
            } else {
                std::unique_lock<std::mutex> lk(out_mu);
                out_cv.wait(lk, [&] { return out_ready; });
            }
            uint64_t off = 0, n = 0;
            kc_result_slice(res[r], &off, &n);
            if (n && kc_result_copy(res[r], &out[off], maskf ? &mask[off] : nullptr) != KC_OK) { std::cerr << "kmercamel_amd: rank " << r << " cannot copy its slice" << std::endl; _exit(1); }
            kc_result_destroy(res[r]);
        }
        kc_comm_destroy(comm);
        kc_ctx_destroy(ctx);
    };
    std::vector<std::thread> th;
    for (int r = 1; r < gpus; ++r) th.emplace_back(rank_main, r);
    rank_main(0);
    for (auto &t : th) t.join();
    kc_thread_group_destroy(group);
    if (rcs[0] == KC_E_NO_KMERS) {
        std::cerr << "Path '" << path << "' contains no k-mers. Make sure that your file is a FASTA or gzipped FASTA." << std::endl;
        return usage_subcommand("compute");
    }
    if (rcs[0] == KC_E_UNSUPPORTED_SHARDED) { *why = err[0]; return KC_CLI_SHARDED_FALLBACK; }   // (what the ranks found out together, e.g. the sparse branch: the caller runs the input on one GPU)
    if (!assume_simplitigs) WriteLog("Finished collecting k-mers: " + std::to_string(st[0]) + " " + std::to_string(k) + "-mers.");
    WriteName(path, "greedy", k, false, !complements, *of);
    if (maskf != nullptr) WriteName(path, "greedy", k, true, !complements, *maskf);
    WriteLog("Finished 1. part: simplitigs (" + std::to_string(st[1]) + " simplitigs).");
    WriteLog("Finished 2. part: Hamiltonian path.");
    of->write(out.data(), (std::streamsize)out.size());
    if (maskf != nullptr) { maskf->write(mask.data(), (std::streamsize)mask.size()); *maskf << std::endl; }   // global.h:206-208
    WriteLog("Finished 3. part: masked superstring (l=" + std::to_string(st[2]) + ").");
    *of << std::endl;
    return 0;
}

int run_compute(const std::string &path, int k, std::ostream *of, std::ostream *maskf, bool complements, bool assume_simplitigs, bool lower_bound,
                int min_frequency, const std::string &out_path, const std::string &mask_path, int gpus = 1, bool gpus_explicit = true) {
    bool started_logged = false;
    Records recs;             // the input, once it has been read on the host
    bool have_recs = false;
    if (gpus > 1) {
        // what the hash-sharded path does not cover runs on one GPU, as without the option
        if (!lower_bound && (assume_simplitigs ? maskf == nullptr : k >= 18)) {
            std::string why;
            int rc = assume_simplitigs ? KC_CLI_STREAM_NA : run_compute_sharded_streamed(path, k, of, maskf, complements, gpus, min_frequency, out_path, mask_path, &why);
            const bool streamed_ranks = rc != KC_CLI_STREAM_NA;
            if (!streamed_ranks) rc = run_compute_sharded(path, k, of, maskf, complements, gpus, min_frequency, assume_simplitigs, &why, &recs);
            if (rc != KC_CLI_SHARDED_FALLBACK) return rc;
            std::cerr << "Note: " << why << "; running on one GPU." << std::endl;
            started_logged = true;
            have_recs = !streamed_ranks;   // (the host-parser form has read the records; after the streamed form the file is simply read again)
        } else if (gpus_explicit) std::cerr << "Note: --gpus applies to `compute -k K [-M F] [-z N]` with K >= 18 and to `compute -S` without -M; running on one GPU." << std::endl;
    }
    if (started_logged) { /* run_compute_sharded has logged the start */ }
    else if (!lower_bound) WriteLog("Started computation of a masked superstring from '" + path + "'.");
    else WriteLog("Started computation of a masked superstring length lower bound from '" + path + "'.");
    const bool trace = std::getenv("KC_CLI_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (trace) std::cerr << "[cli] " << what << ": " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count() << " ms" << std::endl;
    };
    kcamd::warm_context();   // the device comes up while the file is read
    MappedOutput om;
    std::string hdr;
    if (!lower_bound && !out_path.empty()) {
        std::ostringstream hs;
        WriteName(path, "greedy", k, false, !complements, hs);
        hdr = hs.str();
        om.prepare(path, out_path, hdr.size(), trace, t_start);
    }
    MappedOutput omm;   // -M: the mask file the same way
    std::string mhdr;
    if (!lower_bound && maskf != nullptr && !mask_path.empty()) {
        std::ostringstream hs;
        WriteName(path, "greedy", k, true, !complements, hs);
        mhdr = hs.str();
        omm.prepare(path, mask_path, mhdr.size(), trace, t_start);
    }
    kc_ctx *ctx = nullptr;
    void *d_seq = nullptr;
    uint64_t d_len = 0;
    bool streamed = false;
    // the device tokeniser consumes the input before it can say "this is FASTQ, use the host parser": only inputs that can be opened a
    // second time go through it (regular files); stdin, FIFOs and process substitutions are read once, by the host parser
    bool reopenable = false;
    {
        struct stat sti;
        reopenable = path != "-" && ::stat(path.c_str(), &sti) == 0 && S_ISREG(sti.st_mode);
    }
    if (!(assume_simplitigs && lower_bound) && reopenable && !have_recs && !std::getenv("KC_CLI_NOSTREAM")) {   // (`lowerbound` on raw FASTA and `compute -S` too)
        // FASTA goes to the device in pinned chunks and is tokenised there while the next chunk is read (no joined host copy)
        ctx = kcamd::context();
        mark("device context ready");
        int is_fastq = 0;
        int rc = kc_read_fasta_device(ctx, path.c_str(), &d_seq, &d_len, &is_fastq);
        if (rc == KC_E_IO) {   // parser.h:95-97 for a file that cannot be opened; a read / gzip error further in says so
            const std::string e = kc_last_error(ctx);
            if (e.find("read error") != std::string::npos) throw std::runtime_error(e);
            throw std::invalid_argument("couldn't open file " + path);
        }
        kcamd::check(rc);
        streamed = !is_fastq;
        if (streamed) mark("input on the device");
    }
    if (!streamed) {
        if (!have_recs) recs = load_records(path, assume_simplitigs);
        mark("records loaded");
        ctx = kcamd::context();
        mark("device context ready");
    }
    if (lower_bound) {
        // LowerBoundLength (lower_bound.h:7-36): simplitigs, then the cycle cover of the overlap graph
        kc_simplitigs *s = nullptr;
        if (assume_simplitigs) kcamd::check(kc_simplitigs_from_strings(ctx, k, recs.joined.data(), recs.offsets.data(), recs.n, &s));
        else {
            kc_set *set = nullptr;
            kcamd::check(kc_set_create(ctx, k, complements, streamed ? d_len : recs.joined.size(), &set));
            if (min_frequency > 1) kcamd::check(kc_set_enable_counting(set));
            if (streamed) { kcamd::check(kc_set_add_sequence_device(set, d_seq, d_len, 0)); kc_device_free(ctx, d_seq); d_seq = nullptr; }
            else kcamd::check(kc_set_add_sequence(set, recs.joined.data(), recs.joined.size(), 0));
            if (min_frequency > 1) kcamd::check(kc_set_filter_min_frequency(set, min_frequency));
            uint64_t n = 0;
            kcamd::check(kc_set_size(set, &n));
            if (!n) {
                std::cerr << "Path '" << path << "' contains no k-mers. Make sure that your file is a FASTA or gzipped FASTA." << std::endl;
                return usage_subcommand("compute");
            }
            WriteLog("Finished collecting k-mers: " + std::to_string(n) + " " + std::to_string(k) + "-mers.");
            kcamd::check(kc_simplitigs_build(set, &s));
            uint64_t S = 0;
            kc_simplitigs_count(s, &S, nullptr);
            if (S * 5 >= n) {   // main.cpp:175-181
                kc_simplitigs *sp = nullptr;
                kcamd::check(kc_simplitigs_to_sparse(s, &sp));
                kc_simplitigs_destroy(s);
                s = sp;
            }
            kc_set_destroy(set);
        }
        uint64_t S = 0, lb = 0;
        kc_simplitigs_count(s, &S, nullptr);
        WriteLog("Finished 1. part: simplitigs (" + std::to_string(S) + " simplitigs).");
        kcamd::check(kc_lower_bound(s, complements, &lb));
        kc_simplitigs_destroy(s);
        WriteLog("Finished 2. part: Hamiltonian path.");
        WriteLog("Finished 3. part: lower bound = " + std::to_string(lb) + ".");
        std::cout << lb;
        *of << std::endl;
        return 0;
    }
    uint32_t flags = (complements ? KC_FLAG_COMPLEMENTS : 0) | (assume_simplitigs ? KC_FLAG_SIMPLITIGS_INPUT : 0) | (maskf ? KC_FLAG_WANT_MASK : 0) |
                     KC_FLAG_MIN_FREQUENCY(min_frequency);
    kc_result *res = nullptr;
    int rc = streamed ? kc_compute(ctx, d_seq, d_len, nullptr, 0, k, flags | KC_FLAG_INPUT_ON_DEVICE, &res)
                      : kc_compute(ctx, recs.joined.data(), recs.joined.size(), assume_simplitigs ? recs.offsets.data() : nullptr, recs.n, k, flags, &res);
    if (streamed) kc_device_free(ctx, d_seq);
    if (rc == KC_E_NO_KMERS) {
        std::cerr << "Path '" << path << "' contains no k-mers. Make sure that your file is a FASTA or gzipped FASTA." << std::endl;
        return usage_subcommand("compute");
    }
    kcamd::check(rc);
    mark("kc_compute returned");
    if (trace) {
        uint64_t ps[4] = {0, 0, 0, 0};
        kc_ctx_stats(ctx, ps);
        std::cerr << "[cli] device pool: peak " << ps[0] / 1e9 << " GB, hipMalloc " << ps[1] / 1e9 << " GB in " << ps[2] << " calls, " << ps[3] / 1e3 << " ms inside hipMalloc" << std::endl;
    }
    uint64_t st[16];
    kc_result_stats(res, st);
    if (!assume_simplitigs) WriteLog("Finished collecting k-mers: " + std::to_string(st[0]) + " " + std::to_string(k) + "-mers.");
    const bool mapped = om.ok && hdr.size() + st[2] + 1 <= om.cap;
    if (!mapped && om.fd >= 0) om.abandon();
    if (!mapped) WriteName(path, "greedy", k, false, !complements, *of);
    const bool mmapped = omm.ok && mhdr.size() + st[2] + 1 <= omm.cap;
    if (!mmapped && omm.fd >= 0) omm.abandon();
    if (maskf != nullptr && !mmapped) WriteName(path, "greedy", k, true, !complements, *maskf);
    WriteLog("Finished 1. part: simplitigs (" + std::to_string(st[1]) + " simplitigs).");
    if (st[3]) WriteLog("2. part: Number of simplitigs over threshold, computing directly from k-mers.");
    WriteLog("Finished 2. part: Hamiltonian path.");
    // the characters leave through pinned chunks, device-to-host copy overlapping the write (kc_result_write_fd); the
    // streams only carry the header lines, so they are flushed first and the data follows through the descriptor
    of->flush();
    if (maskf) maskf->flush();
    auto emit = [&](const std::string &p, int which) {
        int fd = p.empty() ? 1 : ::open(p.c_str(), O_WRONLY | O_APPEND);
        if (fd < 0) throw std::runtime_error("cannot reopen " + p);
        kcamd::check(kc_result_write_fd(res, fd, which));
        if (which == 1) write_all(fd, "\n", 1);   // global.h:206-208
        if (!p.empty()) ::close(fd);
    };
    if (mapped) {
        om.commit(res, hdr, st[2], 0, mark);
    } else emit(out_path, 0);
    if (maskf != nullptr) { if (mmapped) omm.commit(res, mhdr, st[2], 1, mark); else emit(mask_path, 1); }
    kc_result_destroy(res);
    mark("result written");
    WriteLog("Finished 3. part: masked superstring (l=" + std::to_string(st[2]) + ").");
    if (!mapped) {
        int fd = out_path.empty() ? 1 : ::open(out_path.c_str(), O_WRONLY | O_APPEND);
        if (fd >= 0) { write_all(fd, "\n", 1); if (!out_path.empty()) ::close(fd); }   // main.cpp:210
    }
    mark("output written");
    return 0;
}

int camel_compute(int argc, char **argv, bool lowerbound) {
    std::string subcommand = lowerbound ? "lowerbound" : "compute";
    std::string path;
    if (argc > 1 && std::string(argv[argc - 1]) != "-h") { path = argv[argc - 1]; argc--; }
    int k = 0, d_max = 5, opt;
    std::ofstream output, maskOutput;
    std::ostream *of = &std::cout, *maskf = nullptr;
    std::string algorithm = "greedy", out_path, mask_path;
    bool complements = true, d_set = false, assume_simplitigs = false;
    int min_frequency = 1;
    int gpus = 1;
    bool gpus_explicit = false;   // --gpus on the command line (the environment variable applies silently where the sharded path does)
    if (const char *e = std::getenv("KMERCAMEL_GPUS")) {
        char *end = nullptr;
        const long v = std::strtol(e, &end, 10);
        if (*e && end && !*end && v >= 1 && v <= 64) gpus = (int)v;
        else std::cerr << "Warning: KMERCAMEL_GPUS='" << e << "' is not an integer between 1 and 64; ignored." << std::endl;
    }
    // the reference's options are short ones (main.cpp:234); the multi-GPU opt-in is a NEW long option, existing command lines are unchanged
    static const struct option long_opts[] = {{"gpus", required_argument, nullptr, 1000}, {nullptr, 0, nullptr, 0}};
    try {
        while ((opt = getopt_long(argc, argv, lowerbound ? "k:huxSz:" : "k:d:a:o:huxM:Sz:", long_opts, nullptr)) != -1) {
            switch (opt) {
                case 1000: gpus = std::stoi(optarg); gpus_explicit = true; break;
                case 'o': output.open(optarg); of = &output; out_path = optarg; break;
                case 'k': k = std::stoi(optarg); break;
                case 'd': d_set = true; d_max = std::stoi(optarg); break;
                case 'a': algorithm = optarg; break;
                case 'u': complements = false; break;
                case 'x': std::cerr << "Warning: The parameter -x currently has no effect due to the improvement in the underlying algorithm." << std::endl; break;
                case 'M': maskOutput.open(optarg); maskf = &maskOutput; mask_path = optarg; break;
                case 'S': assume_simplitigs = true; break;
                case 'h': usage_subcommand(subcommand); return 0;
                case 'z': min_frequency = std::stoi(optarg); break;
                default: return usage_subcommand(subcommand);
            }
        }
    } catch (std::invalid_argument &) {
        return usage_subcommand(subcommand);
    }
    algorithm = normalize_compute_algorithm(algorithm);
    auto fail = [&](const std::string &m) { std::cerr << m << std::endl; return usage_subcommand(subcommand); };
    if (path.empty()) return fail("Required positional parameter path to the file not set.");
    if (k == 0) return fail("Required parameter k not set.");
    if (k < 0) return fail("k must be positive.");
    if (d_max < 0) return fail("d must be non-negative.");
    if (k > MAX_K && (algorithm == "local-greedy" || algorithm == "greedy"))
        return fail("k > " + std::to_string(MAX_K) + " not supported for the algorithm '" + algorithm + "'. Use the  AC version of the algorithm instead.");
    if (d_set && (algorithm == "greedy-ac" || algorithm == "greedy" || algorithm == "streaming")) return fail("Unsupported argument d for algorithm '" + algorithm + "'.");
    if (maskf != nullptr && algorithm != "greedy")
        return fail("Outputting mask maximizing number of ones is possible only with greedy. For other algorithms, use separately kmercamel optimize.");
    if (assume_simplitigs && algorithm != "greedy") return fail("Optimization for the input being simplitigs is possible only with greedy.");
    if (min_frequency >= 256 || min_frequency < 1) return fail("Minimum frequency '-z' must be between 1 and 255.");
    if (min_frequency != 1 && assume_simplitigs) return fail("Inputting simplitigs is not compatible with frequency filterring.");
    if (algorithm != "greedy") return fail("Algorithm '" + algorithm + "' not supported by the MI355X build (greedy only).");
    if (gpus < 1 || gpus > 64) return fail("--gpus must be between 1 and 64.");
    return run_compute(path, k, of, maskf, complements, assume_simplitigs, lowerbound, min_frequency, out_path, mask_path, gpus, gpus_explicit);
}

// Optimize (masks.h:239-261) for max-one / min-one
int camel_optimize(int argc, char **argv) {
    std::string subcommand = "maskopt", path;
    if (argc > 1 && std::string(argv[argc - 1]) != "-h") { path = argv[argc - 1]; argc--; }
    int k = 0, opt;
    std::ofstream output;
    std::ostream *of = &std::cout;
    std::string algorithm = "max-one", out_path;
    bool complements = true;
    try {
        while ((opt = getopt(argc, argv, "k:t:o:hu")) != -1) {
            switch (opt) {
                case 'o': output.open(optarg); of = &output; out_path = optarg; break;
                case 'k': k = std::stoi(optarg); break;
                case 't': algorithm = optarg; break;
                case 'u': complements = false; break;
                case 'h': usage_subcommand(subcommand); return 0;
                default: return usage_subcommand(subcommand);
            }
        }
    } catch (std::invalid_argument &) {
        return usage_subcommand(subcommand);
    }
    algorithm = normalize_maskopt_algorithm(algorithm);
    auto fail = [&](const std::string &m) { std::cerr << m << std::endl; return usage_subcommand(subcommand); };
    if (path.empty()) return fail("Required positional parameter path to the file not set.");
    if (k == 0) return fail("Required parameter k not set.");
    if (k < 0) return fail("k must be positive.");
    WriteLog("Started optimization of a masked superstring from '" + path + "'.");
    if (algorithm == "min-run" || algorithm == "approx-min-run") return fail("Algorithm '" + algorithm + "' (GLPK ILP) is outside the MI355X build's scope.");
    if (algorithm != "max-one" && algorithm != "min-one") return fail("Algorithm '" + algorithm + "' not recognized.");
    // Large plain files take the streamed path of `compute`: the masked superstring goes to the device in pinned chunks and is tokenised
    // there (kc_read_fasta_device), optimised there (kc_maskopt_device) and leaves through the mapping of the fallocate'd output file --
    // 2.55 GB in and out in ~0.6 s where the iostream path below takes ~4 s.  Everything else (small inputs, gzip, stdin, FASTQ-looking
    // input, several records) goes through the host parser as before.
    {
        struct stat sti;
        const bool trace = std::getenv("KC_CLI_TRACE") != nullptr;
        const auto t_start = std::chrono::steady_clock::now();
        auto mark = [&](const char *what) {
            if (trace) std::cerr << "[cli] " << what << ": " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count() << " ms" << std::endl;
        };
        if (path != "-" && !std::getenv("KC_CLI_NOSTREAM") && ::stat(path.c_str(), &sti) == 0 && S_ISREG(sti.st_mode) && sti.st_size > (off_t)(64 << 20)) {
            // the header line, parsed as the host tokenizer parses it (name up to the first white space, then the comment)
            std::string head;
            bool plain = false;
            if (FILE *f = fopen(path.c_str(), "rb")) {
                head.resize(1 << 20);
                head.resize(fread(&head[0], 1, head.size(), f));
                fclose(f);
                plain = head.size() >= 2 && !((unsigned char)head[0] == 0x1f && (unsigned char)head[1] == 0x8b);
            }
            size_t p0 = 0;
            while (p0 < head.size() && head[p0] != '>' && head[p0] != '@') ++p0;
            const size_t eol = head.find('\n', p0);
            if (plain && p0 < head.size() && head[p0] == '>' && eol != std::string::npos) {
                const auto hrec = kcamd::parse_records(head.substr(p0, eol - p0 + 1));
                if (hrec.size() == 1) {
                    std::string hdr = ">" + hrec[0].name + " reoptimized=" + algorithm;   // ReprintSequenceHeader (masks.h:27-37)
                    if (!hrec[0].comment.empty()) hdr += " " + hrec[0].comment;
                    hdr += "\n";
                    kcamd::warm_context();
                    MappedOutput om;
                    if (!out_path.empty()) om.prepare(path, out_path, hdr.size(), trace, t_start);
                    kc_ctx *ctx = kcamd::context();
                    mark("device context ready");
                    void *d_seq = nullptr;
                    uint64_t d_len = 0;
                    int is_fastq = 0;
                    int rc = kc_read_fasta_device(ctx, path.c_str(), &d_seq, &d_len, &is_fastq);
                    kc_result *res = nullptr;
                    if (rc == KC_OK && !is_fastq) {
                        mark("input on the device");
                        rc = kc_maskopt_device(ctx, d_seq, d_len, k, complements, algorithm == "min-one", &res);
                        kc_device_free(ctx, d_seq);
                        mark("kc_maskopt_device returned");
                        if (res) {
                            uint64_t r_off = 0, r_len = 0;
                            kc_result_slice(res, &r_off, &r_len);   // (the result's own length: the tokeniser's leading separator is not part of it)
                            d_len = r_len;
                            if (om.ok && hdr.size() + d_len + 1 <= om.cap) om.commit(res, hdr, d_len, 0, mark);
                            else {
                                if (om.fd >= 0) om.abandon();
                                *of << hdr;
                                of->flush();
                                int fd = out_path.empty() ? 1 : ::open(out_path.c_str(), O_WRONLY | O_APPEND);
                                if (fd < 0) throw std::runtime_error("cannot reopen " + out_path);
                                kcamd::check(kc_result_write_fd(res, fd, 0));
                                write_all(fd, "\n", 1);
                                if (!out_path.empty()) ::close(fd);
                            }
                            kc_result_destroy(res);
                            mark("result written");
                            if (rc == KC_E_INVALID_CHAR) throw std::invalid_argument("Masked superstring contains invalid characters.");   // masks.h:75-77
                            kcamd::check(rc);
                            WriteLog("Finished optimization.");
                            return 0;
                        }
                        // (several records, or an error before anything was produced: the host path below decides what to say)
                    } else if (rc == KC_OK) { /* FASTQ-looking input: host parser */ }
                    if (om.fd >= 0) om.abandon();
                }
            }
        }
    }
    auto recs = kcamd::read_records(path);
    if (recs.empty()) throw std::invalid_argument("couldn't read a masked superstring from " + path);
    const kcamd::Record &ms = recs[0];
    std::string out(ms.seq.size(), 'N');
    int rc = kc_maskopt_ones(kcamd::context(), ms.seq.data(), ms.seq.size(), k, complements, algorithm == "min-one", out.data());
    // ReprintSequenceHeader (masks.h:27-37)
    *of << ">" << ms.name << " reoptimized=" << algorithm;
    if (!ms.comment.empty()) *of << " " << ms.comment;
    *of << std::endl;
    if (rc == KC_OK || rc == KC_E_INVALID_CHAR) *of << out << std::endl;
    if (rc == KC_E_INVALID_CHAR) throw std::invalid_argument("Masked superstring contains invalid characters.");   // masks.h:75-77
    kcamd::check(rc);
    if (recs.size() > 1) throw std::invalid_argument("Expecting only a single FASTA record -- the masked superstring.");   // masks.h:258
    WriteLog("Finished optimization.");
    return 0;
}

// ms2mssep (main.cpp:445-504).  As in the reference, `-s` alone does not count as a redirection: at least `-m` is needed.
int camel_split_ms(int argc, char **argv) {
    std::string subcommand = "ms2mssep", path;
    if (argc > 1 && std::string(argv[argc - 1]) != "-h") { path = argv[argc - 1]; argc--; }
    bool mask_set = false;
    std::ofstream mask, superstring;
    std::ostream *maskf = &std::cout, *superstringf = &std::cout;
    int opt;
    while ((opt = getopt(argc, argv, "m:s:h")) != -1) {
        switch (opt) {
            case 'm':
                if (mask_set) { std::cerr << "Error: -m parameter provided multiple times" << std::endl; return usage_subcommand(subcommand); }
                mask_set = true; mask.open(optarg); maskf = &mask; break;
            case 's': superstring.open(optarg); superstringf = &superstring; break;
            case 'h': usage_subcommand(subcommand); return 0;
            default: return usage_subcommand(subcommand);
        }
    }
    auto fail = [&](const std::string &m) { std::cerr << m << std::endl; return usage_subcommand(subcommand); };
    if (path.empty()) return fail("Required positional parameter path to the file not set.");
    if (!mask_set) return fail("Cannot have both superstring and mask redirected to stdout.");
    WriteLog("Started splitting masked superstring '" + path + "'.");
    split_ms(*superstringf, *maskf, path);
    WriteLog("Finished masked superstring splitting.");
    return 0;
}

// mssep2ms (main.cpp:506-568)
int camel_join_ms(int argc, char **argv) {
    std::string subcommand = "mssep2ms";
    bool mask_set = false, superstring_set = false;
    std::ifstream mask, superstring;
    std::istream *maskf = &std::cin, *superstringf = &std::cin;
    std::ofstream output;
    std::ostream *of = &std::cout;
    int opt;
    while ((opt = getopt(argc, argv, "m:s:o:h")) != -1) {
        switch (opt) {
            case 'm':
                if (mask_set) { std::cerr << "Error: -m parameter provided multiple times" << std::endl; return usage_subcommand(subcommand); }
                mask_set = true; mask.open(optarg); maskf = &mask; break;
            case 's':
                if (superstring_set) { std::cerr << "Error: -s parameter provided multiple times" << std::endl; return usage_subcommand(subcommand); }
                superstring_set = true; superstring.open(optarg); superstringf = &superstring; break;
            case 'o': output.open(optarg); of = &output; break;
            case 'h': usage_subcommand(subcommand); return 0;
            default: return usage_subcommand(subcommand);
        }
    }
    if (!mask_set && !superstring_set) {
        std::cerr << "Cannot have both superstring and mask redirected from stdin." << std::endl;
        return usage_subcommand(subcommand);
    }
    WriteLog("Started masked superstring joining.");
    join_ms(*superstringf, *maskf, *of);
    WriteLog("Finished masked superstring joining.");
    return 0;
}

// ms2spss / spss2ms (main.cpp:570-667)
int camel_spss(int argc, char **argv, bool to_spss) {
    std::string subcommand = to_spss ? "ms2spss" : "spss2ms", path;
    if (argc > 1 && std::string(argv[argc - 1]) != "-h") { path = argv[argc - 1]; argc--; }
    std::ofstream output;
    std::ostream *of = &std::cout;
    int k = 0, opt;
    try {
        while ((opt = getopt(argc, argv, "o:k:h")) != -1) {
            switch (opt) {
                case 'o': output.open(optarg); of = &output; break;
                case 'k': k = std::stoi(optarg); break;
                case 'h': usage_subcommand(subcommand); return 0;
                default: return usage_subcommand(subcommand);
            }
        }
    } catch (std::invalid_argument &) {
        return usage_subcommand(subcommand);
    }
    auto fail = [&](const std::string &m) { std::cerr << m << std::endl; return usage_subcommand(subcommand); };
    if (path.empty()) return fail("Required positional parameter path to the file not set.");
    if (k == 0) return fail("Required parameter k not set.");
    if (k < 0) return fail("k must be positive.");
    if (to_spss) {
        WriteLog("Started rSPSS computation from masked supertring '" + path + "'.");
        ms_to_spss(path, *of, k);
        WriteLog("Finished computing a rSPSS representing the same set.");
    } else {
        WriteLog("Started masked superstring computation corresponding to (r)SPSS '" + path + "'.");
        spss_to_ms(path, *of, k);
        WriteLog("Finished computing a masked superstring corresponding to the (r)SPSS.");
    }
    return 0;
}

}  // namespace

static void trace_epoch(const char *what) {
    if (!std::getenv("KC_CLI_TRACE")) return;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    char buf[96];
    snprintf(buf, sizeof buf, "[cli] epoch %s %.6f\n", what, (double)ts.tv_sec + ts.tv_nsec * 1e-9);
    std::cerr << buf;
}

int main(int argc, char **argv) {
    trace_epoch("main");
    if (argc == 1 || std::string(argv[1]) == "-h") { usage(); return 0; }
    std::string cmd = argv[1];
    if (cmd == "-v" || cmd == "--version") { std::cerr << KC_CLI_VERSION << std::endl; return 0; }
    // the device-backed subcommands leave through done(): all output files are closed by then, and tearing the HIP runtime
    // down at exit costs ~0.1 s that a command-line run has no use for
    // (KC_CLI_NO_FAST_EXIT: profilers write their results from exit handlers)
    auto done = [](int rc) -> int {
        if (std::getenv("KC_CLI_DESTROY_AT_EXIT")) { trace_epoch("before destroy"); kc_ctx_destroy(kcamd::context()); }   // (experiment: explicit hipFree of the pool)
        trace_epoch("exit");
        std::cout.flush(); std::cerr.flush();
        if (!std::getenv("KC_CLI_NO_FAST_EXIT")) _exit(rc);
        return rc;
    };
    if (cmd == "compute") return done(camel_compute(argc - 1, argv + 1, false));
    if (cmd == "lowerbound") return done(camel_compute(argc - 1, argv + 1, true));
    if (cmd == "maskopt") return done(camel_optimize(argc - 1, argv + 1));
    if (cmd == "ms2mssep") return camel_split_ms(argc - 1, argv + 1);
    if (cmd == "mssep2ms") return camel_join_ms(argc - 1, argv + 1);
    if (cmd == "ms2spss") return camel_spss(argc - 1, argv + 1, true);
    if (cmd == "spss2ms") return camel_spss(argc - 1, argv + 1, false);
    std::cerr << "Command '" << argv[1] << "' not recognized." << std::endl;
    return usage();
}
