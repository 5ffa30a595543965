// ctucopy -- command-line front end of the MI355X feature engine.
//
// Keeps the reference's command line (src/io/opts.cc:644-846), `-S` list format
// (`fin fout [spk] [vadfile]`, src/io/batch.cc:349-356), input decoders (raw / a-law / mu-law / WAVE,
// src/io/in.cc:434-619, src/io/amulaw.h:20-53) and feature writers (HTK src/io/out.cc:115-213, KALDI ark+scp
// :648-781, ICSI pfile src/io/pfile.cc:435-592), so that it is a drop-in for batch feature extraction.
// The per-frame chain itself runs on the GPU(s) behind include/ctu_engine.h; this file only moves bytes.
//
// Host loop (the counterpart of BATCH::process, src/io/batch.cc:326-421), three stages that overlap batch by batch:
//   reader   - sizes the next files, shards them over the GPUs by length (LPT; no collective: every utterance is
//              independent), lays each shard out with ctu_arena_layout and reads / decodes the files straight into
//              page-locked arenas with a pool of I/O threads;
//   engines  - one thread per GPU: ctu_plan_create + ctu_engine_run_host on its shard;
//   writer   - HTK / raw / WAVE / VAD files by the same pool straight from the page-locked rows, ark / pfile and the
//              verbose lines in list order.
// Files before a failing one are written as far as their batch got through, then the reference's message and exit
// status -1 (src/main.cpp:54-60).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <fstream>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <sys/stat.h>

#include "g711.h"
#include "ctu_engine.h"
#include "opts.h"

namespace {

struct Fatal : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Item {
    std::string fin, fout, spk, fvad;
};

// ---------------------------------------------------------------- decoders
uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

struct FileCloser {
    void operator()(FILE *f) const {
        if (f) std::fclose(f);
    }
};
typedef std::unique_ptr<FILE, FileCloser> File;

// canonical 44-byte WAVE header only, like src/io/in.cc:550-597; returns the number of samples the file holds
size_t wave_samples(const ctu::Opts &o, FILE *f, size_t file_bytes) {
    uint8_t b[44];
    if (file_bytes < 44 || std::fread(b, 1, 44, f) != 44 || std::memcmp(b, "RIFF", 4)) throw Fatal("IN: No RIFF header in file!");
    if (std::memcmp(b + 8, "WAVE", 4)) throw Fatal("IN: Not a WAVE file!");
    if (rd16(&b[20]) != 1) throw Fatal("IN: Not a PCM WAVE file!");
    if ((long)rd32(&b[24]) != o.fs) throw Fatal("IN: WAVE file reports different sampling rate than specified!");
    if (rd16(&b[22]) != 1) throw Fatal("IN: Input WAVE file is not mono!");
    if (rd16(&b[34]) != 16) throw Fatal("IN: Not 16 bits per sample!");
    return std::min<size_t>(rd32(&b[40]) / 2, (file_bytes - 44) / 2);
}

// number of samples `path` will decode to, without reading its data
int64_t probe_samples(const ctu::Opts &o, const std::string &path) {
    const bool wave = o.format_in == "wave";
    if (o.format_in != "raw" && o.format_in != "alaw" && o.format_in != "mulaw" && !wave) throw Fatal("IN: Unknown input file format!");
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) {
        // not a plain file (or missing): let fopen decide, sizes by reading to the end
        File f(std::fopen(path.c_str(), "rb"));
        if (!f) throw Fatal(wave ? "IN: Cannot open file!" : "IN: Cannot open data file!");
        size_t n = 0, got;
        uint8_t tmp[1 << 16];
        while ((got = std::fread(tmp, 1, sizeof tmp, f.get())) > 0) n += got;
        if (wave) {
            std::rewind(f.get());
            return (int64_t)wave_samples(o, f.get(), n);
        }
        return (int64_t)(o.format_in == "raw" ? n / 2 : n);
    }
    const size_t bytes = (size_t)st.st_size;
    if (wave) {
        File f(std::fopen(path.c_str(), "rb"));
        if (!f) throw Fatal("IN: Cannot open file!");
        return (int64_t)wave_samples(o, f.get(), bytes);
    }
    return (int64_t)(o.format_in == "raw" ? bytes / 2 : bytes);
}

// reads `n` samples of `path` into dst (raw / a-law / mu-law / WAVE, src/io/in.cc:434-619, src/io/amulaw.h:20-53)
void decode_into(const ctu::Opts &o, const std::string &path, int16_t *dst, size_t n) {
    const bool wave = o.format_in == "wave";
    File f(std::fopen(path.c_str(), "rb"));
    if (!f) throw Fatal(wave ? "IN: Cannot open file!" : "IN: Cannot open data file!");
    if (wave && std::fseek(f.get(), 44, SEEK_SET) != 0) throw Fatal("IN: No RIFF header in file!");
    uint8_t *bytes = reinterpret_cast<uint8_t *>(dst);
    if (o.format_in == "alaw" || o.format_in == "mulaw") {
        // the codes go to the upper half of the samples' own bytes and are expanded from the front (code i sits at byte n + i >= 2 i + 1)
        if (std::fread(bytes + n, 1, n, f.get()) != n) throw Fatal("IN: Cannot read data file!");
        const bool alaw = o.format_in == "alaw";
        for (size_t i = 0; i < n; i++) dst[i] = g711_to_linear(bytes[n + i], alaw);
        return;
    }
    if (std::fread(bytes, 2, n, f.get()) != n) throw Fatal("IN: Cannot read data file!");
    if (o.swap_in && !wave)
        for (size_t i = 0; i < n; i++) dst[i] = (int16_t)((uint16_t)dst[i] << 8 | (uint16_t)dst[i] >> 8);
}

// ---------------------------------------------------------------- writers
void put32(std::vector<uint8_t> &v, uint32_t x, bool big) {
    for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (big ? 24 - 8 * i : 8 * i)));
}
void put16(std::vector<uint8_t> &v, uint16_t x, bool big) {
    for (int i = 0; i < 2; i++) v.push_back((uint8_t)(x >> (big ? 8 - 8 * i : 8 * i)));
}
void putf(std::vector<uint8_t> &v, float f, bool big) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    put32(v, x, big);
}

void write_file(const std::string &path, const std::vector<uint8_t> &bytes, const char *err) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw Fatal(err);
    if (!bytes.empty() && std::fwrite(bytes.data(), 1, bytes.size(), f) != bytes.size()) {
        std::fclose(f);
        throw Fatal("OUT: Error in stream writing!");
    }
    std::fclose(f);
}

// HTK: nSamples, sampPeriod (100 ns), sampSize (bytes), parmKind, then float32 rows (src/io/out.cc:115-213)
void write_htk(const std::string &path, const float *rows, int64_t n, const ctu_dims &d) {
    std::vector<uint8_t> h;
    const bool big = d.swap_out;
    put32(h, (uint32_t)n, big);
    put32(h, d.htk_period, big);
    put16(h, (uint16_t)(4 * d.row_floats), big);
    put16(h, (uint16_t)d.htk_kind, big);
    File f(std::fopen(path.c_str(), "wb"));
    if (!f) throw Fatal("OUT: Cannot create output file!");
    const size_t nf = (size_t)n * d.row_floats;
    bool ok = std::fwrite(h.data(), 1, h.size(), f.get()) == h.size();
    if (!big) ok = ok && (nf == 0 || std::fwrite(rows, 4, nf, f.get()) == nf);  // rows are little-endian float32 as they stand
    else {
        std::vector<uint32_t> sw(nf);
        for (size_t i = 0; i < nf; i++) {
            uint32_t x;
            std::memcpy(&x, rows + i, 4);
            sw[i] = __builtin_bswap32(x);
        }
        ok = ok && (nf == 0 || std::fwrite(sw.data(), 4, nf, f.get()) == nf);
    }
    if (!ok) throw Fatal("OUT: Error in stream writing!");
}

// KALDI binary matrix archive + index (src/io/out.cc:648-781)
struct ArkWriter {
    FILE *ark = nullptr, *scp = nullptr;
    std::string arkname;
    explicit ArkWriter(const std::string &name) : arkname(name) {
        ark = std::fopen(name.c_str(), "wb");
        if (!ark) throw Fatal("OUT: Cannot create output ark file!");
        // "x.ark" -> "x.scp": text up to the first ".ark" component, as rename_path_ark_to_scp does
        std::string scpname, rest = name;
        bool found = false;
        std::stringstream ss(name);
        std::string tok;
        while (std::getline(ss, tok, '.')) {
            if (tok.empty()) continue;
            if (tok == "ark") {
                found = true;
                break;
            }
            scpname += tok + ".";
        }
        (void)found;
        scpname += "scp";
        scp = std::fopen(scpname.c_str(), "wt");
        if (!scp) throw Fatal("Cannot open output scp file for writing!");
    }
    void add(const std::string &key, const float *rows, int64_t n, int cols) {
        std::fprintf(ark, "%s %cBFM %c", key.c_str(), 0, 4);
        const long long idx = (long long)ftello(ark) - 6;  // offset of the \0 that starts the binary marker
        const int32_t r = (int32_t)n, c = cols;
        std::fwrite(&r, 4, 1, ark);
        std::fputc(4, ark);
        std::fwrite(&c, 4, 1, ark);
        std::fprintf(scp, "%s %s:%lld\n", key.c_str(), arkname.c_str(), idx);
        if (n) std::fwrite(rows, 4, (size_t)n * cols, ark);
    }
    ~ArkWriter() {
        if (ark) std::fclose(ark);
        if (scp) std::fclose(scp);
    }
};

// ICSI pfile: 32768-byte ASCII header, big-endian rows [sent, frame, features], sentence index table
// (src/io/pfile.cc:435-468,470-505,573-592).  The reference opens it with the internal vector width, not the
// written row width (src/io/out.cc:252); that is reproduced: nfea_pf floats per row, zero padded / truncated.
struct PfileWriter {
    std::string name;
    int nfea;
    std::vector<uint8_t> data;
    std::vector<uint32_t> sent_start{0};
    uint32_t nframes = 0;
    PfileWriter(const std::string &n, int nf) : name(n), nfea(nf) {}
    void add(const float *rows, int64_t n, int cols) {
        const uint32_t sid = (uint32_t)sent_start.size() - 1;
        for (int64_t t = 0; t < n; t++) {
            put32(data, sid, true);
            put32(data, (uint32_t)t, true);
            for (int i = 0; i < nfea; i++) putf(data, i < cols ? rows[t * cols + i] : 0.f, true);
        }
        nframes += (uint32_t)n;
        sent_start.push_back(nframes);
    }
    void close() {
        const unsigned long long hsize = 32768, dsize = (unsigned long long)(nfea + 2) * nframes;
        std::string h;
        char line[256];
        auto addf = [&](const char *fmt, auto... a) {
            std::snprintf(line, sizeof line, fmt, a...);
            h += line;
        };
        addf("-pfile_header version %u size %llu\n", 0u, hsize);
        addf("-num_sentences %u\n", (unsigned)sent_start.size() - 1);
        addf("-num_frames %u\n", nframes);
        addf("-first_feature_column %u\n", 2u);
        addf("-num_features %u\n", (unsigned)nfea);
        addf("-first_label_column %u\n", (unsigned)(2 + nfea));
        addf("-num_labels %u\n", 0u);
        h += "-format dd" + std::string(nfea, 'f') + "\n";
        addf("-data size %llu offset %llu ndim %u nrow %u ncol %u\n", dsize, 0ull, 2u, nframes, (unsigned)(nfea + 2));
        addf("-sent_table_data size %llu offset %llu ndim %u\n", (unsigned long long)sent_start.size(), dsize, 1u);
        h += "-end\n";
        std::vector<uint8_t> b(h.begin(), h.end());
        b.resize(hsize, 0);
        b.insert(b.end(), data.begin(), data.end());
        for (uint32_t s : sent_start) put32(b, s, true);
        write_file(name, b, "OUT: Cannot create output file!");
    }
};

// ---------------------------------------------------------------- GPU side
struct Gpu {
    ctu_engine *eng = nullptr;
    ~Gpu() {
        if (eng) ctu_engine_destroy(eng);
    }
};

// rawOUT::write (src/io/out.cc:493-499): int16 samples, byte-swapped for -endian_out big
void write_raw(const std::string &path, const int16_t *x, size_t n, bool big) {
    File f(std::fopen(path.c_str(), "wb"));
    if (!f) throw Fatal("OUT: Cannot open output stream!");
    bool ok;
    if (!big) ok = n == 0 || std::fwrite(x, 2, n, f.get()) == n;
    else {
        std::vector<uint16_t> sw(n);
        for (size_t i = 0; i < n; i++) sw[i] = __builtin_bswap16((uint16_t)x[i]);
        ok = n == 0 || std::fwrite(sw.data(), 2, n, f.get()) == n;
    }
    if (!ok) throw Fatal("OUT: Error in stream writing!");
}

// waveOUT (src/io/out.cc:517-564): canonical 44-byte RIFF header, sizes patched at close, samples in host order
void write_wave(const std::string &path, const int16_t *x, size_t n, int fs) {
    std::vector<uint8_t> b;
    const uint32_t data = (uint32_t)(2 * n);
    for (char c : std::string("RIFF")) b.push_back((uint8_t)c);
    put32(b, data + 36, false);
    for (char c : std::string("WAVEfmt ")) b.push_back((uint8_t)c);
    put32(b, 16, false);
    put16(b, 1, false);
    put16(b, 1, false);
    put32(b, (uint32_t)fs, false);
    put32(b, (uint32_t)fs * 2, false);
    put16(b, 2, false);
    put16(b, 16, false);
    for (char c : std::string("data")) b.push_back((uint8_t)c);
    put32(b, data, false);
    File f(std::fopen(path.c_str(), "wb"));
    if (!f) throw Fatal("OUT: Cannot open data file!");
    if (std::fwrite(b.data(), 1, b.size(), f.get()) != b.size() || (n && std::fwrite(x, 2, n, f.get()) != n)) throw Fatal("OUT: Error in stream writing!");
}

// ---------------------------------------------------------------- the pipeline's plumbing
// bounded hand-over between two stages; close() wakes everybody up (end of the list, or a stage gave up)
template <class T>
struct Chan {
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> q;
    size_t cap;
    bool closed = false;
    explicit Chan(size_t c) : cap(c) {}
    bool push(T v) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return q.size() < cap || closed; });
        if (closed) return false;
        q.push_back(std::move(v));
        cv.notify_all();
        return true;
    }
    bool pop(T &v) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        v = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return true;
    }
    void close() {
        std::lock_guard<std::mutex> l(m);
        closed = true;
        cv.notify_all();
    }
};

// fn(i) for i in [0, n) on up to `threads` threads; the exception of the lowest failing index is rethrown (what a
// sequential loop would have hit first)
template <class F>
void parallel_for(int threads, size_t n, F fn) {
    if (n == 0) return;
    const int nt = (int)std::min<size_t>((size_t)std::max(threads, 1), n);
    std::atomic<size_t> next{0};
    std::mutex em;
    size_t err_at = n;
    std::exception_ptr err;
    auto body = [&] {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) return;
            try {
                fn(i);
            } catch (...) {
                std::lock_guard<std::mutex> l(em);
                if (i < err_at) {
                    err_at = i;
                    err = std::current_exception();
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(body);
    body();
    for (auto &t : th) t.join();
    if (err) std::rethrow_exception(err);
}

// page-locked buffers are expensive to make (the runtime pins every page): they go round between the batches
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
};
struct PinPool {
    std::mutex m;
    std::vector<PinBuf> idle;
    PinBuf get(size_t bytes) {
        bytes = std::max<size_t>(bytes, 4096);
        {
            std::lock_guard<std::mutex> l(m);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].cap >= bytes && (best == idle.size() || idle[i].cap < idle[best].cap)) best = i;
            if (best < idle.size()) {
                PinBuf b = idle[best];
                idle.erase(idle.begin() + best);
                return b;
            }
            if (!idle.empty()) {  // nothing fits: trade the smallest idle one in
                size_t sm = 0;
                for (size_t i = 1; i < idle.size(); i++)
                    if (idle[i].cap < idle[sm].cap) sm = i;
                ctu_host_free(idle[sm].p);
                idle.erase(idle.begin() + sm);
            }
        }
        PinBuf b;
        b.cap = bytes + bytes / 8;
        b.p = ctu_host_alloc(b.cap);
        if (!b.p) throw Fatal("ENGINE: cannot allocate page-locked host memory");
        return b;
    }
    void put(PinBuf b) {
        if (!b.p) return;
        std::lock_guard<std::mutex> l(m);
        idle.push_back(b);
    }
    ~PinPool() {
        for (auto &b : idle) ctu_host_free(b.p);
    }
};

// one GPU's part of a batch
struct Shard {
    std::vector<size_t> idx;      // positions in the batch, list order
    std::vector<int64_t> ns, so;  // samples per utterance, arena offsets (ctu_arena_layout)
    std::vector<int32_t> hidx;    // VAD: the majority filter's ring index every utterance starts with (the list is one process's)
    int64_t total_samples = 0, total_frames = 0;
    PinBuf arena, rows;           // int16 in; float32 rows (or int16 samples with -format_out raw|wave) out
    std::vector<int64_t> ro, kept, nout;
    std::vector<uint8_t> vad;
};
struct Batch {
    size_t pos = 0, n = 0;                        // items [pos, pos + n)
    std::vector<Shard> sh;                        // one per GPU
    std::vector<std::pair<int, size_t>> where;    // item -> (gpu, position in its shard)
    std::exception_ptr err;                       // the reader failed on this batch: rethrown once the earlier ones are through
};

// the engine's part of a shard: plan over its lengths, H2D + kernels + D2H from / to the page-locked buffers
void run_shard(ctu_engine *eng, Shard &sh, PinPool &pool, bool signal_out, bool has_vad, int row_floats) {
    ctu_plan *plan = nullptr;
    if (ctu_plan_create(eng, sh.ns.data(), (int)sh.ns.size(), &plan) != CTU_OK) throw Fatal(ctu_last_error(eng));
    struct PlanGuard {
        ctu_plan *p;
        ~PlanGuard() { ctu_plan_destroy(p); }
    } guard{plan};
    const size_t n = sh.ns.size();
    const int64_t *so = ctu_plan_sample_offsets(plan), *ro = ctu_plan_row_offsets(plan);
    if (ctu_plan_total_samples(plan) != sh.total_samples || !std::equal(sh.so.begin(), sh.so.end(), so))
        throw Fatal("ENGINE: internal: the plan's arena layout differs from ctu_arena_layout");
    if (!sh.hidx.empty() && ctu_plan_set_vad_ring(plan, sh.hidx.data()) != CTU_OK) throw Fatal(ctu_last_error(eng));
    sh.total_frames = ctu_plan_total_frames(plan);
    sh.ro.assign(ro, ro + n + 1);
    sh.kept.assign(n, 0);
    if (signal_out) {  // -format_out raw|wave: samples at the utterances' own offsets (ctu_engine_run_signal_host)
        sh.rows = pool.get((size_t)sh.total_samples * sizeof(int16_t));
        const int64_t *no = ctu_plan_out_samples(plan);
        sh.nout.assign(no, no + n);
        if (ctu_engine_run_signal_host(eng, plan, static_cast<const int16_t *>(sh.arena.p), static_cast<int16_t *>(sh.rows.p)) != CTU_OK)
            throw Fatal(ctu_last_error(eng));
        return;
    }
    sh.rows = pool.get((size_t)sh.total_frames * row_floats * sizeof(float));
    sh.vad.assign(has_vad ? (size_t)sh.total_frames : 0, 0);
    // rows_per_utt < frames only with -vad_apply_mode drop (rows compacted in place by the library)
    if (ctu_engine_run_host(eng, plan, static_cast<const int16_t *>(sh.arena.p), static_cast<float *>(sh.rows.p), has_vad ? sh.vad.data() : nullptr,
                            sh.kept.data()) != CTU_OK)
        throw Fatal(ctu_last_error(eng));
}

int real_main(int argc, char **argv) {
    std::vector<std::string> args;
    int ngpu = 1;
    std::vector<int> gpu_map;  // --gpu-map a,b,...: device ordinal of every engine (default 0 .. N-1); an ordinal may repeat, which puts
                               // several engines on one device - the multi-engine host path rehearsed on a box with fewer GPUs
    int io_threads = 0;     // --io-threads N: file readers (default: the hardware threads, at most 16)
    int write_threads = 0;  // --write-threads N: file writers (default 1: creating files in one directory does not scale)
    int batch_mib = 256;    // --batch-mib M: PCM per batch
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) ngpu = std::max(1, std::atoi(argv[++i]));
        else if (!std::strcmp(argv[i], "--io-threads") && i + 1 < argc) io_threads = std::max(1, std::atoi(argv[++i]));
        else if (!std::strcmp(argv[i], "--write-threads") && i + 1 < argc) write_threads = std::max(1, std::atoi(argv[++i]));
        else if (!std::strcmp(argv[i], "--batch-mib") && i + 1 < argc) batch_mib = std::max(1, std::atoi(argv[++i]));
        else if (!std::strcmp(argv[i], "--gpu-map") && i + 1 < argc) {
            std::istringstream ms(argv[++i]);
            std::string tok;
            while (std::getline(ms, tok, ',')) gpu_map.push_back(std::atoi(tok.c_str()));
        }
        else args.emplace_back(argv[i]);
    }
    if (io_threads == 0) {
        const char *ev = std::getenv("CTU_IO_THREADS");
        io_threads = ev ? std::max(1, std::atoi(ev)) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    }
    if (write_threads == 0) {
        const char *ev = std::getenv("CTU_WRITE_THREADS");
        write_threads = ev ? std::max(1, std::atoi(ev)) : 1;
    }
    if (!gpu_map.empty() && (int)gpu_map.size() != ngpu) throw Fatal("ENGINE: --gpu-map needs one device ordinal per engine of --gpus");
    ctu::Opts o;
    try {
        o = ctu::Opts::from_args(args);
    } catch (const ctu::OptsError &e) {
        if (args.empty()) std::fputs(ctu::Opts().usage().c_str(), stderr);
        throw Fatal(e.what());
    }
    if (o.help) {
        std::fputs(o.usage().c_str(), stdout);
        return 0;
    }
    const bool signal_out = o.format_out == "raw" || o.format_out == "wave";
    if (!signal_out && o.format_out != "htk" && o.format_out != "ark" && o.format_out != "pfile")
        throw Fatal("OUT: Unknown output file format!");
    // the work list
    std::vector<Item> items;
    if (o.pipe_in || o.pipe_out) throw Fatal("ENGINE: online (pipe) mode is not supported");
    if (!o.in.empty()) {
        items.push_back({o.in, o.out, "", o.vad_out});
    } else {
        if (o.list.empty()) throw Fatal("BATCH: Nothing to do!");
        std::ifstream lf(o.list);
        if (!lf) throw Fatal("BATCH: Cannot open list file!");
        std::string line;
        while (std::getline(lf, line)) {
            std::istringstream ss(line);
            Item it;
            if (!(ss >> it.fin >> it.fout)) {
                if (line.find_first_not_of(" \t\r") == std::string::npos) continue;
                throw Fatal("BATCH: Bad list format!");
            }
            ss >> it.spk >> it.fvad;
            if (o.do_vad() && it.fvad.empty()) throw Fatal("BATCH: Bad list format!");
            // CMVN lists (src/io/batch.cc:358-367): "<in> <speaker>" when only statistics are wanted,
            // "<in> <out> <speaker>" when they are applied
            if (o.apply_cmvn && it.spk.empty()) throw Fatal(" Bad list format for applying cmvn!");
            if (o.stat_cmvn && it.spk.empty()) it.spk = it.fout;
            items.push_back(it);
        }
    }
    // hwss / fwss / 2fwss chain the list through the noise seed (src/nr/nr.cc:212-221): one GPU, files in list order
    const bool chained = o.nr_mode == "hwss" || o.nr_mode == "fwss" || o.nr_mode == "2fwss";
    if (chained && ngpu > 1) {
        if (o.verbose) std::fprintf(stderr, "ENGINE: -nr_mode %s chains the files of the list: running on one GPU\n", o.nr_mode.c_str());
        ngpu = 1;
    }
    // engines, one per GPU
    std::vector<const char *> cargs;
    for (auto &a : args) cargs.push_back(a.c_str());
    // CMVN together with the VAD (src/io/batch.cc:193-204,230-241): the statistics are taken over every frame (the passes that sum do not
    // call save_frame, so the VAD does not see them), the last pass normalises a vector and THEN hands it to save_frame - the VAD's ring,
    // its decision, the drop.  The engine therefore delivers every row (apply mode none) and this loop does the VAD's part on the
    // normalised rows: the ring's phase along the list, the rows a short file does not write, the dropped frames.
    const bool cmvn_vad = (o.stat_cmvn || o.apply_cmvn) && o.do_vad();
    if (cmvn_vad) {
        cargs.push_back("-vad_apply_mode");
        cargs.push_back("none");
    }
    std::vector<Gpu> gpus(ngpu);
    for (int g = 0; g < ngpu; g++)
        if (ctu_engine_create((int)cargs.size(), cargs.data(), gpu_map.empty() ? g : gpu_map[g], &gpus[g].eng) != CTU_OK) throw Fatal(ctu_create_error());
    ctu_dims d;
    ctu_engine_dims(gpus[0].eng, &d);

    // -vad file=<f> with hwss / fwss / 2fwss: `char vad = fgetc(fvad); if (vad != EOF) ... else throw` (src/nr/nr.cc:297-302) - the reference
    // runs out of decisions at some FRAME, after every file in front of that one has been written.  The engine takes a batch whole, so
    // the list is cut in front of the file whose frames pass the end of the stream (or its first 0xFF byte, which the signed char
    // compares equal to EOF) and the reference's message is raised once the files before it are out.
    std::string deferred_error;
    if (chained && o.vadmode == "file") {
        std::ifstream vf(o.filevad, std::ios::binary);
        if (vf) {
            std::vector<unsigned char> bytes((std::istreambuf_iterator<char>(vf)), std::istreambuf_iterator<char>());
            int64_t usable = (int64_t)(std::find(bytes.begin(), bytes.end(), (unsigned char)0xFF) - bytes.begin()), used = 0;
            for (size_t i = 0; i < items.size(); i++) {
                int64_t T;
                try {
                    T = ctu_num_frames(gpus[0].eng, probe_samples(o, items[i].fin));
                } catch (...) {
                    break;  // the reader reports the unreadable file where the reference would
                }
                if (T < 0) break;
                if (used + T > usable) {
                    items.resize(i);
                    deferred_error = "NR: Unexpected end of VAD file!";
                    break;
                }
                used += T;
            }
        }
    }

    std::unique_ptr<ArkWriter> ark;
    std::unique_ptr<PfileWriter> pf;
    if (o.format_out == "ark") ark.reset(new ArkWriter(o.arkfilename));
    if (o.format_out == "pfile") {
        // internal vector width: row width without E, plus c0 when it is switched off for the row (src/io/out.cc:252)
        int nfea_pf = d.row_floats - (o.fea_E ? 1 : 0);
        if ((o.fea_kind == "dctc" || o.fea_kind == "lpc") && !o.fea_c0) nfea_pf += 1;
        if (o.fea_kind == "lpa") nfea_pf += 1;
        pf.reset(new PfileWriter(o.pfilename, nfea_pf));
    }

    // ---- per-speaker CMVN (src/io/batch.cc:131-171,331-419).  -apply_cmvn <f>: the reference first tries to read <f>;
    // its reader keeps "mean" as every speaker's name and only fea_ncepcoefs+1 values (src/io/in.cc:735-770), after
    // which add_spk allocates fresh all-zero statistics for each real speaker - the run divides by zero.  That path is
    // not reproduced.  When <f> does not exist the reference computes the statistics, writes them to <f> and applies
    // them in a third pass; -stat_cmvn <f> alone computes and writes them and produces no feature files.
    const bool cmvn = o.stat_cmvn || o.apply_cmvn;
    std::string stat_path = o.fcmvn_stat_out;
    if (o.apply_cmvn) {
        if (std::ifstream(o.fcmvn_stat_in).good())
            throw Fatal("ENGINE: applying an existing CMVN statistics file is not on the accelerated path (the reference's "
                        "reader loses the speaker names and the statistics, src/io/in.cc:735-770)");
        std::printf("IN: Cannot open stat. cmvn file!\nIN: Stat. cmvn file is being created: %s\n", o.fcmvn_stat_in.c_str());
        stat_path = o.fcmvn_stat_in;
    }
    std::vector<std::vector<float>> all_rows(cmvn ? items.size() : 0);
    std::vector<int64_t> all_ns(cmvn ? items.size() : 0);
    std::vector<std::vector<uint8_t>> all_vads(cmvn_vad ? items.size() : 0);

    // ---- the pipeline: reader -> engines (this thread) -> writer, one batch in each at a time
    const size_t batch_samples = (size_t)batch_mib << 19;  // samples of PCM per batch (default 1 GiB)
    PinPool pool;
    // CTU_HOST_TIMING=1: seconds each stage was busy (not waiting for its neighbours), on stderr at the end
    const bool timing = std::getenv("CTU_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_read = 0, t_engine = 0, t_write = 0;
    const double t_loop0 = now();
    Chan<std::unique_ptr<Batch>> to_engine(1), to_writer(1);
    std::atomic<bool> giving_up{false};

    // The reference's VAD keeps its majority filter for the whole list and cleanFilter() does not reset its ring index between files
    // (src/vad/vad.h:110-121): the rows of a file depend on the frame counts of the files in front of it (include/ctu_engine.h).
    const bool vad_ring = d.has_vad && o.vad_filter_order > 1 && !cmvn_vad;  // with CMVN the ring acts on the normalised rows: below
    std::thread reader([&] {
        size_t pos = 0;
        int32_t ring_hidx = 0, ring_hsize = 0;
        // sizes of the files probed so far (a group is probed ahead of the batch it fills: what the batch leaves is not probed again)
        std::vector<int64_t> probed;
        std::vector<std::exception_ptr> probe_err;
        while (pos < items.size() && !giving_up) {
            std::unique_ptr<Batch> b(new Batch);
            b->pos = pos;
            const double t0 = now();
            try {
                // sizes first (stat, WAVE headers), in groups, until the batch is full
                std::vector<int64_t> ns;
                size_t total = 0, end = pos;
                bool stop = false;
                while (!stop && end < items.size() && (total < batch_samples || end == pos)) {
                    if (end >= probed.size()) {
                        const size_t first = probed.size(), group = std::min<size_t>(items.size() - first, 1024);
                        probed.resize(first + group);
                        probe_err.resize(first + group);
                        parallel_for(io_threads, group, [&](size_t i) {
                            try {
                                probed[first + i] = probe_samples(o, items[first + i].fin);
                                if (ctu_num_frames(gpus[0].eng, probed[first + i]) < 0) throw Fatal("IO: Signal shorter than one frame!");
                            } catch (...) {
                                probe_err[first + i] = std::current_exception();
                            }
                        });
                    }
                    for (; end < probed.size() && (total < batch_samples || end == pos); end++) {
                        if (probe_err[end]) {  // a bad file ends the batch in front of it; it fails the batch it would start
                            if (end == pos) std::rethrow_exception(probe_err[end]);
                            stop = true;
                            break;
                        }
                        ns.push_back(probed[end]);
                        total += (size_t)probed[end];
                    }
                }
                const size_t n = b->n = end - pos;
                // longest-processing-time sharding over the GPUs by length; list order inside a shard
                std::vector<size_t> order(n);
                for (size_t i = 0; i < n; i++) order[i] = i;
                std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return ns[x] > ns[y]; });
                b->sh.resize(ngpu);
                std::vector<size_t> load(ngpu, 0);
                for (size_t i : order) {
                    const int g = (int)(std::min_element(load.begin(), load.end()) - load.begin());
                    b->sh[g].idx.push_back(i);
                    load[g] += (size_t)ns[i];
                }
                b->where.resize(n);
                std::vector<int32_t> hidx_of(n, 0);
                if (vad_ring)
                    for (size_t i = 0; i < n; i++) {  // list order
                        // At filter orders of 5 and more a file with no more frames than the filter's delay leaves the reference's
                        // historySize half drained (one flush_frame per unready file, src/vad/vad.h:156-175): the file behind it gets
                        // ready early and writes more rows and decisions than it has frames.  Not reproduced - and not passed over in silence.
                        if (ring_hsize != 0)
                            throw Fatal("VAD: " + items[pos + i - 1].fin + " has no more frames than the majority filter delays (-vad_filter_order " +
                                        std::to_string(o.vad_filter_order) + "): the reference's filter stays half drained for the files behind it "
                                        "(src/vad/vad.h:156-175), which is not reproduced");
                        hidx_of[i] = ring_hidx;
                        ctu_vad_ring_step(o.vad_filter_order, std::max<int64_t>(ctu_num_frames(gpus[0].eng, ns[i]), 0), &ring_hidx, &ring_hsize);
                    }
                for (int g = 0; g < ngpu; g++) {
                    Shard &sh = b->sh[g];
                    std::sort(sh.idx.begin(), sh.idx.end());
                    for (size_t k = 0; k < sh.idx.size(); k++) {
                        sh.ns.push_back(ns[sh.idx[k]]);
                        if (vad_ring) sh.hidx.push_back(hidx_of[sh.idx[k]]);
                        b->where[sh.idx[k]] = {g, k};
                    }
                    sh.so.resize(sh.ns.size() + 1);
                    sh.total_samples = ctu_arena_layout(sh.ns.data(), (int)sh.ns.size(), sh.so.data());
                    if (sh.idx.empty()) continue;
                    // the bytes between utterances are read under zero weights and need no particular value (include/ctu_engine.h)
                    sh.arena = pool.get((size_t)sh.total_samples * sizeof(int16_t));
                }
                parallel_for(io_threads, n, [&](size_t i) {
                    Shard &sh = b->sh[b->where[i].first];
                    const size_t k = b->where[i].second;
                    decode_into(o, items[pos + i].fin, static_cast<int16_t *>(sh.arena.p) + sh.so[k], (size_t)sh.ns[k]);
                });
                pos = end;
                t_read += now() - t0;
            } catch (...) {
                for (Shard &sh : b->sh)  // page-locked arenas already taken for this batch go back to the pool
                    if (sh.arena.p) {
                        pool.put(sh.arena);
                        sh.arena = PinBuf();
                    }
                b->err = std::current_exception();
                to_engine.push(std::move(b));
                break;
            }
            if (!to_engine.push(std::move(b))) break;
        }
        to_engine.close();
    });

    std::exception_ptr writer_err;
    std::thread writer([&] {
        std::unique_ptr<Batch> b;
        while (to_writer.pop(b)) {
            const double t0 = now();
            try {
                const size_t n = b->n;
                auto rows_of = [&](size_t i, int64_t &nr) {
                    const Shard &sh = b->sh[b->where[i].first];
                    const size_t k = b->where[i].second;
                    nr = sh.kept[k];
                    return static_cast<const float *>(sh.rows.p) + sh.ro[k] * d.row_floats;
                };
                if (o.verbose)
                    for (size_t i = 0; i < n; i++) {
                        const Shard &sh = b->sh[b->where[i].first];
                        const size_t k = b->where[i].second;
                        std::fprintf(stderr, "processing: %s - %lld frames.\n", items[b->pos + i].fin.c_str(),
                                     (long long)(signal_out ? sh.ro[k + 1] - sh.ro[k] : sh.kept[k]));
                    }
                if (signal_out) {  // speech enhancement: samples instead of rows (src/io/batch.cc:62-65,223-227)
                    parallel_for(write_threads, n, [&](size_t i) {
                        const Shard &sh = b->sh[b->where[i].first];
                        const size_t k = b->where[i].second;
                        const int16_t *x = static_cast<const int16_t *>(sh.rows.p) + sh.so[k];
                        if (o.format_out == "raw") write_raw(items[b->pos + i].fout, x, (size_t)sh.nout[k], d.swap_out != 0);
                        else write_wave(items[b->pos + i].fout, x, (size_t)sh.nout[k], o.fs);
                    });
                } else if (cmvn) {  // rows wait for the corpus statistics
                    for (size_t i = 0; i < n; i++) {
                        int64_t nr;
                        const float *r = rows_of(i, nr);
                        const Shard &sh = b->sh[b->where[i].first];
                        const size_t k = b->where[i].second;
                        if (cmvn_vad) {  // every frame's row counts for the statistics, also those of a file the VAD writes nothing for
                            nr = sh.ro[k + 1] - sh.ro[k];
                            all_vads[b->pos + i].assign(sh.vad.begin() + sh.ro[k], sh.vad.begin() + sh.ro[k + 1]);
                        }
                        all_rows[b->pos + i].assign(r, r + nr * d.row_floats);
                        all_ns[b->pos + i] = sh.ns[k];
                    }
                } else {
                    const bool vad_files = d.has_vad && o.vad_out_mode != "none";
                    if (vad_files)
                        for (size_t i = 0; i < n; i++)
                            if (items[b->pos + i].fvad.empty()) throw Fatal("VAD::new_file(): invalid filename!");
                    const bool per_file = !ark && !pf;
                    parallel_for(write_threads, n, [&](size_t i) {
                        const Item &it = items[b->pos + i];
                        const Shard &sh = b->sh[b->where[i].first];
                        const size_t k = b->where[i].second;
                        if (vad_files) {  // one ASCII '0'/'1' per frame (src/vad/vad.h:67-70); NUL = nothing written (include/ctu_engine.h)
                            std::vector<uint8_t> v;
                            for (int64_t t = sh.ro[k]; t < sh.ro[k + 1]; t++)
                                if (sh.vad[(size_t)t]) v.push_back(sh.vad[(size_t)t]);
                            write_file(it.fvad, v, "FileWriter: cannot open file!");
                        }
                        if (per_file) {
                            // -fea_trap: the reference's writers overwrite fea_kind with "spec" when they save their first frame
                            // (src/io/out.cc:182), so every header after the first file carries base kind 8 (out.cc:146-152).
                            ctu_dims dh = d;
                            if (o.fea_trap && b->pos + i > 0) dh.htk_kind = (d.htk_kind & ~077) | 8;
                            int64_t nr;
                            const float *r = rows_of(i, nr);
                            write_htk(it.fout, r, nr, dh);
                        }
                    });
                    if (!per_file)
                        for (size_t i = 0; i < n; i++) {
                            int64_t nr;
                            const float *r = rows_of(i, nr);
                            if (ark) ark->add(items[b->pos + i].fout, r, nr, d.row_floats);
                            else pf->add(r, nr, d.row_floats);
                        }
                }
            } catch (...) {
                writer_err = std::current_exception();
                giving_up = true;
            }
            for (auto &sh : b->sh) {
                pool.put(sh.arena);
                pool.put(sh.rows);
            }
            b.reset();
            t_write += now() - t0;
            if (writer_err) break;
        }
        to_writer.close();  // a failed writer must not leave the engines waiting to hand over
    });

    std::exception_ptr engine_err;
    {
        std::unique_ptr<Batch> b;
        while (!engine_err && to_engine.pop(b)) {
            if (b->err) {
                engine_err = b->err;
                break;
            }
            const double t0 = now();
            try {
                std::vector<std::exception_ptr> errs(ngpu);
                std::vector<std::thread> th;
                for (int g = 0; g < ngpu; g++)
                    th.emplace_back([&, g] {
                        try {
                            if (!b->sh[g].idx.empty()) run_shard(gpus[g].eng, b->sh[g], pool, signal_out, d.has_vad != 0, d.row_floats);
                        } catch (...) {
                            errs[g] = std::current_exception();
                        }
                    });
                for (auto &t : th) t.join();
                for (auto &e : errs)
                    if (e) std::rethrow_exception(e);
            } catch (...) {
                engine_err = std::current_exception();
                for (auto &sh : b->sh) {
                    pool.put(sh.arena);
                    pool.put(sh.rows);
                }
                break;
            }
            t_engine += now() - t0;
            if (!to_writer.push(std::move(b))) break;
        }
    }
    giving_up = giving_up || (bool)engine_err;
    to_engine.close();
    {   // a reader blocked on a full hand-over sees the close; batches it had queued go back to the pool
        std::unique_ptr<Batch> b;
        while (to_engine.pop(b))
            for (auto &sh : b->sh) pool.put(sh.arena);
    }
    reader.join();
    to_writer.close();
    writer.join();
    if (timing)
        std::fprintf(stderr, "host stages busy: reader %.3f s, engines %.3f s, writer %.3f s; loop %.3f s, %d + %d I/O threads, %d engine(s)\n", t_read, t_engine,
                     t_write, now() - t_loop0, io_threads, write_threads, ngpu);
    if (writer_err) std::rethrow_exception(writer_err);
    if (engine_err) std::rethrow_exception(engine_err);
    if (!deferred_error.empty()) throw Fatal(deferred_error);

    if (cmvn) {
        // speaker table in order of first appearance (cmvn_POST::add_spk, src/fea/post_impl.cc:120-142)
        std::vector<std::string> spk_names;
        std::vector<int32_t> spk_of(items.size());
        std::unordered_map<std::string, int32_t> spk_index;
        for (size_t i = 0; i < items.size(); i++) {
            auto ins = spk_index.emplace(items[i].spk, (int32_t)spk_names.size());
            if (ins.second) spk_names.push_back(items[i].spk);
            spk_of[i] = ins.first->second;
        }
        const int n_spk = (int)spk_names.size(), cols = ctu_cmvn_cols(gpus[0].eng);
        // shards over the whole corpus; every GPU keeps its rows in one block behind a plan with the same geometry
        struct Shard {
            std::vector<size_t> idx;
            std::vector<int32_t> spk;
            std::vector<float> rows;
            ctu_plan *plan = nullptr;
        };
        std::vector<Shard> sh(ngpu);
        {
            std::vector<size_t> load(ngpu, 0);
            for (size_t i = 0; i < items.size(); i++) {
                const int g = (int)(std::min_element(load.begin(), load.end()) - load.begin());
                sh[g].idx.push_back(i);
                load[g] += all_rows[i].size();
            }
        }
        for (int g = 0; g < ngpu; g++) {
            std::vector<int64_t> ns;
            for (size_t i : sh[g].idx) {
                ns.push_back(all_ns[i]);
                sh[g].spk.push_back(spk_of[i]);
                sh[g].rows.insert(sh[g].rows.end(), all_rows[i].begin(), all_rows[i].end());
                std::vector<float>().swap(all_rows[i]);
            }
            if (ctu_plan_create(gpus[g].eng, ns.data(), (int)ns.size(), &sh[g].plan) != CTU_OK) throw Fatal(ctu_last_error(gpus[g].eng));
        }
        auto reduce = [&](const double *mean) {  // one pass over every GPU's rows, partial sums added on the host
            std::vector<double> acc((size_t)n_spk * (cols + 1), 0.0);
            std::vector<std::vector<double>> part(ngpu, acc);
            std::vector<std::string> errs(ngpu);
            std::vector<std::thread> th;
            for (int g = 0; g < ngpu; g++)
                th.emplace_back([&, g] {
                    if (sh[g].idx.empty()) return;
                    if (ctu_cmvn_accumulate_host(gpus[g].eng, sh[g].plan, sh[g].rows.data(), sh[g].spk.data(), n_spk, mean,
                                                 part[g].data()) != CTU_OK)
                        errs[g] = ctu_last_error(gpus[g].eng);
                });
            for (auto &t : th) t.join();
            for (auto &e : errs)
                if (!e.empty()) throw Fatal(e);
            for (int g = 0; g < ngpu; g++)
                for (size_t k = 0; k < acc.size(); k++) acc[k] += part[g][k];
            return acc;
        };
        std::vector<double> mean((size_t)n_spk * cols), var((size_t)n_spk * cols);
        {
            const std::vector<double> a = reduce(nullptr);  // sum_fea + stat_cm (post_impl.cc:51-76)
            for (int s_ = 0; s_ < n_spk; s_++)
                for (int k = 0; k < cols; k++) mean[(size_t)s_ * cols + k] = a[(size_t)s_ * (cols + 1) + k] / a[(size_t)s_ * (cols + 1) + cols];
            const std::vector<double> b = reduce(mean.data());  // sum_cv + stat_cv (post_impl.cc:78-102)
            for (int s_ = 0; s_ < n_spk; s_++)
                for (int k = 0; k < cols; k++) var[(size_t)s_ * cols + k] = b[(size_t)s_ * (cols + 1) + k] / (b[(size_t)s_ * (cols + 1) + cols] - 1);
        }
        {  // cmvnOUT::save_frame (src/io/out.cc:591-613)
            FILE *f = std::fopen(stat_path.c_str(), "wt");
            if (!f) throw Fatal("OUT: Cannot create output file with stat. of cmvn!");
            for (int s_ = 0; s_ < n_spk; s_++) {
                std::fprintf(f, "%s\nmean\t", spk_names[s_].c_str());
                for (int k = 0; k < cols; k++) std::fprintf(f, k + 1 < cols ? "%f " : "%f", mean[(size_t)s_ * cols + k]);
                std::fprintf(f, "\nvar\t");
                for (int k = 0; k < cols; k++) std::fprintf(f, k + 1 < cols ? "%f " : "%f\n", var[(size_t)s_ * cols + k]);
            }
            std::fclose(f);
        }
        if (!o.apply_cmvn && cmvn_vad && o.vad_out_mode != "none")  // statistics only: VAD::new_file still opens every file's VAD output
            for (const Item &it : items)
                if (!it.fvad.empty()) write_file(it.fvad, std::vector<uint8_t>(), "FileWriter: cannot open file!");
        if (o.apply_cmvn) {
            std::vector<std::string> errs(ngpu);
            std::vector<std::thread> th;
            for (int g = 0; g < ngpu; g++)
                th.emplace_back([&, g] {
                    if (sh[g].idx.empty()) return;
                    if (ctu_cmvn_apply_host(gpus[g].eng, sh[g].plan, sh[g].rows.data(), sh[g].spk.data(), n_spk, mean.data(),
                                            var.data()) != CTU_OK)
                        errs[g] = ctu_last_error(gpus[g].eng);
                });
            for (auto &t : th) t.join();
            for (auto &e : errs)
                if (!e.empty()) throw Fatal(e);
            // back to list order, then the writers
            std::vector<std::pair<int, size_t>> where(items.size());  // (gpu, position in that GPU's shard)
            for (int g = 0; g < ngpu; g++)
                for (size_t k = 0; k < sh[g].idx.size(); k++) where[sh[g].idx[k]] = {g, k};
            int32_t ring_hidx = 0, ring_hsize = 0;  // the VAD's majority filter along the list (this is the only pass that pushes into it)
            std::vector<float> vrows;
            std::vector<int32_t> src;
            for (size_t i = 0; i < items.size(); i++) {
                const int g = where[i].first;
                const size_t k = where[i].second;
                const int64_t *ro = ctu_plan_row_offsets(sh[g].plan);
                int64_t nr = ro[k + 1] - ro[k];
                const float *r = sh[g].rows.data() + (size_t)ro[k] * d.row_floats;
                if (cmvn_vad) {
                    // what BATCH::save_frame does with the normalised vector (src/io/batch.cc:230-241): through the majority filter's ring -
                    // whose index the previous files of the list have left somewhere (include/ctu_engine.h) -, the decision, the drop.  The
                    // energy column does not go through the ring (the engine has already moved it to the row the writer reads it with).
                    const int64_t T = nr, D = d.row_floats;
                    const int e_col = o.fea_E ? (int)D - 1 : -1;
                    src.assign((size_t)std::max<int64_t>(T, 1), -1);
                    if (ring_hsize != 0)  // as in the reader above: a half-drained filter is not reproduced
                        throw Fatal("VAD: " + items[i - 1].fin + " has no more frames than the majority filter delays: the reference's filter stays "
                                    "half drained for the files behind it (src/vad/vad.h:156-175), which is not reproduced");
                    const int64_t n_out = ctu_vad_ring_rows(o.vad_filter_order, T, ring_hidx, src.data());
                    ctu_vad_ring_step(o.vad_filter_order, T, &ring_hidx, &ring_hsize);
                    const std::vector<uint8_t> &v = all_vads[i];
                    std::vector<uint8_t> dec;
                    vrows.clear();
                    for (int64_t q = 0; q < n_out; q++) {
                        const uint8_t b = v[(size_t)q];
                        if (b) dec.push_back(b);
                        if (o.vad_apply_mode == "drop" && b != '1') continue;
                        const size_t at = vrows.size();
                        vrows.resize(at + (size_t)D, 0.f);
                        for (int c = 0; c < (int)D; c++) {
                            if (c == e_col) vrows[at + c] = r[(size_t)q * D + c];
                            else if (src[(size_t)q] >= 0) vrows[at + c] = r[(size_t)src[(size_t)q] * D + c];
                        }
                    }
                    if (o.vad_out_mode != "none") {
                        if (items[i].fvad.empty()) throw Fatal("VAD::new_file(): invalid filename!");
                        write_file(items[i].fvad, dec, "FileWriter: cannot open file!");
                    }
                    nr = (int64_t)(vrows.size() / (size_t)D);
                    r = vrows.data();
                }
                if (ark) ark->add(items[i].fout, r, nr, d.row_floats);
                else if (pf) pf->add(r, nr, d.row_floats);
                else write_htk(items[i].fout, r, nr, d);
            }
        }
        for (auto &x : sh)
            if (x.plan) ctu_plan_destroy(x.plan);
    }
    if (pf) pf->close();
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    try {
        return real_main(argc, argv);
    } catch (const std::exception &e) {  // the reference prints the thrown text and returns -1 (src/main.cpp:54-60)
        std::fprintf(stderr, "%s\n", e.what());
        return -1;
    }
}
