// ctucopy -- command-line front end of the MI355X feature engine.
//
// Keeps the reference's command line (src/io/opts.cc:644-846), `-S` list format
// (`fin fout [spk] [vadfile]`, src/io/batch.cc:349-356), input decoders (raw / a-law / mu-law / WAVE,
// src/io/in.cc:434-619, src/io/amulaw.h:20-53) and feature writers (HTK src/io/out.cc:115-213, KALDI ark+scp
// :648-781, ICSI pfile src/io/pfile.cc:435-592), so that it is a drop-in for batch feature extraction.
// The per-frame chain itself runs on the GPU(s) behind include/ctu_engine.h; this file only moves bytes.
//
// Host loop (the counterpart of BATCH::process, src/io/batch.cc:326-421): files are decoded into a packed PCM
// arena in bounded batches, utterances of a batch are sharded over the GPUs by frame count (no collective:
// every utterance is independent), results are written in list order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

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
std::vector<uint8_t> read_all(const std::string &path, const char *err) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw Fatal(err);
    std::vector<uint8_t> buf;
    uint8_t tmp[1 << 16];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    return buf;
}

uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

std::vector<int16_t> decode(const ctu::Opts &o, const std::string &path) {
    std::vector<int16_t> pcm;
    if (o.format_in == "raw") {
        const std::vector<uint8_t> b = read_all(path, "IN: Cannot open data file!");
        pcm.resize(b.size() / 2);
        for (size_t i = 0; i < pcm.size(); i++)
            pcm[i] = (int16_t)(o.swap_in ? (b[2 * i] << 8 | b[2 * i + 1]) : (b[2 * i + 1] << 8 | b[2 * i]));
    } else if (o.format_in == "alaw" || o.format_in == "mulaw") {
        const std::vector<uint8_t> b = read_all(path, "IN: Cannot open data file!");
        pcm.resize(b.size());
        const bool alaw = o.format_in == "alaw";
        for (size_t i = 0; i < b.size(); i++) pcm[i] = g711_to_linear(b[i], alaw);
    } else if (o.format_in == "wave") {  // canonical 44-byte header only, like src/io/in.cc:550-597
        const std::vector<uint8_t> b = read_all(path, "IN: Cannot open file!");
        if (b.size() < 44 || std::memcmp(b.data(), "RIFF", 4)) throw Fatal("IN: No RIFF header in file!");
        if (std::memcmp(b.data() + 8, "WAVE", 4)) throw Fatal("IN: Not a WAVE file!");
        if (rd16(&b[20]) != 1) throw Fatal("IN: Not a PCM WAVE file!");
        if ((long)rd32(&b[24]) != o.fs) throw Fatal("IN: WAVE file reports different sampling rate than specified!");
        if (rd16(&b[22]) != 1) throw Fatal("IN: Input WAVE file is not mono!");
        if (rd16(&b[34]) != 16) throw Fatal("IN: Not 16 bits per sample!");
        size_t n = rd32(&b[40]) / 2;
        n = std::min(n, (b.size() - 44) / 2);
        pcm.resize(n);
        for (size_t i = 0; i < n; i++) pcm[i] = (int16_t)rd16(&b[44 + 2 * i]);
    } else {
        throw Fatal("IN: Unknown input file format!");
    }
    return pcm;
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
    std::vector<uint8_t> b;
    b.reserve(12 + (size_t)n * d.row_floats * 4);
    const bool big = d.swap_out;
    put32(b, (uint32_t)n, big);
    put32(b, d.htk_period, big);
    put16(b, (uint16_t)(4 * d.row_floats), big);
    put16(b, (uint16_t)d.htk_kind, big);
    for (int64_t i = 0; i < n * d.row_floats; i++) putf(b, rows[i], big);
    write_file(path, b, "OUT: Cannot create output file!");
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

void run_shard(ctu_engine *eng, const std::vector<const std::vector<int16_t> *> &utts, std::vector<std::vector<float>> &out,
               std::vector<std::string> &vad_out, bool has_vad, int row_floats, std::string &err) {
    std::vector<int64_t> ns;
    for (auto *u : utts) ns.push_back((int64_t)u->size());
    ctu_plan *plan = nullptr;
    if (ctu_plan_create(eng, ns.data(), (int)ns.size(), &plan) != CTU_OK) {
        err = ctu_last_error(eng);
        return;
    }
    const int64_t *so = ctu_plan_sample_offsets(plan), *ro = ctu_plan_row_offsets(plan);
    // page-locked staging for the arena and the rows: DMA at the link rate (include/ctu_engine.h, ctu_host_alloc)
    const size_t ns_total = (size_t)ctu_plan_total_samples(plan), nr_total = (size_t)ctu_plan_total_frames(plan) * row_floats;
    int16_t *arena = static_cast<int16_t *>(ctu_host_alloc(ns_total * sizeof(int16_t)));
    float *rows = static_cast<float *>(ctu_host_alloc((nr_total ? nr_total : 1) * sizeof(float)));
    if (!arena || !rows) {
        err = "ENGINE: cannot allocate page-locked host memory";
        ctu_host_free(arena);
        ctu_host_free(rows);
        ctu_plan_destroy(plan);
        return;
    }
    std::memset(arena, 0, ns_total * sizeof(int16_t));
    for (size_t i = 0; i < utts.size(); i++) std::copy(utts[i]->begin(), utts[i]->end(), arena + so[i]);
    std::vector<uint8_t> vad(has_vad ? (size_t)ctu_plan_total_frames(plan) : 0);
    std::vector<int64_t> kept(utts.size());
    if (ctu_engine_run_host(eng, plan, arena, rows, has_vad ? vad.data() : nullptr, kept.data()) != CTU_OK)
        err = ctu_last_error(eng);
    else
        for (size_t i = 0; i < utts.size(); i++) {
            // rows_per_utt < frames only with -vad_apply_mode drop (rows compacted in place by the library)
            out[i].assign(rows + ro[i] * row_floats, rows + (ro[i] + kept[i]) * row_floats);
            if (has_vad) vad_out[i].assign(vad.begin() + ro[i], vad.begin() + ro[i + 1]);
        }
    ctu_host_free(arena);
    ctu_host_free(rows);
    ctu_plan_destroy(plan);
}

// -format_out raw|wave: one shard through ctu_engine_run_signal_host, samples per utterance back in `out`
void run_shard_signal(ctu_engine *eng, const std::vector<const std::vector<int16_t> *> &utts, std::vector<std::vector<int16_t>> &out,
                      std::string &err) {
    std::vector<int64_t> ns;
    for (auto *u : utts) ns.push_back((int64_t)u->size());
    ctu_plan *plan = nullptr;
    if (ctu_plan_create(eng, ns.data(), (int)ns.size(), &plan) != CTU_OK) {
        err = ctu_last_error(eng);
        return;
    }
    const int64_t *so = ctu_plan_sample_offsets(plan), *no = ctu_plan_out_samples(plan);
    std::vector<int16_t> arena((size_t)ctu_plan_total_samples(plan), 0), res(arena.size(), 0);
    for (size_t i = 0; i < utts.size(); i++) std::copy(utts[i]->begin(), utts[i]->end(), arena.begin() + so[i]);
    if (ctu_engine_run_signal_host(eng, plan, arena.data(), res.data()) != CTU_OK) err = ctu_last_error(eng);
    else
        for (size_t i = 0; i < utts.size(); i++) out[i].assign(res.begin() + so[i], res.begin() + so[i] + no[i]);
    ctu_plan_destroy(plan);
}

// rawOUT::write (src/io/out.cc:493-499): int16 samples, byte-swapped for -endian_out big
void write_raw(const std::string &path, const std::vector<int16_t> &x, bool big) {
    std::vector<uint8_t> b;
    b.reserve(x.size() * 2);
    for (int16_t v : x) put16(b, (uint16_t)v, big);
    write_file(path, b, "OUT: Cannot open output stream!");
}

// waveOUT (src/io/out.cc:517-564): canonical 44-byte RIFF header, sizes patched at close, samples in host order
void write_wave(const std::string &path, const std::vector<int16_t> &x, int fs) {
    std::vector<uint8_t> b;
    b.reserve(44 + x.size() * 2);
    const uint32_t data = (uint32_t)(2 * x.size());
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
    for (int16_t v : x) put16(b, (uint16_t)v, false);
    write_file(path, b, "OUT: Cannot open data file!");
}

int real_main(int argc, char **argv) {
    std::vector<std::string> args;
    int ngpu = 1;
    std::vector<int> gpu_map;  // --gpu-map a,b,...: device ordinal of every engine (default 0 .. N-1); an ordinal may repeat, which puts
                               // several engines on one device - the multi-engine host path rehearsed on a box with fewer GPUs
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) ngpu = std::max(1, std::atoi(argv[++i]));
        else if (!std::strcmp(argv[i], "--gpu-map") && i + 1 < argc) {
            std::istringstream ms(argv[++i]);
            std::string tok;
            while (std::getline(ms, tok, ',')) gpu_map.push_back(std::atoi(tok.c_str()));
        }
        else args.emplace_back(argv[i]);
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
    std::vector<Gpu> gpus(ngpu);
    for (int g = 0; g < ngpu; g++)
        if (ctu_engine_create((int)cargs.size(), cargs.data(), gpu_map.empty() ? g : gpu_map[g], &gpus[g].eng) != CTU_OK) throw Fatal(ctu_create_error());
    ctu_dims d;
    ctu_engine_dims(gpus[0].eng, &d);

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

    const size_t batch_samples = 512u << 20;  // ~1 GiB of PCM per batch
    size_t pos = 0;
    while (pos < items.size()) {
        std::vector<std::vector<int16_t>> pcm;
        size_t total = 0, end = pos;
        while (end < items.size() && (total < batch_samples || end == pos)) {
            pcm.push_back(decode(o, items[end].fin));
            if (ctu_num_frames(gpus[0].eng, (int64_t)pcm.back().size()) < 0) throw Fatal("IO: Signal shorter than one frame!");
            total += pcm.back().size();
            end++;
        }
        const size_t n = end - pos;
        // longest-processing-time sharding over the GPUs by frame count
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return pcm[a].size() > pcm[b].size(); });
        std::vector<std::vector<size_t>> shard(ngpu);
        std::vector<size_t> load(ngpu, 0);
        for (size_t i : order) {
            const int g = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            shard[g].push_back(i);
            load[g] += pcm[i].size();
        }
        for (auto &sh : shard) std::sort(sh.begin(), sh.end());  // list order inside a shard
        if (signal_out) {  // speech enhancement: samples instead of rows (src/io/batch.cc:62-65,223-227)
            std::vector<std::vector<int16_t>> wav(n);
            std::vector<std::string> errs(ngpu);
            std::vector<std::thread> th;
            for (int g = 0; g < ngpu; g++)
                th.emplace_back([&, g] {
                    std::vector<const std::vector<int16_t> *> u;
                    std::vector<std::vector<int16_t>> out(shard[g].size());
                    for (size_t i : shard[g]) u.push_back(&pcm[i]);
                    if (!u.empty()) run_shard_signal(gpus[g].eng, u, out, errs[g]);
                    for (size_t k = 0; k < shard[g].size(); k++) wav[shard[g][k]] = std::move(out[k]);
                });
            for (auto &t : th) t.join();
            for (auto &e : errs)
                if (!e.empty()) throw Fatal(e);
            for (size_t i = 0; i < n; i++) {
                const Item &it = items[pos + i];
                if (o.verbose) std::fprintf(stderr, "processing: %s - %lld frames.\n", it.fin.c_str(),
                                            (long long)ctu_num_frames(gpus[0].eng, (int64_t)pcm[i].size()));
                if (o.format_out == "raw") write_raw(it.fout, wav[i], d.swap_out != 0);
                else write_wave(it.fout, wav[i], o.fs);
            }
            pos = end;
            continue;
        }
        std::vector<std::vector<float>> rows(n);
        std::vector<std::string> vads(n);
        std::vector<std::string> errs(ngpu);
        std::vector<std::thread> th;
        for (int g = 0; g < ngpu; g++)
            th.emplace_back([&, g] {
                std::vector<const std::vector<int16_t> *> u;
                std::vector<std::vector<float>> out(shard[g].size());
                std::vector<std::string> vout(shard[g].size());
                for (size_t i : shard[g]) u.push_back(&pcm[i]);
                if (!u.empty()) run_shard(gpus[g].eng, u, out, vout, d.has_vad != 0, d.row_floats, errs[g]);
                for (size_t k = 0; k < shard[g].size(); k++) {
                    rows[shard[g][k]] = std::move(out[k]);
                    vads[shard[g][k]] = std::move(vout[k]);
                }
            });
        for (auto &t : th) t.join();
        for (auto &e : errs)
            if (!e.empty()) throw Fatal(e);
        for (size_t i = 0; i < n; i++) {
            const Item &it = items[pos + i];
            const int64_t nr = (int64_t)rows[i].size() / d.row_floats;
            if (o.verbose) std::fprintf(stderr, "processing: %s - %lld frames.\n", it.fin.c_str(), (long long)nr);
            if (cmvn) {  // rows wait for the corpus statistics
                all_rows[pos + i] = std::move(rows[i]);
                all_ns[pos + i] = (int64_t)pcm[i].size();
                continue;
            }
            if (d.has_vad && o.vad_out_mode != "none") {  // one ASCII '0'/'1' per frame (src/vad/vad.h:67-70)
                if (it.fvad.empty()) throw Fatal("VAD::new_file(): invalid filename!");
                write_file(it.fvad, std::vector<uint8_t>(vads[i].begin(), vads[i].end()), "FileWriter: cannot open file!");
            }
            if (ark) ark->add(it.fout, rows[i].data(), nr, d.row_floats);
            else if (pf) pf->add(rows[i].data(), nr, d.row_floats);
            else {
                // -fea_trap: the reference's writers overwrite fea_kind with "spec" when they save their first frame
                // (src/io/out.cc:182), so every header after the first file carries base kind 8 (out.cc:146-152).
                ctu_dims dh = d;
                if (o.fea_trap && pos + i > 0) dh.htk_kind = (d.htk_kind & ~077) | 8;
                write_htk(it.fout, rows[i].data(), nr, dh);
            }
        }
        pos = end;
    }
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
            for (size_t i = 0; i < items.size(); i++) {
                const int g = where[i].first;
                const size_t k = where[i].second;
                const int64_t *ro = ctu_plan_row_offsets(sh[g].plan);
                const int64_t nr = ro[k + 1] - ro[k];
                const float *r = sh[g].rows.data() + (size_t)ro[k] * d.row_floats;
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
