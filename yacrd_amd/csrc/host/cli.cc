// cli.cc — `yacrd` drop-in command line over the MI355X engine.
//
// Same flags, defaults, file-type rules and outputs as the reference binary
// (src/cli.rs:33-137, src/main.rs:36-137):
//   yacrd -i <overlaps.paf|.m4|.mhap|report.yacrd> -o <report.yacrd> [-t N] [-c COV] [-n RATIO]
//         [--read-buffer-size N] [-d PREFIX] [--ondisk-buffer-size N]
//         [scrubb|filter|extract|split -i <in> -o <out>]
// Additive flags only: --gpus N (default 1): an input beyond one GPU's memory is partitioned over N GPUs by id handle.
// The bad-region computation and the read classification run on the GPU through
// include/yacrd_engine.h; there is no CPU fallback — without a gfx950 device this exits non-zero.
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/yacrd_engine.h"
#include "../../../include/yacrd_host.h"

namespace {

const char *kVersion = "1.0.0 Magby";

[[noreturn]] void die(const std::string &msg, int code = 1)
{
    std::fprintf(stderr, "Error: %s\n", msg.c_str());
    std::exit(code);
}

void usage(FILE *f)
{
    std::fprintf(f,
                 "yacrd %s (MI355X engine)\n"
                 "USAGE:\n    yacrd [OPTIONS] --input <INPUT> --output <OUTPUT> [SUBCOMMAND]\n\n"
                 "OPTIONS:\n"
                 "    -i, --input <INPUT>                  overlap file (.paf|.m4|.mhap) or yacrd report (.yacrd)\n"
                 "    -o, --output <OUTPUT>                path output file\n"
                 "    -t, --thread <THREADS>               host threads for the overlap parse [default: 1, like the\n"
                 "                                         reference's rayon pool]; 0 = every usable CPU, which is what\n"
                 "                                         you want: the parse is the whole wall clock\n"
                 "    -c, --coverage <COVERAGE>            if coverage reach this value region is marked as bad [default: 0]\n"
                 "    -n, --not-coverage <NOT_COVERAGE>    bad-region ratio above which a read is NotCovered [default: 0.8]\n"
                 "        --read-buffer-size <N>           accepted for compatibility [default: 8192]\n"
                 "    -d, --ondisk <PREFIX>                accepted for compatibility (HBM + host RAM hold the overlaps)\n"
                 "        --ondisk-buffer-size <N>         accepted for compatibility [default: 64000000]\n"
                 "        --gpus <N>                       GPUs to partition the reads over [default: 1]\n"
                 "    -h, --help    -V, --version\n\n"
                 "SUBCOMMANDS (each takes -i <input> -o <output>):\n"
                 "    scrubb     all bad region of read is removed\n"
                 "    filter     record mark as chimeric or NotCovered is filter\n"
                 "    extract    record mark as chimeric or NotCovered is extract\n"
                 "    split      record mark as chimeric or NotCovered is split\n",
                 kVersion);
}

bool parse_u64(const char *s, unsigned long long &v)
{
    if (!*s) return false;
    errno = 0;
    char *end = nullptr;
    if (*s == '-') return false;
    v = std::strtoull(s, &end, 10);
    return errno == 0 && end && *end == '\0';
}

bool has(const std::string &s, const char *needle) { return s.find(needle) != std::string::npos; }

} // namespace

int main(int argc, char **argv)
{
    std::string input, output, sub, sub_in, sub_out, ondisk;
    unsigned long long threads = 1, coverage = 0, gpus = 1, tmp = 0;
    double not_coverage = 0.8;

    int i = 1;
    auto value = [&](const char *flag) -> const char * {
        if (i + 1 >= argc) die(std::string("The argument '") + flag + "' requires a value", 2);
        return argv[++i];
    };
    for (; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            usage(stdout);
            return 0;
        } else if (a == "-V" || a == "--version") {
            std::printf("yacrd %s\n", kVersion);
            return 0;
        } else if (a == "-i" || a == "--input") input = value("--input");
        else if (a == "-o" || a == "--output") output = value("--output");
        else if (a == "-t" || a == "--thread") {
            if (!parse_u64(value("--thread"), threads)) die("Invalid value for '--thread'", 2);
        } else if (a == "-c" || a == "--coverage") {
            if (!parse_u64(value("--coverage"), coverage)) die("Invalid value for '--coverage'", 2);
        } else if (a == "-n" || a == "--not-coverage") {
            char *end = nullptr;
            const char *v = value("--not-coverage");
            not_coverage = std::strtod(v, &end);
            if (!*v || (end && *end)) die("Invalid value for '--not-coverage'", 2);
        } else if (a == "--read-buffer-size") {
            if (!parse_u64(value("--read-buffer-size"), tmp)) die("Invalid value for '--read-buffer-size'", 2);
        } else if (a == "-d" || a == "--ondisk") ondisk = value("--ondisk");
        else if (a == "--ondisk-buffer-size") (void)value("--ondisk-buffer-size");
        else if (a == "--gpus") {
            if (!parse_u64(value("--gpus"), gpus) || gpus == 0) die("Invalid value for '--gpus'", 2);
        } else if (a == "scrubb" || a == "filter" || a == "extract" || a == "split") {
            sub = a;
            for (i++; i < argc; i++) {
                const std::string b = argv[i];
                if (b == "-i" || b == "--input") sub_in = value("--input");
                else if (b == "-o" || b == "--output") sub_out = value("--output");
                else die("Found argument '" + b + "' which wasn't expected in subcommand " + sub, 2);
            }
            if (sub_in.empty() || sub_out.empty())
                die("The following required arguments were not provided: --input --output (" + sub + ")", 2);
        } else {
            die("Found argument '" + a + "' which wasn't expected", 2);
        }
    }
    if (input.empty() || output.empty()) {
        usage(stderr);
        die("The following required arguments were not provided: --input <INPUT> --output <OUTPUT>", 2);
    }
    if (!ondisk.empty())
        std::fprintf(stderr, "[INFO] --ondisk is accepted for compatibility; overlaps are kept in "
                             "host memory and HBM (same results, reference tests/run.rs:120-160)\n");

    // YACRD_CLI_TIMING=1: wall clock of every stage on stderr (tools/e2e_scrubb_full.py)
    const bool timing = std::getenv("YACRD_CLI_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto stage = [&](const char *what) {
        const auto now = std::chrono::steady_clock::now();
        if (timing) std::fprintf(stderr, "[timing] %s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };

    // ---- engines (one per GPU)
    std::vector<yacrd_engine *> engines;
    // YACRD_GPUS_ON_DEVICE=<d>: every engine on device d (how the N > 1 path is tested on a one-GPU box)
    const char *same_dev = std::getenv("YACRD_GPUS_ON_DEVICE");
    for (unsigned long long g = 0; g < gpus; g++) {
        // a process runs ONE batch per engine: a short one goes out as one kernel launch (csrc/one_batch.h; anything that
        // launch does not take falls back to the default path inside the engine); YACRD_CLI_NO_ONE_LAUNCH=1 for A/B
        const char *no_one = std::getenv("YACRD_CLI_NO_ONE_LAUNCH");
        const uint32_t eflags = (no_one && *no_one && *no_one != '0') ? YACRD_F_DEFAULT : YACRD_F_ONE_LAUNCH;
        yacrd_engine_cfg cfg = {same_dev && *same_dev ? (int32_t)std::atoi(same_dev) : (int32_t)g, eflags};
        yacrd_engine *e = nullptr;
        if (yacrd_engine_create(&cfg, &e) != YACRD_OK) die(yacrd_last_error());
        engines.push_back(e);
    }
    const uint32_t cov32 = coverage > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)coverage;
    stage("engine_create");

    yacrd_csr *csr = nullptr;
    yacrd_report *rep = nullptr;
    yacrd_result res{};
    std::vector<uint8_t> rep_types;
    yacrd_badparts_view bp{};
    yacrd_csr_view view{};
    yacrd_reads dev_reads{};
    yacrd_text text{}; // a compressed input, inflated (empty otherwise); lives as long as what was parsed from it

    // src/main.rs:43-60: a .yacrd input bypasses detection (FromReport), anything else is overlaps
    const bool m4 = has(input, ".m4") || has(input, ".mhap"), paf = has(input, ".paf");
    if (!m4 && !paf && has(input, ".yacrd")) {
        if (yacrd_report_read(input.c_str(), &rep)) die(yacrd_host_last_error());
        yacrd_report_get(rep, &bp);
        rep_types.resize((size_t)bp.n_reads + 1);
        // type_of_read with this invocation's -n, on the GPU (kernel #2)
        if (yacrd_engine_classify(engines[0], bp.bad_offsets, bp.bad_regions, bp.lengths, bp.n_reads,
                                  not_coverage, rep_types.data()) != YACRD_OK)
            die(yacrd_last_error());
        bp.read_type = rep_types.data();
        view.n_reads = bp.n_reads;
        view.name_off = bp.name_off;
        view.names = bp.names;
        view.lengths = bp.lengths;
    } else {
        int dev_parse = YACRD_EFALLBACK;
        // YACRD_NO_DEVICE_PARSER=1: the host parser for everything (A/B, tools/e2e_cli_paf.py)
        const char *no_dev = std::getenv("YACRD_NO_DEVICE_PARSER");
        // (--gpus N > 1: every GPU moves and parses a byte range of the text over its own link and sweeps a range of the reads,
        // yacrd_engines_ingest_overlaps; an input beyond the GPUs' memory — the call estimates every device's need, its ranges
        // and the records it gathers from the others, before it allocates anything, and an allocation that fails later after all
        // comes back as an error too — is routed to all N by the host parser's stream group)
        if ((paf || m4) && !(no_dev && *no_dev == '1')) {
            // one GPU, PAF or M4 text: the host only moves the file to HBM, the device parses it, numbers the reads,
            // builds the CSR and runs the engine (yacrd_engine_ingest_overlaps).  Whatever is not a plain file of
            // plain records (compressed, quoted fields, lone CRs, 0x integers, malformed lines ...) comes back as
            // YACRD_EFALLBACK and takes the host parser below, which knows the whole syntax and the messages.
            // The threads that move the file into pinned memory are not the reference's rayon pool: their number does
            // not follow -t (default 1 = 8 GB/s of a 56 GB/s link); the engine's own default (6) unless
            // YACRD_COPY_THREADS says otherwise.
            const char *ct = std::getenv("YACRD_COPY_THREADS");
            const int copy_threads = ct && *ct ? std::max(0, std::atoi(ct)) : 0;
            // a gzip / bzip2 / xz file (src/util.rs:57-70 sniffs every input): inflated into memory first — BGZF members on
            // every usable CPU, any other stream on one thread, which then is the wall clock whichever parser follows
            const int rct = yacrd_text_from_file(input.c_str(), threads == 1 ? 0 : (int)threads, &text);
            if (rct == 1) die(yacrd_host_last_error());
            stage("inflate");
            if (rct == 0)
                dev_parse = yacrd_engines_ingest_overlaps_mem(engines.data(), (uint32_t)engines.size(), text.data, text.n, m4 ? 2 : 1,
                                                              copy_threads, cov32, not_coverage, &res, &dev_reads, nullptr);
            else
                dev_parse = yacrd_engines_ingest_overlaps(engines.data(), (uint32_t)engines.size(), input.c_str(), m4 ? 2 : 1, copy_threads,
                                                          cov32, not_coverage, &res, &dev_reads, nullptr);
            if (dev_parse == YACRD_ENOMEM) { // HBM ran out on the way (the parse wants ~2.6 x the file): the streamed host parse needs a fifth
                std::fprintf(stderr, "[INFO] device parser: %s; falling back to the host parser\n", yacrd_last_error());
                for (yacrd_engine *en : engines) (void)yacrd_engine_trim(en);
                dev_parse = YACRD_EFALLBACK;
            }
            if (dev_parse != YACRD_OK && dev_parse != YACRD_EFALLBACK) die(yacrd_last_error());
        }
        if (dev_parse == YACRD_OK) {
            if (std::getenv("YACRD_CLI_TIMING")) std::fprintf(stderr, "[info] device parser: %zu engine(s)\n", engines.size());
            view.n_reads = dev_reads.n_reads;
            view.name_off = dev_reads.name_off;
            view.names = dev_reads.names;
            view.lengths = dev_reads.lengths;
        } else {
            // The parser's records cross PCIe from pinned buffers while it is still parsing — routed by
            // handle mod N to the GPUs of their two reads (yacrd_stream_group_*; with one GPU the sink is the
            // stream's own) — every GPU builds the CSR of its reads in HBM and runs on it, no collective
            // (reads are independent, src/stack.rs:61); results come back in first-appearance order.
            yacrd_stream_group *grp = nullptr;
            if (yacrd_stream_group_open(engines.data(), (uint32_t)engines.size(), 0, 0, &grp) != YACRD_OK)
                die(yacrd_last_error());
            yacrd_rec_sink sink;
            yacrd_stream_group_sink(grp, &sink);
            // (a compressed input the device parser handed over is parsed from the text already inflated)
            if (text.data ? yacrd_ingest_stream_memory(text.data, (size_t)text.n, m4 ? 2 : 1, (int)threads, &sink, &csr)
                          : yacrd_ingest_stream(input.c_str(), 0, (int)threads, &sink, &csr))
                die(yacrd_host_last_error());
            yacrd_csr_get(csr, &view);
            const uint32_t *map = nullptr;
            uint64_t n_handles = 0;
            if (yacrd_csr_handle_map(csr, &map, &n_handles)) die(yacrd_host_last_error());
            if (yacrd_stream_group_finish(grp, map, n_handles, view.lengths, view.n_reads, cov32, not_coverage,
                                          &res) != YACRD_OK)
                die(yacrd_last_error());
            yacrd_stream_group_close(grp);
        }
        bp.n_reads = view.n_reads;
        bp.name_off = view.name_off;
        bp.names = view.names;
        bp.lengths = view.lengths;
        bp.bad_offsets = res.bad_offsets;
        bp.bad_regions = res.bad_regions;
        bp.read_type = res.read_type;
    }

    stage("detect");
    // src/main.rs:62-84: the report, one line per read
    if (yacrd_report_write(output.c_str(), &view, bp.bad_offsets, bp.bad_regions, bp.read_type))
        die(yacrd_host_last_error());
    stage("report");

    // src/main.rs:86-118: optional post operation
    if (!sub.empty()) {
        const int op = sub == "scrubb" ? YACRD_OP_SCRUBB
                                       : sub == "filter" ? YACRD_OP_FILTER
                                                         : sub == "extract" ? YACRD_OP_EXTRACT : YACRD_OP_SPLIT;
        if (yacrd_edit_file(op, sub_in.c_str(), sub_out.c_str(), &bp)) die(yacrd_host_last_error());
        stage("edit");
    }

    yacrd_result_free(&res);
    yacrd_reads_free(&dev_reads);
    yacrd_text_free(&text);
    if (csr) yacrd_csr_free(csr);
    if (rep) yacrd_report_free(rep);
    for (yacrd_engine *e : engines) yacrd_engine_destroy(e);
    return 0;
}
