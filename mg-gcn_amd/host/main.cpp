// main.cpp -- the `mg_gcn` command line of the reference (src/main.cpp) on the HIP engine.
//
//   mg_gcn [-h] [-P gpus] [-R 0/1] [-E epochs] [-S x] [-N] train <dir> <k> <h1> ... <hk>
//
// Same flag grammar (getopt "h?P:R:E:S:N", src/main.cpp:58), same dataset files
// (graph.bin / features.bin / labels.bin / sets.bin, :82-85), same stderr lines
// ("n nnz", "num_labels = ", "feature size = ", then per epoch "e loss acc seconds", :87-91,
// :130, :167), same per-epoch timer dump "csvs/<name>_<sizes>_<P>.csv" (:100-111, :131, :168),
// same hyper-parameters (Adam 1e-2 / 0.9 / 0.999 / 5e-4 / 1e-8, :126).  Like the reference,
// P > 1 trains only with -R 1 (row partition); classes are padded to a multiple of P (:135).
// `-R 1` routes through the distributed classes at any P (at -P 1 too: one rank, same schedule).
// Environment: MGGCN_DIST_MODE=allgather|halo|rounds picks the exchange schedule (ops.hpp; rounds = the
// reference's broadcast pipeline); MGGCN_FUSED=0 replays the reference's launch sequence;
// MGGCN_OVERSUBSCRIBE=1 lets -P exceed the visible GPUs (ranks wrap over them, peer-copy transport);
// MGGCN_HOIST_FIRST_AGGREGATION=1 (single GPU) pre-computes the first layer's A.X once (6 SpMMs per epoch: not the
// reference's epoch, same results at 1e-4); MGGCN_TIMING=1 prints the start-up stages.
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "gcn.hpp"

using x_t = unsigned;
using v_t = unsigned;
using r_t = float;

class arg_error : public std::runtime_error {
public:
    template <typename T>
    arg_error(T t) : std::runtime_error(t) {}
};

static int usage_(char *prog) {
    std::cout << "Usage: " << prog << " [-h] [-P gpus] [-R 0/1] [-E epochs] [-S x] [-N] train <dir> <k> <h1..hk>" << std::endl;
    return 0;
}

static int help_() {
    std::cout << "\nMG-GCN full-graph multi-GPU GCN training, MI355X (gfx950) engine.\n\n"
                 "Options:\n"
                 "    -P : number of GPUs\n"
                 "    -R : enable the 1D row partition (required for -P > 1)\n"
                 "    -E : number of epochs (default 20)\n"
                 "    -S : disable communication/computation overlap\n"
                 "Arguments:\n"
                 "    train <dir> <k> <h1> ... <hk> : dataset directory, number of hidden layers and their widths\n";
    return EXIT_SUCCESS;
}

// MGGCN_DUMP_WEIGHTS=<dir>: before every epoch write each layer's W and b as dense .bin files
// (<dir>/e<epoch>_W<layer>.bin, _b<layer>.bin; the reference's own dense format, u32 N, u32 M, f32 payload) so
// that a checker can replay any epoch from the exact state it started in (tests/test_gpu_host_cpp.py).
static void dump_dense(const std::filesystem::path &path, const dn_matrix<float> &A) {
    const auto h = A.to_host();
    std::ofstream out(path, std::ios::binary);
    const std::uint32_t shape[2] = {(std::uint32_t)A.n(), (std::uint32_t)A.m()};
    out.write(reinterpret_cast<const char *>(shape), sizeof shape);
    out.write(reinterpret_cast<const char *>(h.data()), (std::streamsize)(h.size() * sizeof(float)));
}

// MGGCN_TIMING=1: host seconds of every start-up stage on stderr ("[mggcn timing] <stage> <s>"), for the set-up figures
// DESIGN.md quotes; off by default (the reference prints nothing there)
struct stage_timer {
    const bool on = std::getenv("MGGCN_TIMING") && std::string(std::getenv("MGGCN_TIMING")) != "0";
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void operator()(const char *what) {
        const auto now = std::chrono::steady_clock::now();
        if (on) std::cerr << "[mggcn timing] " << what << ' ' << std::chrono::duration<double>(now - last).count() << std::endl;
        last = now;
    }
};

static bool env_is(const char *name, const char *value) {
    const char *s = std::getenv(name);
    return s && std::string(s) == value;
}

int main_(int argc, char **argv) {
    opterr = 0;
    std::size_t P = 1, row_partition = 0, num_epochs = 20;
    bool overlap = true;
    while (optind < argc) {
        int c = getopt(argc, argv, "h?P:R:E:S:N");
        if (c == -1) break;
        switch (c) {
            case '?': return usage_(argv[0]);
            case 'h': usage_(argv[0]); return help_();
            case 'P': P = std::stoull(optarg); break;
            case 'R': row_partition = std::stoull(optarg); break;
            case 'E': num_epochs = std::stoull(optarg); break;
            case 'S': overlap = false; break;
            case 'N': overlap = true; break;                       // no_wait: parsed, unused (reference :67)
            default: throw arg_error("Unknown argument.");
        }
    }
    const bool fused = !env_is("MGGCN_FUSED", "0");
    const char *mode_env = std::getenv("MGGCN_DIST_MODE");
    const dist_mode mode = dist_mode_from_string(mode_env ? mode_env : "");
    const bool oversubscribe = std::getenv("MGGCN_OVERSUBSCRIBE") && !env_is("MGGCN_OVERSUBSCRIBE", "0");

    while (optind < argc && argv[optind] != nullptr) {
        const std::string command = argv[optind++];
        if (command.rfind("train", 0) != 0) throw arg_error("Unknown command.");
        if (optind >= argc) throw arg_error("train needs a dataset directory.");
        const std::filesystem::path dir = argv[optind++];
        if ((int)mggcn_device_count() < (int)std::max<std::size_t>(P, 1) && !(oversubscribe && mggcn_device_count() > 0))
            throw arg_error("not enough GPUs visible for -P");

        stage_timer stage;
        mggcn_set_device(0);
        stage("device");
        csr_matrix<x_t, v_t, r_t> A(dir / "graph.bin");
        dn_matrix<r_t> X(dir / "features.bin");
        dn_matrix<std::int32_t> Y(dir / "labels.bin");
        dn_matrix<std::int32_t> S(dir / "sets.bin");               // loaded, never used (reference :85)
        (void)S;
        stage("load-files");
        std::cerr << A.n() << ' ' << A.nnz() << std::endl;
        const auto labels = Y.to_host();
        const auto num_labels = 1 + *std::max_element(labels.begin(), labels.end());
        std::cerr << "num_labels = " << num_labels << std::endl;
        std::cerr << "feature size = " << X.m() << std::endl;

        if (optind >= argc) throw arg_error("train needs the number of hidden layers.");
        const int num_sizes = std::stoi(argv[optind++]);
        std::vector<std::size_t> sizes{X.m()};
        for (int i = 0; i < num_sizes; i++) {
            if (optind >= argc) throw arg_error("missing hidden layer width.");
            sizes.push_back(std::stoull(argv[optind++]));
        }
        sizes.push_back((std::size_t)num_labels);

        // csvs/<[permuted_]name>_<sizes>_<P>.csv (reference :100-111)
        std::string filename;
        bool permuted = false;
        for (const auto &part : (dir / "graph.bin").parent_path()) {
            if (part == "permuted") permuted = true;
            else if (!part.empty() && part != "/" && part != ".") filename = (permuted ? std::string("permuted_") : std::string("")) + part.string();
        }
        for (auto s : sizes) filename += "_" + std::to_string(s);
        std::filesystem::create_directories("csvs");
        std::ofstream of("csvs/" + filename + "_" + std::to_string(P) + ".csv");

        if (P <= 1 && !row_partition) {
            auto ctx = context(0);
            gcn<x_t, v_t, r_t> G(A, sizes, false, fused);
            if (env_is("MGGCN_HOIST_FIRST_AGGREGATION", "1")) G.set_hoist_first_aggregation(true);   // optional 6-SpMM epoch
            ctx.sync();
            stage("model (normalize, transpose, layers)");
            ctx.record("training-start", 0);
            for (std::size_t e = 0; e < num_epochs; e++) {
                if (const char *dd = std::getenv("MGGCN_DUMP_WEIGHTS")) {
                    std::filesystem::create_directories(dd);
                    for (std::size_t l = 0; l < G.layers().size(); l++) {
                        dump_dense(std::filesystem::path(dd) / ("e" + std::to_string(e) + "_W" + std::to_string(l) + ".bin"), G.layers()[l].W());
                        dump_dense(std::filesystem::path(dd) / ("e" + std::to_string(e) + "_b" + std::to_string(l) + ".bin"), G.layers()[l].b());
                    }
                }
                const auto start = std::chrono::system_clock::now();
                auto [loss, acc] = G.train_forward(ctx, X, Y);
                G.backward(ctx);
                G.adam_update(ctx, 1e-2, 0.9, 0.999, 5e-4, 1e-8);
                ctx.sync();
                const auto duration = std::chrono::duration<double>{std::chrono::system_clock::now() - start}.count();
                std::cerr << e << ' ' << loss << ' ' << acc << ' ' << duration << std::endl;
                if (e == 0) stage("epoch 0 (SpMM plans built on first use)");
                ctx.dump_timers(of, std::to_string(e) + "_0_");
            }
        } else if (row_partition) {
            sizes.back() = (sizes.back() + P - 1) / P * P;          // reference :135
            auto ctx = dist_context(P, overlap);
            std::vector<v_t> p(P + 1);
            for (std::size_t i = 1; i < p.size(); i++) p[i] = (v_t)(i * A.n() / P);
            stage("dist_context");
            if (stage.on) std::cerr << "[mggcn timing] transport " << ctx.transport() << " enqueue-threads " << (int)ctx.threaded() << std::endl;
            A.normalize(true);
            auto A_T = A.transpose();
            stage("normalize + transpose");
            dist_row_dn_matrix<std::int32_t> Yd(ctx, Y);
            dist_row_csr_matrix<x_t, v_t, r_t> Ad(ctx, A, p, p);
            dist_row_csr_matrix<x_t, v_t, r_t> A_Td(ctx, A_T, p, p);
            stage("block split (A, A_T)");
            dist_gcn<true, x_t, v_t, r_t> G(ctx, Ad, A_Td, sizes, false, fused, mode);
            dist_row_dn_matrix<r_t> Xd(ctx, X);
            ctx.sync();
            stage("model + shards");
            ctx.record("training-start", 0);
            for (std::size_t e = 0; e < num_epochs; e++) {
                const auto start = std::chrono::system_clock::now();
                const double waited = ctx.device_wait_seconds();
                auto [loss, acc] = G.train_forward(ctx, Xd, Yd);
                G.backward(ctx);
                G.adam_update(ctx, 1e-2, 0.9, 0.999, 5e-4, 1e-8);
                ctx.sync();
                const auto duration = std::chrono::duration<double>{std::chrono::system_clock::now() - start}.count();
                std::cerr << e << ' ' << loss << ' ' << acc << ' ' << duration << "\n";
                // how much of the epoch the host needed to ISSUE it (wall time minus the time it sat waiting for the devices)
                if (stage.on) std::cerr << "[mggcn timing] epoch " << e << " host-issue-ms " << (duration - (ctx.device_wait_seconds() - waited)) * 1e3 << std::endl;
                if (e == 0) stage("epoch 0 (exchange forms + SpMM plans built on first use)");
                ctx.dump_timers(of, std::to_string(e) + "_");
            }
        }
        // P > 1 without -R 1 trains nothing, exactly like the reference (:145, :171-189)
    }
    return EXIT_SUCCESS;
}

int main(int argc, char **argv) {
    try {
        return main_(argc, argv);
    } catch (const std::exception &e) {
        std::cerr << "Error: uncaught exception: '" << e.what() << "' Aborting." << std::endl;
        std::exit(EXIT_FAILURE);
    }
}
