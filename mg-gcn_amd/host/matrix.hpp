// matrix.hpp -- context, csr_matrix, dn_matrix of the C++17 host layer.
//
// Same class and member names as the reference's src/matrix.hpp (context :69-158,
// csr_matrix :214-468, dn_matrix :478-639) so that code written against the reference
// compiles against this layer.  Underneath, everything is the C ABI of include/mggcn.h:
//   * context       two prioritised streams + named events + timers + GEMM scratch
//   * csr_matrix    host CSR (source of truth for preprocessing) + lazily uploaded device copy
//   * dn_matrix<T>  row-major device matrix; host access = explicit blocking copies
// Differences forced by the platform: the reference allocates managed memory and touches it
// from the host through raw pointers (begin()/end(), operator[] returning a reference); here
// operator[] READS through a synchronising copy and writes go through init()/from_host().
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <ostream>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <thread>
#include <vector>

#include "mg_gcn.hpp"

class matrix_error : public std::runtime_error {
public:
    template <typename T>
    matrix_error(T t) : std::runtime_error(t) {}
};

// ---------------------------------------------------------------------------------------
// context: reference src/matrix.hpp:69-158
// ---------------------------------------------------------------------------------------
class context {
    struct state {
        std::size_t rank = 0;     // device ordinal this context drives
        mggcn_stream_t streams[2] = {nullptr, nullptr};
        std::map<std::string, mggcn_event_t> events;
        std::map<std::string, std::pair<std::string, std::string>> timers;
        mggcn::device_ptr<char> gemm_ws;
        std::size_t gemm_ws_bytes = 0;
        ~state() {
            mggcn_set_device((int)rank);
            for (auto &e : events) mggcn_event_destroy(e.second);
            for (auto &s : streams) mggcn_stream_destroy(s);
        }
    };
    std::shared_ptr<state> s_;   // contexts are passed by value and share their handles

public:
    context() = default;
    explicit context(std::size_t index) : s_(std::make_shared<state>()) {
        s_->rank = index;
        mggcn_set_device((int)index);
        // stream 0: low priority, every op; stream 1: high priority, communication
        // (reference stream_create(i,1) / stream_create(i,0), src/matrix.hpp:53-60, :82)
        s_->streams[0] = mggcn_stream_create(0);
        s_->streams[1] = mggcn_stream_create(1);
    }

    std::size_t device() const { return s_->rank; }
    void set() const { mggcn_set_device((int)s_->rank); }
    void sync() const { set(); mggcn_device_synchronize(); }
    mggcn_stream_t stream(std::size_t id = 0) const { return s_->streams[id]; }

    void record(const std::string &name, std::size_t stream_id) const {
        set();
        auto it = s_->events.find(name);
        if (it == s_->events.end()) it = s_->events.emplace(name, mggcn_event_create()).first;
        mggcn_event_record(it->second, s_->streams[stream_id]);
    }
    void wait(const std::string &name, std::size_t stream_id) const {
        set();
        mggcn_stream_wait_event(s_->streams[stream_id], s_->events.at(name));
    }
    void register_timer(const std::string &name, const std::string &beg, const std::string &end) const {
        s_->timers[name] = {beg, end};
    }
    float measure(const std::string &name) const {
        auto it = s_->timers.find(name);
        if (it == s_->timers.end()) return 0.f;
        set();
        return mggcn_event_elapsed_ms(s_->events.at(it->second.first), s_->events.at(it->second.second));
    }
    // "<prefix><name>:<ms>" per line, name-sorted (reference src/matrix.hpp:150-157)
    void dump_timers(std::ostream &out, const std::string &prefix) const {
        for (const auto &t : s_->timers) out << prefix << t.first << ':' << measure(t.first) << '\n';
    }

    // scratch for split-K GEMMs (cuBLAS keeps the equivalent inside its handle)
    void *gemm_workspace(std::size_t bytes) const {
        if (bytes > s_->gemm_ws_bytes) {
            set();
            sync();
            s_->gemm_ws = mggcn::device_malloc<char>(bytes);
            s_->gemm_ws_bytes = bytes;
        }
        return s_->gemm_ws.get();
    }
};

// scope timer of the reference (src/matrix.hpp:160-187): prints "<name> has started" / "<name> took <ms>ms" when built with
// -DLOG (CMake option LOG, src/CMakeLists.txt:10-13), silent otherwise; synchronises the context it is given before it
// reads the clock.  The reference only instantiates it in its column-partition code (src/cuda_utils.hpp:213-226,
// src/dist_matrix.hpp:359-367), which is out of scope here; the class is kept so that code written against it compiles.
inline std::mutex timer__iomutex;

template <typename ctx_t>
class timer {
    const std::string name;
    const std::chrono::high_resolution_clock::time_point start;
    ctx_t *ctx;
    std::mutex &mtx;
    std::ostream &os;

public:
    timer(std::string s, ctx_t *ctx = nullptr, std::mutex &mtx_ = timer__iomutex, std::ostream &os_ = std::cerr)
        : name(std::move(s)), start(std::chrono::high_resolution_clock::now()), ctx(ctx), mtx(mtx_), os(os_) {
#ifdef LOG
        std::lock_guard<std::mutex> lock(mtx);
        os << name << " has started" << std::endl;
#endif
    }
    ~timer() {
#ifdef LOG
        std::lock_guard<std::mutex> lock(mtx);
        if (ctx) ctx->sync();
        os << name << " took " << time() * 1000 << "ms" << std::endl;
#endif
    }
    double time() const {
        return std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::high_resolution_clock::now() - start).count();
    }
};

template <typename r_t>
class dn_matrix;

// the reference's opaque cuSPARSE workspace (cuda_ptr<char>, src/cuda_utils.hpp:94-102) becomes the SpMM plan
using spmm_buffer = std::shared_ptr<mggcn_spmm_plan>;

// ---------------------------------------------------------------------------------------
// csr_matrix: reference src/matrix.hpp:214-468
// ---------------------------------------------------------------------------------------
template <typename x_t, typename v_t, typename r_t>
class csr_matrix {
    static_assert(std::is_same_v<x_t, unsigned> && std::is_same_v<v_t, unsigned> && std::is_same_v<r_t, float>,
                  "the engine is CSR<u32,u32,f32> like the reference binary (src/main.cpp:43-45)");
    struct storage {
        v_t N = 0, M = 0;
        std::vector<x_t> indptr;
        std::vector<v_t> indices;
        std::vector<r_t> data;
        mggcn::device_ptr<x_t> d_indptr;
        mggcn::device_ptr<v_t> d_indices;
        mggcn::device_ptr<r_t> d_data;
        int device = -1;
        unsigned generation = 0;     // bumped whenever the values change (normalize): caches keyed on the matrix see it
        // SpMM plans built from this matrix, keyed (device, narrow lanes-per-row class | 0 = wide): the
        // layers of a model multiply by the same two matrices, so the plan (0.9 GB, ~1 s of host work at
        // the Reddit shape) is built once and shared (ops.hpp: get_matmul_buffer)
        std::map<std::pair<int, unsigned>, std::pair<unsigned, spmm_buffer>> plans;
    };
    std::shared_ptr<storage> st_ = std::make_shared<storage>();

    // PIGO-CSR-v2 (written by test/data/prep.py:46-76; read through pigo::CSR at
    // src/matrix.hpp:224-234): magic, two width bytes, u32 n, nnz, nrows, ncols, then arrays
    void read_file(const std::filesystem::path &path) {
        std::ifstream in(path, std::ios::binary);
        if (!in) throw matrix_error("cannot open " + path.string());
        char magic[11];
        in.read(magic, 11);
        unsigned char w[2] = {0, 0};
        in.read(reinterpret_cast<char *>(w), 2);
        if (!in || std::memcmp(magic, "PIGO-CSR-v2", 11) != 0) throw matrix_error(path.string() + ": not a PIGO-CSR-v2 file");
        if (w[0] != 4 || w[1] != 4) throw matrix_error(path.string() + ": only 4-byte index/offset widths are supported");
        std::uint32_t hdr[4];
        in.read(reinterpret_cast<char *>(hdr), sizeof hdr);
        if (!in) throw matrix_error(path.string() + ": truncated header");
        const std::uint32_t nnz = hdr[1];
        st_->N = hdr[2];
        st_->M = hdr[3];
        st_->indptr.resize((std::size_t)st_->N + 1);
        st_->indices.resize(nnz);
        st_->data.resize(nnz);
        in.read(reinterpret_cast<char *>(st_->indptr.data()), (std::streamsize)(st_->indptr.size() * sizeof(x_t)));
        in.read(reinterpret_cast<char *>(st_->indices.data()), (std::streamsize)((std::size_t)nnz * sizeof(v_t)));
        in.read(reinterpret_cast<char *>(st_->data.data()), (std::streamsize)((std::size_t)nnz * sizeof(r_t)));
        if (!in) throw matrix_error(path.string() + ": truncated payload");
    }

public:
    csr_matrix() = default;

    csr_matrix(const std::filesystem::path &path) {
        if (path.extension() != ".bin") throw matrix_error("File type is not supported.");   // src/matrix.hpp:282
        read_file(path);
    }

    csr_matrix(std::vector<x_t> indptr, std::vector<v_t> indices, std::vector<r_t> data, v_t M) {
        st_->N = (v_t)(indptr.size() - 1);
        st_->M = M;
        st_->indptr = std::move(indptr);
        st_->indices = std::move(indices);
        st_->data = std::move(data);
    }

    auto n() const { return st_->N; }
    auto m() const { return st_->M; }
    auto nnz() const { return st_->indptr.empty() ? 0u : st_->indptr[st_->N] - st_->indptr[0]; }
    auto begin(std::size_t i) const { return st_->indptr[i]; }
    auto end(std::size_t i) const { return st_->indptr[i + 1]; }
    auto operator[](std::size_t i) const { return st_->indices[i]; }
    auto shape() const { return std::make_pair((std::size_t)st_->N, (std::size_t)st_->M); }
    unsigned generation() const { return st_->generation; }
    // every row holds at least one entry (a normalised adjacency with this property is row-stochastic)
    bool every_row_nonempty() const {
        for (v_t r = 0; r < st_->N; r++) if (st_->indptr[r + 1] == st_->indptr[r]) return false;
        return true;
    }

    // host arrays (the reference returns its managed pointers here, src/matrix.hpp:263-265)
    const std::vector<x_t> &indptr() const { return st_->indptr; }
    const std::vector<v_t> &indices() const { return st_->indices; }
    const std::vector<r_t> &data() const { return st_->data; }

    // device copy, uploaded on first use on the calling context's GPU
    auto buffer() const {
        int dev = mggcn_get_device();
        if (!st_->d_indptr || st_->device != dev) {
            st_->d_indptr = mggcn::device_malloc<x_t>(st_->indptr.size());
            st_->d_indices = mggcn::device_malloc<v_t>(std::max<std::size_t>(st_->indices.size(), 1));
            st_->d_data = mggcn::device_malloc<r_t>(std::max<std::size_t>(st_->data.size(), 1));
            mggcn::upload(st_->d_indptr.get(), st_->indptr.data(), st_->indptr.size());
            if (!st_->indices.empty()) {
                mggcn::upload(st_->d_indices.get(), st_->indices.data(), st_->indices.size());
                mggcn::upload(st_->d_data.get(), st_->data.data(), st_->data.size());
            }
            st_->device = dev;
        }
        return std::make_tuple(st_->d_indptr, st_->d_indices, st_->d_data);
    }

    // reference src/matrix.hpp:340-390; axis == true: column-normalise (what gcn uses)
    void normalize(bool axis = false) {
        mggcn_csr_normalize_host(st_->N, st_->M, st_->indptr.data(), st_->indices.data(), st_->data.data(), axis);
        st_->d_indptr.reset();
        st_->plans.clear();          // a sweep plan carries its own copy of the values
        st_->generation++;
    }

    // plan of this matrix on the current device for feature width d (built on first use, shared by all
    // holders of this matrix; any plan serves any width <= its max_d, the hint only picks the fast form)
    spmm_buffer plan(std::size_t d) const {
        const int dev = mggcn_get_device();
        const unsigned form = (d >= 1 && d <= 64) ? (unsigned)((d + 15) / 16) : 0u;
        const unsigned max_d = (unsigned)std::max<std::size_t>(d, 128);
        auto &slot = st_->plans[{dev, form}];
        if (!slot.second || slot.first < max_d) {
            slot.second = spmm_buffer(mggcn_spmm_plan_create_for(st_->N, st_->M, st_->indptr.data(), st_->indices.data(),
                                                                 st_->data.data(), max_d, (unsigned)d),
                                      &mggcn_spmm_plan_destroy);
            slot.first = max_d;
        }
        return slot.second;
    }

    // Builds the plans (matrix, width, device) of a model CONCURRENTLY -- up to four host threads at a time, each plan
    // builder threading its own passes -- and files them in the matrices' caches: a model multiplies by two matrices at
    // two widths (four plans of 0.4-1.0 s of host work each at the Reddit shape), which built one after the other on
    // first use were 2.9 s inside the first epoch -- more than twenty epochs of training; the single-process P-GPU form
    // has 8 x (1 + 4) per matrix and width.  Entries whose plan exists already (or repeat an earlier entry) are skipped.
    struct plan_want { csr_matrix A; std::size_t d; int dev; };
    static void prebuild_plans(const std::vector<plan_want> &wants, std::size_t max_parallel = 4) {
        struct job { std::shared_ptr<storage> st; int dev; unsigned form, max_d, d; mggcn_spmm_plan *out = nullptr; };
        std::vector<job> jobs;
        for (const auto &w : wants) {
            const unsigned form = (w.d >= 1 && w.d <= 64) ? (unsigned)((w.d + 15) / 16) : 0u;
            const unsigned max_d = (unsigned)std::max<std::size_t>(w.d, 128);
            const auto it = w.A.st_->plans.find({w.dev, form});
            if (it != w.A.st_->plans.end() && it->second.second && it->second.first >= max_d) continue;
            bool dup = false;
            for (auto &j : jobs)
                if (j.st == w.A.st_ && j.form == form && j.dev == w.dev) { dup = true; if (max_d > j.max_d) { j.max_d = max_d; j.d = (unsigned)w.d; } }
            if (!dup) jobs.push_back({w.A.st_, w.dev, form, max_d, (unsigned)w.d});
        }
        const int dev0 = mggcn_get_device();
        mggcn_spmm_plan_concurrent_builders((std::uint32_t)std::min(jobs.size(), max_parallel));   // each builder threads over its share of the cores
        for (std::size_t lo = 0; lo < jobs.size(); lo += max_parallel) {
            std::vector<std::thread> th;
            for (std::size_t k = lo; k < std::min(jobs.size(), lo + max_parallel); k++)
                th.emplace_back([&jobs, k] {
                    job &j = jobs[k];
                    mggcn_set_device(j.dev);
                    j.out = mggcn_spmm_plan_create_for(j.st->N, j.st->M, j.st->indptr.data(), j.st->indices.data(), j.st->data.data(),
                                                       j.max_d, j.d);
                });
            for (auto &t : th) t.join();
        }
        mggcn_spmm_plan_concurrent_builders(1);
        for (auto &j : jobs) j.st->plans[{j.dev, j.form}] = {j.max_d, spmm_buffer(j.out, &mggcn_spmm_plan_destroy)};
        mggcn_set_device(dev0);
    }

    // reference src/matrix.hpp:392-453
    auto transpose() const {
        std::vector<x_t> t_indptr((std::size_t)st_->M + 1);
        std::vector<v_t> t_indices(st_->indices.size());
        std::vector<r_t> t_data(st_->data.size());
        mggcn_csr_transpose_host(st_->N, st_->M, st_->indptr.data(), st_->indices.data(), st_->data.data(),
                                 t_indptr.data(), t_indices.data(), t_data.data());
        return csr_matrix(std::move(t_indptr), std::move(t_indices), std::move(t_data), st_->N);
    }

    // dense copy on the host (reference src/matrix.hpp:328-337 returns a managed dn_matrix;
    // here the dense image is uploaded into one)
    dn_matrix<r_t> as_dn() const;

    void print(std::ostream &out) const {
        for (v_t v = 0; v < st_->N; v++)
            for (auto e = begin(v); e < end(v); e++) out << "(" << v << ", " << st_->indices[e] << ") : " << st_->data[e] << '\n';
        out << std::endl << std::endl;
    }
};

// ---------------------------------------------------------------------------------------
// dn_matrix: reference src/matrix.hpp:478-639
// ---------------------------------------------------------------------------------------
template <typename r_t>
class dn_matrix {
    static_assert(std::is_same_v<r_t, float> || std::is_same_v<r_t, std::int32_t>, "float matrices and int32 label vectors");
    std::size_t N_ = 0, M_ = 0;
    mggcn::device_ptr<r_t> buffer_;

public:
    dn_matrix() = default;

    // u32 N, u32 M, row-major payload (reference reader src/matrix.hpp:486-492)
    dn_matrix(const std::filesystem::path &path) {
        if (path.extension() != ".bin") throw matrix_error("File type is not supported.");   // src/matrix.hpp:518
        std::ifstream in(path, std::ios::binary);
        if (!in) throw matrix_error("cannot open " + path.string());
        std::uint32_t shape[2];
        in.read(reinterpret_cast<char *>(shape), sizeof shape);
        if (!in) throw matrix_error(path.string() + ": truncated header");
        N_ = shape[0];
        M_ = shape[1];
        std::vector<r_t> host(N_ * M_);
        in.read(reinterpret_cast<char *>(host.data()), (std::streamsize)(host.size() * sizeof(r_t)));
        if (!in) throw matrix_error(path.string() + ": truncated payload");
        buffer_ = mggcn::device_malloc<r_t>(N_ * M_);
        mggcn::upload(buffer_.get(), host.data(), host.size());
    }

    dn_matrix(std::size_t N, std::size_t M) : N_(N), M_(M), buffer_(mggcn::device_malloc<r_t>(N * M)) {}
    dn_matrix(std::pair<std::size_t, std::size_t> shape) : dn_matrix(shape.first, shape.second) {}
    // aliasing constructor: shares `buffer` (a null buffer allocates, as in the reference :497-501)
    dn_matrix(std::size_t N, std::size_t M, mggcn::device_ptr<r_t> buffer) : N_(N), M_(M), buffer_(std::move(buffer)) {
        if (!buffer_) buffer_ = mggcn::device_malloc<r_t>(N_ * M_);
    }
    dn_matrix(std::pair<std::size_t, std::size_t> shape, mggcn::device_ptr<r_t> buffer)
        : dn_matrix(shape.first, shape.second, std::move(buffer)) {}

    auto n() const { return N_; }
    auto m() const { return M_; }
    auto size() const { return N_ * M_; }
    auto shape() const { return std::make_pair(N_, M_); }
    r_t *buffer() const { return buffer_.get(); }
    auto shared_buffer() const { return buffer_; }

    // seed-99 uniform init (reference src/matrix.hpp:539-545): host generator, then upload
    void init(r_t gain = (r_t)std::sqrt(2 / (1 + 0.01 * 0.01))) {
        if constexpr (std::is_same_v<r_t, float>) {
            std::vector<float> host(size());
            mggcn_init_uniform_host(host.data(), N_, M_, gain);
            mggcn::upload(buffer_.get(), host.data(), host.size());
        }
    }
    // init(std::vector) overload (reference :547-549)
    void init(const std::vector<r_t> &values) { mggcn::upload(buffer_.get(), values.data(), std::min(values.size(), size())); }
    void fill(r_t value) { init(std::vector<r_t>(size(), value)); }

    std::vector<r_t> to_host() const {
        std::vector<r_t> host(size());
        if (size()) mggcn::download(host.data(), buffer_.get(), host.size());
        return host;
    }
    // element read through a synchronising copy (the reference reads managed memory)
    r_t operator[](std::size_t i) const {
        r_t v;
        mggcn::download(&v, buffer_.get() + i, 1);
        return v;
    }
    r_t operator[](std::pair<std::size_t, std::size_t> p) const { return (*this)[p.first * M_ + p.second]; }

    void copy_to(const context &ctx, const dn_matrix &other) const {
        ctx.set();
        mggcn_memcpy_d2d(other.buffer(), buffer(), size() * sizeof(r_t), ctx.stream(0));
    }
    dn_matrix copy(const context &ctx) const {
        dn_matrix clone(N_, M_);
        copy_to(ctx, clone);
        return clone;
    }
    void zero(const context &ctx) const {
        ctx.set();
        mggcn_memset_zero(buffer(), size() * sizeof(r_t), ctx.stream(0));
    }

    // out-of-place transpose (reference uses cublasSgeam, src/matrix.hpp:621-626; test helper):
    // A^T = A^T . I through the GEMM entry point
    dn_matrix transpose(const context &ctx) const;

    void print(std::ostream &out) const {
        const auto h = to_host();
        for (std::size_t i = 0; i < N_; i++) {
            for (std::size_t j = 0; j < M_; j++) out << h[i * M_ + j] << ' ';
            out << '\n';
        }
        out << std::endl << std::endl;
    }
};

template <typename x_t, typename v_t, typename r_t>
dn_matrix<r_t> csr_matrix<x_t, v_t, r_t>::as_dn() const {
    std::vector<r_t> dense((std::size_t)st_->N * st_->M, (r_t)0);
    for (v_t v = 0; v < st_->N; v++)
        for (auto e = begin(v); e < end(v); e++) dense[(std::size_t)v * st_->M + st_->indices[e]] = st_->data[e];
    dn_matrix<r_t> out(st_->N, st_->M);
    out.init(dense);
    return out;
}

template <typename r_t>
dn_matrix<r_t> dn_matrix<r_t>::transpose(const context &ctx) const {
    static_assert(std::is_same_v<r_t, float>);
    dn_matrix<r_t> eye(N_, N_), out(M_, N_);
    std::vector<r_t> id(N_ * N_, 0.f);
    for (std::size_t i = 0; i < N_; i++) id[i * N_ + i] = 1.f;
    eye.init(id);
    ctx.set();
    const auto ws = mggcn_gemm_workspace_bytes(1, 0, (uint32_t)M_, (uint32_t)N_, (uint32_t)N_);
    mggcn_gemm_f32(ctx.stream(0), 1, 0, (uint32_t)M_, (uint32_t)N_, (uint32_t)N_, 1.f, buffer(), M_, eye.buffer(),
                   N_, 0.f, out.buffer(), N_, ctx.gemm_workspace(ws), ws);
    return out;
}
