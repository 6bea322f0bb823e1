// gcn.hpp -- the layer API of the reference (src/gcn.hpp) over the HIP engine.
//
// sparse_linear (:13-48), dist_sparse_linear (:50-86), linear (:88-189), dist_row_linear
// (:191-296), gcn_layer (:411-518), dist_gcn_layer (:520-637), softmax (:639-676),
// softmax_cross_entropy_loss (:769-823), dist_row_softmax_cross_entropy_loss (:872-935),
// gcn (:937-995), dist_gcn (:997-1056): same names, constructor arguments, members, buffer
// aliasing (model-wide HW_buffer; per-layer AHW_buffer holding AHW forward / G_out backward),
// layer-order rule (out <= in: GEMM first), skipped first-layer backward SpMM, timer names.
// `fused` (default on in gcn / dist_gcn) folds leaky-ReLU into the SpMM epilogue and uses the
// one-pass loss and one-launch Adam kernels; fused = false replays the reference's launches.
#pragma once

#include <cmath>
#include <numeric>
#include <optional>
#include <string>
#include <vector>

#include "dist_matrix.hpp"
#include "matrix.hpp"
#include "ops.hpp"

template <typename x_t, typename v_t, typename r_t>
class sparse_linear {
    using csr_t = csr_matrix<x_t, v_t, r_t>;
    using dn_t = dn_matrix<r_t>;
    std::string name;
    csr_t A, A_T;
    spmm_buffer ext_buffer, ext_buffer2;

public:
    sparse_linear(std::string name, csr_t A, csr_t A_T) : name(name), A(A), A_T(A_T) {}
    const csr_t &forward_matrix() const { return A; }

    void operator()(context ctx, dn_t B, dn_t C, bool discard = true, uint32_t flags = 0) {
        if (!ext_buffer) ext_buffer = get_matmul_buffer(ctx, A, B, C);      // plan is width-independent up to 128+
        ctx.record(name + "0_0_matmul-spmm", 0);
        matmul(ctx, A, B, C, ext_buffer, (r_t)1, discard ? (r_t)0 : (r_t)1, flags);
        ctx.record(name + "0_1_matmul-spmm", 0);
        ctx.register_timer(name + "0_matmul-spmm", name + "0_0_matmul-spmm", name + "0_1_matmul-spmm");
    }

    void backward(context ctx, dn_t G, dn_t G_out, bool discard = true) {
        if (!ext_buffer2) ext_buffer2 = get_matmul_buffer(ctx, A_T, G, G_out);
        ctx.record(name + "1_0_matmul-spmm", 0);
        matmul(ctx, A_T, G, G_out, ext_buffer2, (r_t)1, discard ? (r_t)0 : (r_t)1);
        ctx.record(name + "1_1_matmul-spmm", 0);
        ctx.register_timer(name + "1_matmul-spmm", name + "1_0_matmul-spmm", name + "1_1_matmul-spmm");
    }
};

template <bool row_partition, typename x_t, typename v_t, typename r_t>
class dist_sparse_linear {
    static_assert(row_partition, "only the row partition is live in the reference CLI");
    using csr_t = dist_row_csr_matrix<x_t, v_t, r_t>;
    using dn_t = dist_row_dn_matrix<r_t>;
    std::string name;
    csr_t A, A_T;
    std::vector<mggcn::device_ptr<r_t>> bcast_buffer, bcast_buffer2;
    dist_mode mode;
    dist_spmm_buffers ext, ext2;
    std::size_t M = 0, M2 = 0;                     // width the buffers were built for (reference :28-31)
    std::shared_ptr<dist_halo_plan<r_t>> halo, halo2;

    // receive buffers as matrices of the current width
    std::vector<dn_t> round_views(const dist_context &ctx, const dn_t &B) const {
        return {dn_t(ctx, B.n(), B.m(), bcast_buffer), dn_t(ctx, B.n(), B.m(), bcast_buffer2)};
    }
    std::vector<dn_matrix<r_t>> gather_views(const dist_context &ctx, const dn_t &B) const {
        std::vector<dn_matrix<r_t>> g;
        for (std::size_t j = 0; j < ctx.size(); j++) g.emplace_back(B.n(), B.m(), bcast_buffer[j]);
        return g;
    }

    void run(const dist_context &ctx, const csr_t &Mx, dn_t B, dn_t C, dist_spmm_buffers &e, std::size_t &width,
             std::shared_ptr<dist_halo_plan<r_t>> &hp, r_t beta, const std::string &tag, uint32_t flags) {
        if (width != B.m()) { e = get_matmul_buffer(ctx, Mx, B, C, mode); width = B.m(); }
        switch (mode) {
            case dist_mode::rounds: matmul(ctx, Mx, B, C, e, round_views(ctx, B), (r_t)1, beta, name + tag, flags); break;
            case dist_mode::allgather: matmul_allgather(ctx, Mx, B, C, e, gather_views(ctx, B), (r_t)1, beta, name + tag, flags); break;
            case dist_mode::halo:
                if (!hp) { ctx.drain(); hp = std::make_shared<dist_halo_plan<r_t>>(ctx, Mx); }
                matmul_halo(ctx, Mx, B, C, e, *hp, gather_views(ctx, B), (r_t)1, beta, name + tag, flags);
                break;
        }
    }

public:
    dist_sparse_linear(std::string name, csr_t A, csr_t A_T, std::vector<mggcn::device_ptr<r_t>> bcast_buffer,
                       std::vector<mggcn::device_ptr<r_t>> bcast_buffer2, dist_mode mode = dist_mode::allgather)
        : name(name), A(A), A_T(A_T), bcast_buffer(bcast_buffer), bcast_buffer2(bcast_buffer2), mode(mode) {}

    void operator()(const dist_context ctx, dn_t B, dn_t C, bool discard = true, uint32_t flags = 0) {
        run(ctx, A, B, C, ext, M, halo, discard ? (r_t)0 : (r_t)1, "0_", flags);
    }

    void backward(const dist_context ctx, dn_t G, dn_t G_out, bool discard = true) {
        run(ctx, A_T, G, G_out, ext2, M2, halo2, discard ? (r_t)0 : (r_t)1, "1_", 0u);
    }
};

// shared Adam step of linear / dist_row_linear (reference src/gcn.hpp:146-172)
template <typename r_t>
void adam_step(context ctx, bool fused, dn_matrix<r_t> W, dn_matrix<r_t> G_W, dn_matrix<r_t> mW, dn_matrix<r_t> vW, dn_matrix<r_t> b,
               dn_matrix<r_t> G_b, dn_matrix<r_t> mb, dn_matrix<r_t> vb, r_t lr, r_t beta1, r_t beta2, r_t wd, r_t eps, std::size_t step) {
    const r_t bc1 = 1 - std::pow(beta1, step);
    const r_t bc2 = 1 - std::pow(beta2, step);
    if (fused) {
        adam_fused(ctx, W, G_W, mW, vW, lr, beta1, beta2, wd, bc1, bc2, eps);
        adam_fused(ctx, b, G_b, mb, vb, lr, beta1, beta2, (r_t)0, bc1, bc2, eps);
        return;
    }
    axpy(ctx, W, G_W, wd);
    axpby(ctx, G_W, mW, 1 - beta1, beta1);
    axpby(ctx, G_b, mb, 1 - beta1, beta1);
    aaxpby(ctx, G_W, vW, 1 - beta2, beta2);
    aaxpby(ctx, G_b, vb, 1 - beta2, beta2);
    adam_final(ctx, W, mW, vW, lr, bc1, bc2, eps);
    adam_final(ctx, b, mb, vb, lr, bc1, bc2, eps);
}

template <typename r_t>
class linear {
    using dn_t = dn_matrix<r_t>;
    std::string name;
    dn_t W, G_W, mW, vW, b, G_b, mb, vb, X, ones;
    bool backward_out, fused;
    std::size_t step = 0;

public:
    linear(std::string name, std::size_t in, std::size_t out, bool backward_out = true, bool fused = false)
        : name(name), W(in, out), G_W(in, out), b(1, out), G_b(1, out), backward_out(backward_out), fused(fused) {
        W.init();
        b.init(std::sqrt((r_t)1.0 / 3));
    }

    void setX(dn_t new_X) { X = new_X; }

    void operator()(context ctx, dn_t X, dn_t XW, bool discard = true) {
        if (fused && discard) {                      // bias in the GEMM epilogue
            ctx.record(name + "0_0_matmul-gemm", 0);
            linear_forward(ctx, X, W, b, XW);
        } else {
            broadcast_rows(ctx, b, XW, discard);
            ctx.record(name + "0_0_matmul-gemm", 0);
            matmul(ctx, X, W, XW, (r_t)1, (r_t)1);
        }
        ctx.record(name + "0_1_matmul-gemm", 0);
        ctx.register_timer(name + "0_matmul-gemm", name + "0_0_matmul-gemm", name + "0_1_matmul-gemm");
        this->X = X;
    }

    // mask (fused path): the activated output Z of the layer below; G_out leaves the GEMM already multiplied by
    // leaky_relu'(Z), i.e. it IS that layer's T (reference :462-468)
    void backward(context ctx, dn_t G, dn_t G_out, bool discard = true, const dn_t *mask = nullptr) {
        if (ones.n() != 1 || ones.m() != G.n()) { ones = dn_t(1, G.n()); ones.fill(1); }
        ctx.record(name + "1_0_matmul-gemm", 0);
        if (fused) {                                  // G_b rides on the B tiles of the G_W kernel: one pass over G
            ctx.record(name + "1_1_matmul-gemm", 0);
            linear_backward_weights(ctx, X, G, G_W, G_b);
        } else {
            matmul(ctx, ones, G, G_b, (r_t)1, (r_t)0);
            ctx.record(name + "1_1_matmul-gemm", 0);
            matmul(ctx, X, G, G_W, (r_t)1, (r_t)0, true);
        }
        ctx.record(name + "1_2_matmul-gemm", 0);
        if (backward_out && mask) matmul_lrelu_backward(ctx, G, W, *mask, G_out, (r_t)1, false, true);
        else if (backward_out) matmul(ctx, G, W, G_out, (r_t)1, discard ? (r_t)0 : (r_t)1, false, true);
        ctx.record(name + "1_3_matmul-gemm", 0);
        ctx.register_timer(name + "1_matmul-gemm", name + "1_0_matmul-gemm", name + "1_3_matmul-gemm");
    }

    void update(const context ctx, const r_t lr, const r_t weight_decay) {
        axpby(ctx, G_W, W, -lr, 1 - weight_decay);
        axpy(ctx, G_b, b, -lr);
    }

    bool has_backward_out() const { return backward_out; }

    // Adam moments, zeroed on first use; bump_step() for the model-wide single launch (gcn::adam_update)
    void adam_state(context ctx) {
        if (mW.shape() != W.shape()) {
            mW = dn_t(W.shape()); vW = dn_t(W.shape()); mb = dn_t(b.shape()); vb = dn_t(b.shape());
            mW.zero(ctx); vW.zero(ctx); mb.zero(ctx); vb.zero(ctx);
            step = 0;
        }
    }
    std::size_t bump_step() { return ++step; }
    void adam_tensors(std::vector<std::array<dn_t, 4>> &out, std::vector<r_t> &wd, r_t weight_decay) const {
        out.push_back({W, G_W, mW, vW}); wd.push_back(weight_decay);       // W decays, b does not (reference :163)
        out.push_back({b, G_b, mb, vb}); wd.push_back((r_t)0);
    }

    void adam_update(context ctx, const r_t lr, const r_t beta1, const r_t beta2, const r_t weight_decay, const r_t eps) {
        adam_state(ctx);
        step += 1;
        ctx.record(name + "0_adam-update", 0);
        adam_step(ctx, fused, W, G_W, mW, vW, b, G_b, mb, vb, lr, beta1, beta2, weight_decay, eps, step);
        ctx.record(name + "1_adam-update", 0);
        ctx.register_timer(name + "adam-update", name + "0_adam-update", name + "1_adam-update");
    }

    auto get_b() { return b; }
    auto get_W() { return W; }
    auto get_G_W() { return G_W; }
    auto get_G_b() { return G_b; }
};

template <typename r_t>
class dist_row_linear {
    using dn_t = dist_row_dn_matrix<r_t>;
    using rdn_t = repl_dn_matrix<r_t>;
    std::string name;
    // G_W and G_b of a layer live in ONE buffer per GPU: [G_W | pad to 4 floats | G_b] -> a single in-place
    // all-reduce per layer (the reference all-reduces them separately, :236-240), run on the comm stream
    // while the backward pass goes on; awaited by finish_backward() / adam_update()
    std::vector<mggcn::device_ptr<r_t>> G_flat;
    std::size_t off_b = 0, flat_len = 0;
    rdn_t W, G_W, mW, vW, b, G_b, mb, vb, ones, G_all;
    dn_t X;
    bool backward_out, fused, pending = false;
    std::size_t step = 0;

    static std::vector<mggcn::device_ptr<r_t>> alloc_flat(const dist_context &ctx, std::size_t len) {
        std::vector<mggcn::device_ptr<r_t>> t;
        for (std::size_t i = 0; i < ctx.size(); i++) {
            ctx[i].set();
            t.push_back(mggcn::device_malloc<r_t>(len));
            mggcn_memset_zero(t.back().get(), len * sizeof(r_t), ctx[i].stream(0));     // the padding takes part in the sum
        }
        return t;
    }
    static std::vector<mggcn::device_ptr<r_t>> views(const std::vector<mggcn::device_ptr<r_t>> &owner, std::size_t off) {
        std::vector<mggcn::device_ptr<r_t>> t;
        for (const auto &o : owner) t.push_back(mggcn::device_view(o, off));
        return t;
    }

public:
    dist_row_linear(const dist_context ctx, std::string name, std::size_t in, std::size_t out, bool backward_out = true, bool fused = false)
        : name(name), off_b((in * out + 3) / 4 * 4), flat_len(off_b + out), W(ctx, in, out), b(ctx, 1, out),
          backward_out(backward_out), fused(fused) {
        G_flat = alloc_flat(ctx, flat_len);
        G_W = rdn_t(ctx, in, out, G_flat);
        G_b = rdn_t(ctx, 1, out, views(G_flat, off_b));
        G_all = rdn_t(ctx, 1, flat_len, G_flat);
        W.init(ctx);
        b.init(ctx, std::sqrt((r_t)1.0 / 3));
    }

    void setX(dn_t new_X) { X = new_X; }

    void operator()(dist_context ctx, dn_t X, dn_t XW, bool discard = true) {
        if (fused && discard) {
            ctx.record(name + "0_0_matmul-gemm", 0);
            linear_forward(ctx, X, W, b, XW);
        } else {
            broadcast_rows(ctx, b, XW, discard);
            ctx.record(name + "0_0_matmul-gemm", 0);
            matmul(ctx, X, W, XW, (r_t)1, (r_t)1);
        }
        ctx.record(name + "0_1_matmul-gemm", 0);
        ctx.register_timer(name + "0_matmul-gemm", name + "0_0_matmul-gemm", name + "0_1_matmul-gemm");
        this->X = X;
    }

    void backward(dist_context ctx, dn_t G, dn_t G_out, bool discard = true, const dn_t *mask = nullptr) {
        if (!fused && (ones.size() != ctx.size() || ones.m() != G.n() / ctx.size())) { ctx.drain(); ones = rdn_t(ctx, 1, G.n() / ctx.size()); ones.fill(ctx, 1); }
        const int cs = ctx.bcast_stream_id();
        ctx.record(name + "1_0_matmul-gemm", 0);
        for (std::size_t i = 0; i < ctx.size(); i++) {
            if (fused) {
                ctx.on(i, [c = ctx[i], x = X[i], g = G[i], gw = G_W[i], gb = G_b[i]] { linear_backward_weights(c, x, g, gw, gb); });
                continue;
            }
            ctx.on(i, [c = ctx[i], o = ones[i], x = X[i], g = G[i], gw = G_W[i], gb = G_b[i]] {
                matmul(c, o, g, gb, (r_t)1, (r_t)0);                          // G_b = 1^T G      (local part)
                matmul(c, x, g, gw, (r_t)1, (r_t)0, true);                    // G_W = X^T G      (local part)
            });
        }
        ctx.record(name + "1_1_grad-local", 0);
        ctx.wait(name + "1_1_grad-local", cs);
        G_all.allreduce(ctx, cs);                                             // [G_W | G_b] summed over the GPUs
        ctx.record(name + "1_2_grad-reduced", cs);
        pending = true;
        ctx.record(name + "1_2_matmul-gemm", 0);
        if (backward_out && mask)
            for (std::size_t i = 0; i < ctx.size(); i++)
                ctx.on(i, [c = ctx[i], g = G[i], w = W[i], z = (*mask)[i], go = G_out[i]] { matmul_lrelu_backward(c, g, w, z, go, (r_t)1, false, true); });
        else if (backward_out) matmul(ctx, G, W, G_out, (r_t)1, discard ? (r_t)0 : (r_t)1, true);
        ctx.record(name + "1_3_matmul-gemm", 0);
        ctx.register_timer(name + "1_matmul-gemm", name + "1_0_matmul-gemm", name + "1_3_matmul-gemm");
    }

    // the compute stream sees the summed gradients from here on
    void finish_backward(const dist_context ctx) {
        if (pending) ctx.wait(name + "1_2_grad-reduced", 0);
        pending = false;
    }

    bool has_backward_out() const { return backward_out; }

    void adam_state(dist_context ctx) {
        if (mW.size() != W.size()) {
            mW = rdn_t(ctx, W.shape()); vW = rdn_t(ctx, W.shape()); mb = rdn_t(ctx, b.shape()); vb = rdn_t(ctx, b.shape());
            mW.zero(ctx); vW.zero(ctx); mb.zero(ctx); vb.zero(ctx);
            step = 0;
        }
    }
    std::size_t bump_step() { return ++step; }
    void adam_tensors(std::size_t gpu, std::vector<std::array<dn_matrix<r_t>, 4>> &out, std::vector<r_t> &wd, r_t weight_decay) const {
        out.push_back({W[gpu], G_W[gpu], mW[gpu], vW[gpu]}); wd.push_back(weight_decay);
        out.push_back({b[gpu], G_b[gpu], mb[gpu], vb[gpu]}); wd.push_back((r_t)0);
    }

    void adam_update(dist_context ctx, const r_t lr, const r_t beta1, const r_t beta2, const r_t weight_decay, const r_t eps) {
        finish_backward(ctx);
        adam_state(ctx);
        step += 1;
        ctx.record(name + "0_adam-update", 0);
        for (std::size_t i = 0; i < ctx.size(); i++)
            ctx.on(i, [c = ctx[i], f = fused, w = W[i], gw = G_W[i], mw = mW[i], vw = vW[i], bb = b[i], gb = G_b[i], m_b = mb[i], v_b = vb[i],
                       lr, beta1, beta2, weight_decay, eps, st = step] {
                adam_step(c, f, w, gw, mw, vw, bb, gb, m_b, v_b, lr, beta1, beta2, weight_decay, eps, st);
            });
        ctx.record(name + "1_adam-update", 0);
        ctx.register_timer(name + "adam-update", name + "0_adam-update", name + "1_adam-update");
    }

    auto get_b() { return b; }
    auto get_W() { return W; }
    auto get_G_W() { return G_W; }
    auto get_G_b() { return G_b; }
};

template <typename x_t, typename v_t, typename r_t>
class gcn_layer {
    std::string name;
    sparse_linear<x_t, v_t, r_t> A;
    linear<r_t> lin;
    std::optional<linear<r_t>> res_lin;   // residual connection when in != out (reference :418, :430)
    bool residual_layer;
    dn_matrix<r_t> HW;                    // HW_buffer
    mggcn::device_ptr<r_t> AHW_buffer;
    dn_matrix<r_t> AHW, G_HW, G_out;      // AHW_buffer / HW_buffer / AHW_buffer
    bool activation, backward_spmm, fused;
    dn_matrix<r_t> H;
    // fused backward (set by the model): mask_input_grad -- my G_out GEMM applies leaky_relu'(H) of the layer
    // below; grad_premasked -- the G I receive already carries my own activation's mask
    bool mask_input_grad = false, grad_premasked = false;
    // optional, first layer only (gcn::set_hoist_first_aggregation): A_fwd . X computed once and kept
    bool hoist_input = false;
    dn_matrix<r_t> AX;
    const r_t *AX_src = nullptr;
    unsigned AX_generation = 0;

public:
    // A_fwd (1 b^T) = 1 b^T needs every row of A_fwd to sum to one: a vertex whose row of A_fwd is empty (no self-loop, nobody
    // points at it) would get 0 instead of b -- such a graph keeps the plain path.  (Re-)enabling drops the cached product:
    // the way to pick up an in-place change of the feature matrix, which the (buffer, shape, matrix generation) key cannot see.
    void set_hoist_input(bool on) {
        hoist_input = on && gemm_first() && !residual_layer && A.forward_matrix().every_row_nonempty();
        AX = dn_matrix<r_t>();
        AX_src = nullptr;
    }
    bool hoists_input() const { return hoist_input; }
    bool gemm_first() const { return HW.m() == AHW.m(); }           // out <= in (reference :439)
    bool has_activation() const { return activation; }
    bool propagates() const { return lin.has_backward_out(); }
    void set_fused_backward(bool mask_input, bool premasked) { if (mask_input) mask_input_grad = true; if (premasked) grad_premasked = true; }
    linear<r_t> &linear_layer() { return lin; }

    gcn_layer(std::string name, csr_matrix<x_t, v_t, r_t> A, csr_matrix<x_t, v_t, r_t> A_T, std::size_t in, std::size_t out,
              bool activation, bool residual_layer = false, bool backward_spmm = true,
              mggcn::device_ptr<r_t> HW_buffer = nullptr, bool fused = false)
        : name(name), A(name, A, A_T), lin(name, in, out, backward_spmm, fused),
          res_lin(in == out || !residual_layer ? std::nullopt : std::make_optional(linear<r_t>(name, in, out, backward_spmm, false))),
          residual_layer(residual_layer),
          HW(A.m(), std::min(in, out), HW_buffer ? HW_buffer : mggcn::device_malloc<r_t>(std::max<std::size_t>(A.m(), A_T.n()) * std::min(in, out))),
          AHW_buffer(mggcn::device_malloc<r_t>(std::max((std::size_t)A.n() * out, (std::size_t)A_T.n() * in))),
          AHW(A.n(), out, AHW_buffer), G_HW(A_T.n(), std::min(in, out), HW.shared_buffer()), G_out(A_T.n(), in, AHW_buffer),
          activation(activation), backward_spmm(backward_spmm), fused(fused) {}

    bool has_residual() const { return residual_layer; }
    std::vector<linear<r_t> *> linears() {
        std::vector<linear<r_t> *> v{&lin};
        if (res_lin) v.push_back(&*res_lin);
        return v;
    }

    auto operator()(context ctx, dn_matrix<r_t> H) {
        this->H = H;
        bool act_done = false;
        if (hoist_input && HW.m() == AHW.m()) {
            // layer 0's aggregation is loop-invariant: A_fwd (X W + 1 b^T) = (A_fwd X) W + 1 b^T (A_fwd is row-stochastic,
            // X never changes between epochs): A_fwd X once, one SpMM fewer per epoch.  NOT the reference's epoch
            // (:437-446): an option, off by default.  The backward pass stays the reference's (G_W = X^T T, :954).
            if (!AX.buffer() || AX_src != H.buffer() || AX.n() != AHW.n() || AX.m() != H.m() || AX_generation != A.forward_matrix().generation()) {
                AX = dn_matrix<r_t>(AHW.n(), H.m());
                A(ctx, H, AX);
                AX_src = H.buffer();
                AX_generation = A.forward_matrix().generation();
            }
            lin(ctx, AX, AHW);
            lin.setX(H);
        } else if (HW.m() == AHW.m()) {          // out <= in: GEMM first (reference :439-442)
            lin(ctx, H, HW);
            if (fused && activation) { A(ctx, HW, AHW, true, MGGCN_SPMM_LEAKY_RELU); act_done = true; }
            else A(ctx, HW, AHW);
        } else {                                  // reference :443-446
            A(ctx, H, HW);
            lin(ctx, HW, AHW);
        }
        if (activation && !act_done) {
            ctx.record(name + "0_0_activation", 0);
            leaky_relu_forward(ctx, AHW, AHW);
            ctx.record(name + "0_1_activation", 0);
            ctx.register_timer(name + "0_activation", name + "0_0_activation", name + "0_1_activation");
        }
        if (res_lin) (*res_lin)(ctx, H, AHW, false);          // reference :453-456
        else if (residual_layer) axpy(ctx, H, AHW, (r_t)1);
        return AHW;
    }

    auto backward(context ctx, dn_matrix<r_t> G) {
        auto T = G;
        if (activation && !grad_premasked) {
            ctx.record(name + "1_0_activation", 0);
            leaky_relu_backward(ctx, AHW, G, AHW);
            ctx.record(name + "1_1_activation", 0);
            ctx.register_timer(name + "1_activation", name + "1_0_activation", name + "1_1_activation");
            T = AHW;
        }
        if (HW.m() == AHW.m()) {
            auto g = G_HW;
            if (backward_spmm) A.backward(ctx, T, g); else g = T;
            lin.backward(ctx, g, G_out, true, mask_input_grad ? &H : nullptr);
            return residual_backward(ctx, G, G_out);
        }
        lin.setX(H);
        lin.backward(ctx, T, G_HW);
        if (backward_spmm) { A.backward(ctx, G_HW, G_out); return residual_backward(ctx, G, G_out); }
        return residual_backward(ctx, G, G_HW);
    }

    // reference :484-487: the residual branch sees the incoming (unmasked) gradient and adds to G_out
    dn_matrix<r_t> residual_backward(context ctx, dn_matrix<r_t> G, dn_matrix<r_t> out) {
        if (res_lin) res_lin->backward(ctx, G, out, false);
        else if (residual_layer) axpy(ctx, G, out, (r_t)1);
        return out;
    }

    void update(const context ctx, const r_t lr, const r_t wd) { for (auto *l : linears()) l->update(ctx, lr, wd); }
    void adam_update(const context ctx, const r_t lr, const r_t b1, const r_t b2, const r_t wd, const r_t eps) {
        for (auto *l : linears()) l->adam_update(ctx, lr, b1, b2, wd, eps);
    }
    auto b() { return lin.get_b(); }
    auto W() { return lin.get_W(); }
    auto GW() { return lin.get_G_W(); }
    auto Gb() { return lin.get_G_b(); }
};

template <bool row_partition, typename x_t, typename v_t, typename r_t>
class dist_gcn_layer {
    using csr_t = dist_row_csr_matrix<x_t, v_t, r_t>;
    using dn_t = dist_row_dn_matrix<r_t>;
    using bufs_t = std::vector<mggcn::device_ptr<r_t>>;
    std::string name;
    dist_sparse_linear<row_partition, x_t, v_t, r_t> A;
    dist_row_linear<r_t> lin;
    std::optional<dist_row_linear<r_t>> res_lin;      // reference :527
    bool residual_layer;
    bufs_t AHW_buffer;
    dn_t HW, AHW, G_HW, G_out;
    bool activation, backward_spmm, fused;
    dn_t H;
    bool mask_input_grad = false, grad_premasked = false;     // see gcn_layer

    static bufs_t alloc(const dist_context &ctx, std::size_t per_gpu) {
        bufs_t t;
        for (std::size_t i = 0; i < ctx.size(); i++) { ctx[i].set(); t.push_back(mggcn::device_malloc<r_t>(per_gpu)); }
        return t;
    }

public:
    dist_gcn_layer(const dist_context ctx, std::string name, csr_t A, csr_t A_T, std::size_t in, std::size_t out, bool activation,
                   bool residual_layer = false, bool backward_spmm = true, bufs_t HW_buffer = {}, bufs_t bcast_buffer = {},
                   bufs_t bcast_buffer2 = {}, bool fused = false, dist_mode mode = dist_mode::allgather)
        : name(name), A(name, A, A_T, bcast_buffer, bcast_buffer2, mode), lin(ctx, name, in, out, backward_spmm, fused),
          res_lin(in == out || !residual_layer ? std::nullopt : std::make_optional(dist_row_linear<r_t>(ctx, name, in, out, backward_spmm, false))),
          residual_layer(residual_layer),
          AHW_buffer(alloc(ctx, std::max(A.n() * out, A_T.n() * in) / ctx.size())), HW(ctx, A.m(), std::min(in, out), HW_buffer),
          AHW(ctx, A.n(), out, AHW_buffer), G_HW(ctx, A_T.n(), std::min(in, out), HW_buffer), G_out(ctx, A_T.n(), in, AHW_buffer),
          activation(activation), backward_spmm(backward_spmm), fused(fused) {}

    bool has_residual() const { return residual_layer; }
    std::vector<dist_row_linear<r_t> *> linears() {
        std::vector<dist_row_linear<r_t> *> v{&lin};
        if (res_lin) v.push_back(&*res_lin);
        return v;
    }

    auto operator()(dist_context ctx, dn_t H) {
        this->H = H;
        bool act_done = false;
        if (HW.m() == AHW.m()) {
            lin(ctx, H, HW);
            if (fused && activation) { A(ctx, HW, AHW, true, MGGCN_SPMM_LEAKY_RELU); act_done = true; }
            else A(ctx, HW, AHW);
        } else {
            A(ctx, H, HW);
            lin(ctx, HW, AHW);
        }
        if (activation && !act_done) {
            ctx.record(name + "0_0_activation", 0);
            leaky_relu_forward(ctx, AHW, AHW);
            ctx.record(name + "0_1_activation", 0);
            ctx.register_timer(name + "0_activation", name + "0_0_activation", name + "0_1_activation");
        }
        if (res_lin) (*res_lin)(ctx, H, AHW, false);          // reference :572-575
        else if (residual_layer)
            for (std::size_t i = 0; i < ctx.size(); i++) ctx.on(i, [c = ctx[i], h = H[i], a = AHW[i]] { axpy(c, h, a, (r_t)1); });
        return AHW;
    }

    dn_t residual_backward(dist_context ctx, dn_t G, dn_t out) {      // reference :603-606
        if (res_lin) res_lin->backward(ctx, G, out, false);
        else if (residual_layer)
            for (std::size_t i = 0; i < ctx.size(); i++) ctx.on(i, [c = ctx[i], g = G[i], o = out[i]] { axpy(c, g, o, (r_t)1); });
        return out;
    }

    bool gemm_first() const { return HW.m() == AHW.m(); }
    bool has_activation() const { return activation; }
    bool propagates() const { return lin.has_backward_out(); }
    void set_fused_backward(bool mask_input, bool premasked) { if (mask_input) mask_input_grad = true; if (premasked) grad_premasked = true; }
    dist_row_linear<r_t> &linear_layer() { return lin; }

    auto backward(dist_context ctx, dn_t G) {
        auto T = G;
        if (activation && !grad_premasked) {
            ctx.record(name + "1_0_activation", 0);
            leaky_relu_backward(ctx, AHW, G, AHW);
            ctx.record(name + "1_1_activation", 0);
            ctx.register_timer(name + "1_activation", name + "1_0_activation", name + "1_1_activation");
            T = AHW;
        }
        if (HW.m() == AHW.m()) {
            auto g = G_HW;
            if (backward_spmm) A.backward(ctx, T, g); else g = T;
            lin.backward(ctx, g, G_out, true, mask_input_grad ? &H : nullptr);
            return residual_backward(ctx, G, G_out);
        }
        lin.setX(H);
        lin.backward(ctx, T, G_HW);
        if (backward_spmm) { A.backward(ctx, G_HW, G_out); return residual_backward(ctx, G, G_out); }
        return residual_backward(ctx, G, G_HW);
    }

    void finish_backward(const dist_context ctx) { for (auto *l : linears()) l->finish_backward(ctx); }
    void adam_update(const dist_context ctx, const r_t lr, const r_t b1, const r_t b2, const r_t wd, const r_t eps) {
        for (auto *l : linears()) l->adam_update(ctx, lr, b1, b2, wd, eps);
    }
    auto b() { return lin.get_b(); }
    auto W() { return lin.get_W(); }
    auto GW() { return lin.get_G_W(); }
    auto Gb() { return lin.get_G_b(); }
};

// softmax: row max, exp(x - max), row sums by a GEMM with a ones vector, divide (reference :639-676)
template <typename r_t>
class softmax {
    dn_matrix<r_t> ones, H, H_R, maxs;
    const bool copy;

public:
    softmax(bool copy = true) : copy(copy) {}
    auto operator()(const context ctx, dn_matrix<r_t> temp) {
        if (copy) {
            if (!H.buffer()) H = dn_matrix<r_t>(temp.n(), temp.m());
            temp.copy_to(ctx, H);
        } else {
            H = temp;
        }
        if (!maxs.buffer()) maxs = dn_matrix<r_t>(H.n(), 1);
        max_rows(ctx, H, maxs);
        subtract_rows_exp(ctx, H, maxs, H);
        if (!ones.buffer()) { ones = dn_matrix<r_t>(H.m(), 1); ones.fill(1); }
        if (!H_R.buffer()) H_R = dn_matrix<r_t>(H.n(), 1);
        matmul(ctx, H, ones, H_R, (r_t)1, (r_t)0);
        scale_rows(ctx, H, H_R);
        return H;
    }
};

// One GPU's share of the loss: enqueues everything, leaves {sum|log p_y|, #correct} in
// sums_device; the caller synchronises and reads (reference :785-818 / :890-930).
template <typename r_t, typename x_t>
class loss_kernels {
    softmax<r_t> softmax_layer;
    dn_matrix<r_t> G, L, T;
    dn_matrix<x_t> P;
    const bool copy, fused;
    mggcn::device_ptr<r_t> sums_;

public:
    loss_kernels(bool copy, bool fused) : softmax_layer(copy), copy(copy), fused(fused) {}
    r_t *sums() const { return sums_.get(); }
    auto gradient() const { return G; }

    void enqueue(context ctx, dn_matrix<r_t> H, dn_matrix<x_t> Y, std::size_t n_global) {
        ctx.set();
        if (!sums_) sums_ = mggcn::host_malloc<r_t>(2);          // written by the kernels, read by the host after its sync
        if (fused) {
            if (copy) {                       // reference: copy, then in place (:653-656); here the pass writes elsewhere
                if (!G.buffer() || G.shape() != H.shape()) G = dn_matrix<r_t>(H.n(), H.m());
            } else {
                G = H;
            }
            mggcn_memset_zero(sums_.get(), 2 * sizeof(r_t), ctx.stream(0));
            softmax_xent_fused(ctx, H, G, Y, (r_t)1 / (r_t)n_global, sums_.get());
            return;
        }
        auto O = softmax_layer(ctx, H);
        if (!P.buffer()) P = dn_matrix<x_t>(Y.shape());
        max_row_indices(ctx, O, P);
        if (!L.buffer()) L = dn_matrix<r_t>(Y.shape());
        index_log_rows(ctx, O, Y, L);
        G = O;
        add_indexed_rows(ctx, G, Y, (r_t)-1);
        scale_mat(ctx, G, (r_t)1 / (r_t)n_global);
        if (!T.buffer()) T = dn_matrix<r_t>(Y.shape());
        is_equal(ctx, Y, P, T);
        abssum(ctx, L, sums_.get());
        abssum(ctx, T, sums_.get() + 1);
    }
};

template <typename r_t, typename x_t>
class softmax_cross_entropy_loss {
    std::string name;
    loss_kernels<r_t, x_t> k;

public:
    softmax_cross_entropy_loss(std::string name, bool copy = true, bool fused = false) : name(name), k(copy, fused) {}

    auto operator()(context ctx, dn_matrix<r_t> H, dn_matrix<x_t> Y) {
        ctx.record(name + "0_loss-layer", 0);
        k.enqueue(ctx, H, Y, Y.n());
        ctx.record(name + "1_loss-layer", 0);
        ctx.register_timer(name + "loss-layer", name + "0_loss-layer", name + "1_loss-layer");
        ctx.sync();
        const r_t *s = k.sums();                                 // mapped pinned host memory
        return std::make_pair(s[0] / H.n(), s[1] / H.n());
    }
    auto backward() { return k.gradient(); }
};

template <typename r_t, typename x_t>
class dist_row_softmax_cross_entropy_loss {
    std::string name;
    std::vector<std::shared_ptr<loss_kernels<r_t, x_t>>> ks;    // one per GPU, shared with the commands in flight
    const bool copy, fused;
    dist_row_dn_matrix<r_t> G;

public:
    dist_row_softmax_cross_entropy_loss(std::string name, bool copy = true, bool fused = false) : name(name), copy(copy), fused(fused) {}

    auto operator()(dist_context ctx, dist_row_dn_matrix<r_t> H, dist_row_dn_matrix<x_t> Y) {
        while (ks.size() < ctx.size()) ks.push_back(std::make_shared<loss_kernels<r_t, x_t>>(copy, fused));
        ctx.record(name + "0_loss-layer", 0);
        for (std::size_t i = 0; i < ctx.size(); i++)                                               // global n (reference :908)
            ctx.on(i, [k = ks[i], c = ctx[i], h = H[i], y = Y[i], n = Y.n()] { k->enqueue(c, h, y, n); });
        ctx.record(name + "1_loss-layer", 0);
        ctx.register_timer(name + "loss-layer", name + "0_loss-layer", name + "1_loss-layer");
        ctx.sync();
        r_t loss = 0, acc = 0;
        for (std::size_t i = 0; i < ctx.size(); i++) {          // host sum of the per-GPU scalars (reference :929)
            const r_t *s = ks[i]->sums();                       // mapped pinned host memory
            loss += s[0];
            acc += s[1];
        }
        G = H;                                                   // copy = false: gradient in place, as in dist_gcn
        if (copy) throw std::invalid_argument("dist loss with copy = true is not used by the reference CLI");
        return std::make_pair(loss / H.n(), acc / H.n());
    }
    auto backward() { return G; }
};

// fused backward: layer i+1's G_out GEMM applies layer i's leaky_relu' -- possible when layer i+1 is GEMM-first
// (its G_out comes out of a GEMM, reference :479-481) and propagates a gradient at all
template <typename layers_t>
void link_fused_backward(layers_t &layers, bool fused) {
    for (std::size_t i = 0; i + 1 < layers.size(); i++) {
        // (a residual branch needs the UNMASKED incoming gradient and adds to G_out afterwards: no fusion there)
        const bool ok = fused && layers[i].has_activation() && layers[i + 1].gemm_first() && layers[i + 1].propagates() &&
                        !layers[i].has_residual() && !layers[i + 1].has_residual();
        layers[i + 1].set_fused_backward(ok, false);
        layers[i].set_fused_backward(false, ok);
    }
}

template <typename x_t, typename v_t, typename r_t>
class gcn {
    std::vector<gcn_layer<x_t, v_t, r_t>> layers_;
    softmax_cross_entropy_loss<r_t, std::int32_t> loss_layer;
    mggcn::device_ptr<r_t> HW_buffer;

public:
    // normalises A by column, A_T = A^T, layers get (A_T, A) (reference :946-955)
    gcn(csr_matrix<x_t, v_t, r_t> A, std::vector<std::size_t> sizes, bool residual_layer = false, bool fused = true)
        : loss_layer(std::to_string(sizes.size() - 1) + "_", residual_layer, fused) {
        A.normalize(true);
        auto A_T = A.transpose();
        std::size_t max_d = 0;
        for (std::size_t i = 0; i + 1 < sizes.size(); i++) max_d = std::max(max_d, std::min(sizes[i], sizes[i + 1]));
        HW_buffer = mggcn::device_malloc<r_t>(std::max<std::size_t>(A.n(), A.m()) * max_d);
        for (std::size_t i = 1; i < sizes.size(); i++)
            layers_.emplace_back(std::to_string(i - 1) + "_", A_T, A, sizes[i - 1], sizes[i], i + 1 < sizes.size(), residual_layer,
                                 i != 1, HW_buffer, fused);
        link_fused_backward(layers_, fused);
        fused_ = fused;
        // the SpMM plans of the model, built side by side now instead of one by one inside the first epoch
        std::vector<typename csr_matrix<x_t, v_t, r_t>::plan_want> wants;
        const int dev = mggcn_get_device();
        for (std::size_t i = 1; i < sizes.size(); i++) {
            wants.push_back({A_T, std::min(sizes[i - 1], sizes[i]), dev});             // forward multiplies by A_T (:954)
            if (i != 1) wants.push_back({A, std::min(sizes[i - 1], sizes[i]), dev});   // the first layer's backward SpMM is skipped
        }
        csr_matrix<x_t, v_t, r_t>::prebuild_plans(wants);
    }

    // Optional mode (never the reference's epoch): pre-compute the first layer's aggregation A_fwd . X once -- valid while
    // the SAME feature matrix is passed every epoch (full-graph training does); 6 instead of 7 SpMMs per epoch on the
    // Reddit model.  `mg_gcn` turns it on with MGGCN_HOIST_FIRST_AGGREGATION=1.
    void set_hoist_first_aggregation(bool on) { layers_.front().set_hoist_input(on); }

    // test constructor with given weights (reference :957-963)
    gcn(csr_matrix<x_t, v_t, r_t> A, std::vector<std::size_t> sizes, std::vector<std::pair<std::vector<r_t>, std::vector<r_t>>> weights)
        : gcn(A, sizes) {
        for (std::size_t i = 0; i < layers_.size(); i++) {
            layers_[i].W().init(weights[i].first);
            layers_[i].b().init(weights[i].second);
        }
    }

    auto operator()(const context ctx, dn_matrix<r_t> H) {
        for (auto &layer : layers_) H = layer(ctx, H);
        return H;
    }
    auto train_forward(const context ctx, dn_matrix<r_t> H, dn_matrix<std::int32_t> Y) {
        H = operator()(ctx, H);
        return loss_layer(ctx, H, Y);
    }
    void backward(const context ctx) {
        auto G = loss_layer.backward();
        for (auto l = layers_.rbegin(); l != layers_.rend(); l++) G = l->backward(ctx, G);
    }
    void update(const context ctx, const r_t lr, const r_t wd) { for (auto &l : layers_) l.update(ctx, lr, wd); }
    void adam_update(const context ctx, const r_t lr, const r_t b1, const r_t b2, const r_t wd, const r_t eps) {
        if (!fused_) { for (auto &l : layers_) l.adam_update(ctx, lr, b1, b2, wd, eps); return; }
        // ONE launch for every parameter tensor of the model (reference: 7 launches per layer, :146-172, :990-994)
        std::size_t step = 0;
        for (auto &l : layers_)
            for (auto *lin : l.linears()) { lin->adam_state(ctx); step = lin->bump_step(); }
        if (!adam_ || adam_wd_ != wd) {
            std::vector<std::array<dn_matrix<r_t>, 4>> t;
            std::vector<r_t> w;
            for (auto &l : layers_)
                for (auto *lin : l.linears()) lin->adam_tensors(t, w, wd);
            adam_ = adam_table<r_t>(ctx, t, w);
            adam_wd_ = wd;
        }
        ctx.record("0_adam-update", 0);
        adam_.step(ctx, lr, b1, b2, (r_t)(1 - std::pow(b1, step)), (r_t)(1 - std::pow(b2, step)), eps);
        ctx.record("1_adam-update", 0);
        ctx.register_timer("adam-update", "0_adam-update", "1_adam-update");
    }
    auto &layers() { return layers_; }

private:
    bool fused_ = true;
    adam_table<r_t> adam_;
    r_t adam_wd_ = 0;
};

template <bool row_partition, typename x_t, typename v_t, typename r_t>
class dist_gcn {
    using csr_t = dist_row_csr_matrix<x_t, v_t, r_t>;
    using dn_t = dist_row_dn_matrix<r_t>;
    using idn_t = dist_row_dn_matrix<std::int32_t>;
    using bufs_t = std::vector<mggcn::device_ptr<r_t>>;
    std::vector<dist_gcn_layer<row_partition, x_t, v_t, r_t>> layers_;
    dist_row_softmax_cross_entropy_loss<r_t, std::int32_t> loss_layer;
    bufs_t HW_buffer, bcast_buffer, bcast_buffer2;

public:
    // per-GPU HW_buffer + receive buffers shared by all layers (reference :1016-1021).  The
    // all-gather schedule keeps the whole gathered B resident per GPU (n x max_d floats).
    dist_gcn(const dist_context ctx, csr_t A, csr_t A_T, std::vector<std::size_t> sizes, bool residual_layer = false,
             bool fused = true, dist_mode mode = dist_mode::allgather)
        : loss_layer(std::to_string(sizes.size() - 1) + "_", residual_layer, fused) {
        std::size_t max_d = 0;
        for (std::size_t i = 0; i + 1 < sizes.size(); i++) max_d = std::max(max_d, std::min(sizes[i], sizes[i + 1]));
        const std::size_t nmax = std::max(A.n(), A.m()), shard = nmax * max_d / ctx.size();
        for (std::size_t i = 0; i < ctx.size(); i++) {
            ctx[i].set();
            HW_buffer.push_back(mggcn::device_malloc<r_t>(shard));
            bcast_buffer.push_back(mggcn::device_malloc<r_t>(mode == dist_mode::rounds ? shard : nmax * max_d));
            bcast_buffer2.push_back(mggcn::device_malloc<r_t>(shard));
        }
        for (std::size_t i = 1; i < sizes.size(); i++)
            layers_.emplace_back(ctx, std::to_string(i - 1) + "_", A_T, A, sizes[i - 1], sizes[i], i + 1 < sizes.size(), residual_layer,
                                 i != 1, HW_buffer, bcast_buffer, bcast_buffer2, fused, mode);
        link_fused_backward(layers_, fused);
        fused_ = fused;
    }

    auto operator()(const dist_context ctx, dn_t H) {
        for (auto &layer : layers_) H = layer(ctx, H);
        return H;
    }
    auto train_forward(const dist_context ctx, dn_t H, idn_t Y) {
        H = operator()(ctx, H);
        return loss_layer(ctx, H, Y);
    }
    void backward(const dist_context ctx) {
        auto G = loss_layer.backward();
        for (auto l = layers_.rbegin(); l != layers_.rend(); l++) G = l->backward(ctx, G);
        for (auto &l : layers_) l.finish_backward(ctx);           // gradients are summed over the GPUs from here on
    }
    void adam_update(const dist_context ctx, const r_t lr, const r_t b1, const r_t b2, const r_t wd, const r_t eps) {
        if (!fused_) { for (auto &l : layers_) l.adam_update(ctx, lr, b1, b2, wd, eps); return; }
        std::size_t step = 0;
        for (auto &l : layers_) {
            l.finish_backward(ctx);
            for (auto *lin : l.linears()) { lin->adam_state(ctx); step = lin->bump_step(); }
        }
        if (adam_.size() != ctx.size() || adam_wd_ != wd) {
            adam_.clear();
            for (std::size_t g = 0; g < ctx.size(); g++) {
                std::vector<std::array<dn_matrix<r_t>, 4>> t;
                std::vector<r_t> w;
                for (auto &l : layers_)
                    for (auto *lin : l.linears()) lin->adam_tensors(g, t, w, wd);
                adam_.push_back(std::make_shared<adam_table<r_t>>(ctx[g], t, w));
            }
            adam_wd_ = wd;
        }
        ctx.record("0_adam-update", 0);
        for (std::size_t g = 0; g < ctx.size(); g++)
            ctx.on(g, [t = adam_[g], c = ctx[g], lr, b1, b2, c1 = (r_t)(1 - std::pow(b1, step)), c2 = (r_t)(1 - std::pow(b2, step)), eps] {
                t->step(c, lr, b1, b2, c1, c2, eps);
            });
        ctx.record("1_adam-update", 0);
        ctx.register_timer("adam-update", "0_adam-update", "1_adam-update");
    }
    auto &layers() { return layers_; }

private:
    bool fused_ = true;
    std::vector<std::shared_ptr<adam_table<r_t>>> adam_;     // one per GPU, shared with the commands in flight
    r_t adam_wd_ = 0;
};
