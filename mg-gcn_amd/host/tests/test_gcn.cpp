// test_gcn.cpp -- the reference's known-answer tests (test/test_gcn.cpp:98-249) restated over
// the C++ host layer: same literals, same call sequence, HIP kernels underneath.
#include <vector>

#include "check.hpp"
#include "gcn.hpp"

static const std::vector<float> kLogits = {2, 1, 2, 4, 2, 1, 1, -1, 0};

static void test_cross_entropy(const context ctx, bool fused) {
    dn_matrix<std::int32_t> Y(3, 1);
    Y.init({0, 0, 1});
    dn_matrix<float> logits(3, 3);
    logits.init(kLogits);
    softmax_cross_entropy_loss<float, std::int32_t> loss_layer("0_", true, fused);
    auto [loss, acc] = loss_layer(ctx, logits, Y);
    auto G = loss_layer.backward().to_host();
    CHECK_CLOSE(loss, 1.146482);
    const std::vector<float> expected = {-0.1925604, 0.0517875, 0.1407729, -0.0520684, 0.0380651, 0.0140034, 0.2217470, -0.3033231, 0.0815762};
    for (int i = 0; i < 9; i++) CHECK_CLOSE(G[i], expected[i]);
    (void)acc;
}

static void test_leaky_relu(const context ctx, bool fused) {
    dn_matrix<std::int32_t> Y(3, 1);
    Y.init({0, 0, 1});
    dn_matrix<float> logits(3, 3), H(3, 3);
    logits.init(kLogits);
    softmax_cross_entropy_loss<float, std::int32_t> loss_layer("0_", true, fused);
    leaky_relu_forward(ctx, logits, H);
    auto [loss, acc] = loss_layer(ctx, H, Y);
    auto G = loss_layer.backward();
    leaky_relu_backward(ctx, logits, G, G);
    const auto g = G.to_host();
    CHECK_CLOSE(loss, 0.8637248);
    const std::vector<float> expected = {-0.1925604, 0.0517875, 0.1407729, -0.0520684, 0.0380651, 0.0140034, 0.1924448, -0.0026324, 0.0007080};
    for (int i = 0; i < 9; i++) CHECK_CLOSE(g[i], expected[i]);
    (void)acc;
}

// test_g (dense A) and test_csr_g (CSR A through the SpMM)
static void test_g_chain(const context ctx, bool sparse) {
    dn_matrix<float> X(2, 3), W(3, 2), b(1, 2);
    X.init({4, 2, 1, 1, -1, 0});
    W.init({1, 2, -1, 0, 0.5, 1.5});
    b.init({1, 0.5});
    dn_matrix<std::int32_t> Y(2, 1);
    Y.init({0, 1});
    dn_matrix<float> XW(2, 2), AXW(2, 2), H(2, 2), ones(1, 2), G_b(1, 2), G_XW(2, 2), G_W(3, 2), G_out(2, 3);
    ones.init({1, 1});
    matmul(ctx, X, W, XW, 1.f, 0.f);
    broadcast_rows(ctx, b, AXW);
    csr_matrix<unsigned, unsigned, float> A({0, 1, 3}, {0, 0, 1}, {1, 0.5, 0.5}, 2);
    dn_matrix<float> Ad(2, 2);
    Ad.init({1, 0, 0.5, 0.5});
    if (sparse) {
        auto ext = get_matmul_buffer(ctx, A, XW, AXW, 1.f, 0.f);
        matmul(ctx, A, XW, AXW, ext, 1.f, 1.f);
    } else {
        matmul(ctx, Ad, XW, AXW, 1.f, 1.f);
    }
    leaky_relu_forward(ctx, AXW, H);
    softmax_cross_entropy_loss<float, std::int32_t> loss_layer("0_");
    auto [loss, acc] = loss_layer(ctx, H, Y);
    auto G = loss_layer.backward();
    leaky_relu_backward(ctx, AXW, G, G);
    matmul(ctx, ones, G, G_b, 1.f, 0.f);
    if (sparse) {
        auto A_t = A.transpose();
        auto ext2 = get_matmul_buffer(ctx, A_t, G, G_XW, 1.f, 0.f);
        matmul(ctx, A_t, G, G_XW, ext2, 1.f, 0.f);
    } else {
        matmul(ctx, Ad, G, G_XW, 1.f, 0.f, true);
    }
    matmul(ctx, X, G_XW, G_W, 1.f, 0.f, true);
    matmul(ctx, G_XW, W, G_out, 1.f, 0.f, false, true);
    ctx.sync();
    CHECK_CLOSE(loss, 3.2750449);
    const auto g = G.to_host(), gb = G_b.to_host(), gw = G_W.to_host(), go = G_out.to_host();
    const std::vector<float> eg = {-0.4992494, 0.4992494, 0.0237129, -0.0237129}, egb = {-0.4755365, 0.4755365};
    const std::vector<float> egw = {-1.9377153, 1.9377153, -0.9866424, 0.9866424, -0.4873929, 0.4873929};
    const std::vector<float> ego = {0.4873929, 0.4873929, 0.4873930, -0.0118565, -0.0118565, -0.0118565};
    for (int i = 0; i < 4; i++) CHECK_CLOSE(g[i], eg[i]);
    for (int i = 0; i < 2; i++) CHECK_CLOSE(gb[i], egb[i]);
    for (int i = 0; i < 6; i++) CHECK_CLOSE(gw[i], egw[i]);
    for (int i = 0; i < 6; i++) CHECK_CLOSE(go[i], ego[i]);
    (void)acc;
}

// one model-level run: the layer stack trains and the loss falls
static void test_gcn_trains(const context ctx, const char *toy_dir) {
    csr_matrix<unsigned, unsigned, float> A(std::string(toy_dir) + "/graph.bin");
    dn_matrix<float> X(std::string(toy_dir) + "/features.bin");
    dn_matrix<std::int32_t> Y(std::string(toy_dir) + "/labels.bin");
    gcn<unsigned, unsigned, float> G(A, {X.m(), 4, 2});
    float first = 0, last = 0;
    for (int e = 0; e < 30; e++) {
        auto [loss, acc] = G.train_forward(ctx, X, Y);
        G.backward(ctx);
        G.adam_update(ctx, 1e-2, 0.9, 0.999, 5e-4, 1e-8);
        ctx.sync();
        if (e == 0) first = loss;
        last = loss;
        (void)acc;
    }
    CHECK(std::isfinite(last) && last < first);
}

int main(int argc, char **argv) {
    const char *toy = argc > 1 ? argv[1] : "../../../tests/golden/toyB";
    const auto ctx = context(0);
    RUN(test_cross_entropy, ctx, false);
    RUN(test_cross_entropy, ctx, true);
    RUN(test_leaky_relu, ctx, false);
    RUN(test_leaky_relu, ctx, true);
    RUN(test_g_chain, ctx, false);
    RUN(test_g_chain, ctx, true);
    RUN(test_gcn_trains, ctx, toy);
    return g_failures != 0;
}
