// test_matrix.cpp -- the reference's file / matrix plumbing tests (test/test_matrix.cpp:11-121)
// on the committed toyA / toyB fixtures.
#include <string>
#include <vector>

#include "check.hpp"
#include "matrix.hpp"
#include "ops.hpp"

static std::string g_dir;

static void read_binary_graph_test() {
    csr_matrix<unsigned, unsigned, float> A(g_dir + "/toyA/graph.bin");
    CHECK_EQ(A.n(), 4u); CHECK_EQ(A.m(), 4u); CHECK_EQ(A.nnz(), 8u);
    csr_matrix<unsigned, unsigned, float> B(g_dir + "/toyB/graph.bin");
    CHECK_EQ(B.n(), 4u); CHECK_EQ(B.m(), 4u); CHECK_EQ(B.nnz(), 12u);
}

static void read_binary_features_test() {
    dn_matrix<float> A(g_dir + "/toyA/features.bin");
    CHECK_EQ(A.n(), 4u); CHECK_EQ(A.m(), 2u);
}

static void unsupported_extension_test() {
    bool threw = false;
    try { csr_matrix<unsigned, unsigned, float> A(g_dir + "/toyA/graph.txt"); } catch (const matrix_error &) { threw = true; }
    CHECK(threw);
}

static void csr_to_dn_test() {
    csr_matrix<unsigned, unsigned, float> A(g_dir + "/toyA/graph.bin");
    const std::vector<float> expected = {0, 1, 0, 1, 1, 0, 1, 0, 0, 1, 0, 1, 1, 0, 1, 0};
    const auto dn = A.as_dn().to_host();
    for (std::size_t i = 0; i < expected.size(); i++) CHECK_EQ(dn[i], expected[i]);
}

static void test_dn_transpose() {
    const auto ctx = context(0);
    dn_matrix<float> A(2, 4);
    A.init({1, 2, 3, 4, 5, 6, 7, 8});
    const auto t = A.transpose(ctx).to_host();
    const std::vector<float> expected = {1, 5, 2, 6, 3, 7, 4, 8};
    for (std::size_t i = 0; i < expected.size(); i++) CHECK_EQ(t[i], expected[i]);
}

static void test_csr_transpose() {
    const auto ctx = context(0);
    csr_matrix<unsigned, unsigned, float> A(g_dir + "/toyB/graph.bin");
    const auto B = A.transpose();
    const auto want = A.as_dn().transpose(ctx).to_host();
    const auto got = B.as_dn().to_host();
    for (std::size_t i = 0; i < want.size(); i++) CHECK_EQ(got[i], want[i]);
}

int main(int argc, char **argv) {
    g_dir = argc > 1 ? argv[1] : "../../../tests/golden";
    RUN(read_binary_graph_test);
    RUN(read_binary_features_test);
    RUN(unsupported_extension_test);
    RUN(csr_to_dn_test);
    RUN(test_dn_transpose);
    RUN(test_csr_transpose);
    return g_failures != 0;
}
